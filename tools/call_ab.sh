# A/B of library variants inside one box session: tools/call_ab.sh <variantA> <variantB> [bench args]
mkdir -p gpurun_out
A=$1; B=$2
for v in $A $B; do
CORRIF_LIB=$GRAFT_REPO_ROOT/variants/libcorrif_$v.so timeout -k 10 200 python tools/gemm_vs_lab.py > gpurun_out/ab_lab_$v.log 2>&1
echo "== $v"; grep -v amdgpu gpurun_out/ab_lab_$v.log | cut -c1-90
done
for v in $A $B $A $B; do
CORRIF_LIB=$GRAFT_REPO_ROOT/variants/libcorrif_$v.so timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timing > gpurun_out/ab_bench_$v.log 2>&1
echo "$v: $(grep '^{' gpurun_out/ab_bench_$v.log | cut -c60-150)"
done
