# A/B of two library variants on the weight-gradient shapes and the whole step, inside one box session
mkdir -p gpurun_out
A=$1; B=$2
cat > /tmp/wg_mb.py <<'PY'
import os, sys
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tests")); import helpers  # noqa
import torch, ops, corrif_hip as H
dev = "cuda:0"
def bench(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (R, M, N) in [(25088, 256, 1024), (25088, 1024, 256), (100352, 128, 512), (6272, 512, 2048), (401408, 256, 64), (100352, 512, 128)]:
    dY = torch.randn(R, M, device=dev); X = torch.randn(R, N, device=dev); dW = torch.empty(M, N, device=dev)
    ms = bench(lambda: ops.wgrad(dY.data_ptr(), M, X.data_ptr(), N, N, dW.data_ptr(), N, R, M, N, H.gemm_geom(), dev))
    print("wgrad 1x1   R %7d M %5d N %5d : %8.3f ms  %6.1f TF/s  splits %d" % (R, M, N, ms, 2.0 * R * M * N / ms / 1e9, H.lib().corrif_wgrad_plan(R, M, N)), flush=True)
for (B_, D, Hh, W, Ci, Co) in [(32, 4, 14, 14, 256, 256), (32, 4, 7, 7, 512, 512), (32, 4, 28, 28, 128, 128), (32, 4, 56, 56, 64, 64)]:
    x = torch.randn(B_, D, Hh, W, Ci, device=dev); gy = torch.randn(B_, D, Hh, W, Co, device=dev)
    gw = torch.empty(Co, 9 * Ci, device=dev)
    M = B_ * D * Hh * W
    geom = H.conv_geom((D, Hh, W), (D, Hh, W), (1, 3, 3), (1, 1, 1), (0, 1, 1))
    ms = bench(lambda: ops.wgrad(gy.data_ptr(), Co, x.data_ptr(), Ci, Ci, gw.data_ptr(), 9 * Ci, M, Co, 9 * Ci, geom, dev))
    print("wgrad 1x3x3 R %7d M %5d N %5d : %8.3f ms  %6.1f TF/s  splits %d" % (M, Co, 9 * Ci, ms, 2.0 * M * Co * 9 * Ci / ms / 1e9, H.lib().corrif_wgrad_plan(M, Co, 9 * Ci)), flush=True)
PY
for v in $A $B; do
echo "== $v"
CORRIF_LIB=$GRAFT_REPO_ROOT/variants/libcorrif_$v.so timeout -k 10 200 python /tmp/wg_mb.py 2>&1 | grep -v amdgpu
done
for v in $A $B $A $B; do
CORRIF_LIB=$GRAFT_REPO_ROOT/variants/libcorrif_$v.so timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timing > gpurun_out/ab_bench_$v.log 2>&1
echo "$v: $(grep '^{' gpurun_out/ab_bench_$v.log | cut -c60-150)"
done
