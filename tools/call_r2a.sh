set -x
mkdir -p gpurun_out/golden
(python tests/golden/make_golden_large.py gpurun_out/golden oracle_train_b2_d8_256 > gpurun_out/golden_d8.log 2>&1 &)
python -m pytest tests -m gpu -x -q \
  --deselect "tests/test_model_gpu.py::test_against_reference_fixture_tame[tame_train_b4_d4_224]" \
  --deselect "tests/test_model_gpu.py::test_against_reference_fixture_tame[tame_eval_b3_d3_224]" \
  --deselect "tests/test_model_gpu.py::test_against_reference_fixture_kaiming_bracketed[kaiming_train_b2_d4_224]" \
  --deselect "tests/test_model_gpu.py::test_stage_taps_against_reference_fixture[tame_train_b4_d4_224]" \
  --deselect "tests/test_model_gpu.py::test_stage_taps_against_reference_fixture[kaiming_train_b2_d4_224]" \
  --deselect "tests/test_model_gpu.py::test_large_baseline_configs_fwd_bwd" \
  --deselect "tests/test_train_gpu.py::test_hip_graph_eval_forward_is_bit_identical_to_eager" \
  > gpurun_out/t_r2a.log 2>&1
echo "pytest rc=$?" >> gpurun_out/t_r2a.log
python -m pytest tests/test_train_gpu.py -m gpu -x -q -k hip_graph > gpurun_out/t_r2a_graph.log 2>&1
echo "graph rc=$?" >> gpurun_out/t_r2a_graph.log
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --dump-shapes gpurun_out/shapes_r2a.jsonl > gpurun_out/bench_r2a.log 2>&1
echo "bench rc=$?" >> gpurun_out/bench_r2a.log
for i in $(seq 1 60); do if grep -q "wrote" gpurun_out/golden_d8.log; then break; fi; sleep 10; echo waiting $i; tail -1 gpurun_out/golden_d8.log; done
tail -3 gpurun_out/t_r2a.log gpurun_out/t_r2a_graph.log gpurun_out/bench_r2a.log gpurun_out/golden_d8.log
