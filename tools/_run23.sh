cd $GRAFT_REPO_ROOT
python -m pytest tests/test_train_step_gpu.py tests/test_render_gpu.py::test_forward_eval_two_persons_128_samples_headline_config tests/test_bench_gpu.py::test_single_gpu_line_has_roofline_and_cpu_baseline -q -m gpu -rA -x > gpurun_out/r6_tests5.txt 2>&1
grep -E "passed|failed" gpurun_out/r6_tests5.txt | tail -2; grep -E "^FAILED|^ERROR" gpurun_out/r6_tests5.txt | head; grep -h "zero-pose term\|f16x2 sampler: acc_map\|f16x2 sampler: rgb" gpurun_out/r6_tests5.txt | cut -c1-220
grep -n "Error\|^E  " gpurun_out/r6_tests5.txt | head -20
