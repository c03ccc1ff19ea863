cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_train_gpu.py tests/test_train_step_gpu.py tests/test_loss_gpu.py tests/test_parallel_gpu.py -q -m gpu -rA > gpurun_out/r6_train_tests.txt 2>&1
grep -E "passed|failed" gpurun_out/r6_train_tests.txt | tail -2; grep -E "^FAILED|^ERROR" gpurun_out/r6_train_tests.txt | head
grep -h "worst relative parameter-gradient\|bench workload" gpurun_out/r6_train_tests.txt | cut -c1-400 | head
python tools/train_bench.py 60 4 > gpurun_out/r6_train_bench_v16.txt 2>&1; tail -4 gpurun_out/r6_train_bench_v16.txt
