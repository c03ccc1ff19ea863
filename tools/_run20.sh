cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -q -m gpu -x --durations=12 > gpurun_out/r6_suite4.txt 2>&1
grep -E "passed|failed" gpurun_out/r6_suite4.txt | tail -2; grep -E "^FAILED|^ERROR" gpurun_out/r6_suite4.txt | head
MP_RUN_SLOW=1 python -m pytest tests/test_headline_slow_gpu.py -q -m gpu -s > gpurun_out/r6_slow16k.txt 2>&1
tail -3 gpurun_out/r6_slow16k.txt; cat gpurun_out/parity_16k.txt
