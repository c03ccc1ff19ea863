cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -q -m gpu -rA --durations=15 > gpurun_out/r6_suite3.txt 2>&1
grep -E "passed|failed" gpurun_out/r6_suite3.txt | tail -3
grep -E "^FAILED|^ERROR" gpurun_out/r6_suite3.txt | head -40
