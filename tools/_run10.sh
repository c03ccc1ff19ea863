cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6_smoke.txt 2>&1; tail -2 gpurun_out/r6_smoke.txt
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --breakdown > gpurun_out/r6_bench_quick.json 2> gpurun_out/r6_bench_quick.err; tail -c 1500 gpurun_out/r6_bench_quick.json; tail -25 gpurun_out/r6_bench_quick.err
python -m pytest tests -q -m gpu -rA --durations=15 > gpurun_out/r6_suite2.txt 2>&1
grep -E "passed|failed" gpurun_out/r6_suite2.txt | tail -3
grep -E "^FAILED|^ERROR" gpurun_out/r6_suite2.txt | head -40
