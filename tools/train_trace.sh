set -e
python tools/train_bench.py 20 4 > gpurun_out/train_bench_hull.txt 2>&1 || true
tail -3 gpurun_out/train_bench_hull.txt
R=$PWD; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/tk; rocprofv3 --kernel-trace -d /tmp/tk -o tk -- python $R/tools/train_bench.py 20 4 > /tmp/tk.log 2>&1 || tail -5 /tmp/tk.log
DB=$(find /tmp/tk -name '*.db' | head -1)
python $R/tools/rocpd_summary.py $DB $R/gpurun_out/train_kernels_hull.txt | head -30
