for lib in ""; do
  if [ -n "$lib" ]; then export MP_LIB_PATH=$PWD/multiply_amd/ab_libs/libmultiply_hip_$lib.so; else unset MP_LIB_PATH; fi
  echo "== ${lib:-default}"
  R=$PWD; cd /tmp; export TMPDIR=/tmp
  rm -rf /tmp/tk; rocprofv3 --kernel-trace -d /tmp/tk -o tk -- python $R/tools/train_bench.py 12 3 > /tmp/tk.log 2>&1 || tail -5 /tmp/tk.log
  grep ms_per_iter /tmp/tk.log | cut -c1-40
  DB=$(find /tmp/tk -name '*.db' | head -1)
  python $R/tools/rocpd_summary.py $DB | grep -E "tn_b3w|tn_b3g"
  cd $R
done
