python -m pytest tests -m gpu -x -q -k "mt19937 or reference or replay or same_seed or stream" > gpurun_out/r05_d_tests.log 2>&1; echo tests rc=$?; tail -3 gpurun_out/r05_d_tests.log
for i in 1 2; do
  for v in polar_v1 polar_pairs_only polar_tiers_only; do timeout -k 10 120 python3 tools/mt_ab.py tools/bin/lib_$v.so >> gpurun_out/r05_d_mt_ab.log 2>&1; done
  timeout -k 10 120 python3 tools/mt_ab.py - >> gpurun_out/r05_d_mt_ab.log 2>&1
done
cat gpurun_out/r05_d_mt_ab.log
