python -m pytest tests -m gpu -x -q -k "mt19937 or reference or replay or same_seed or stream" > gpurun_out/r05_e_tests.log 2>&1; echo tests rc=$?; tail -3 gpurun_out/r05_e_tests.log
for i in 1 2; do
  timeout -k 10 120 python3 tools/mt_ab.py tools/bin/lib_polar_v1.so >> gpurun_out/r05_e_mt_ab.log 2>&1
  timeout -k 10 120 python3 tools/mt_ab.py - >> gpurun_out/r05_e_mt_ab.log 2>&1
done
cat gpurun_out/r05_e_mt_ab.log
