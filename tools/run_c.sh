python -m pytest tests -m gpu -x -q > gpurun_out/r05_c_tests.log 2>&1; echo tests rc=$?; tail -3 gpurun_out/r05_c_tests.log
for i in 1 2 3; do
  timeout -k 10 120 python3 tools/ab_native.py tools/bin/lib_noshare.so 20 >> gpurun_out/r05_c_share_ab.log 2>&1
  timeout -k 10 120 python3 tools/ab_native.py - 20 >> gpurun_out/r05_c_share_ab.log 2>&1
done
cat gpurun_out/r05_c_share_ab.log
timeout -k 10 200 python3 tools/bench2048.py tools/bin/lib_noshare.so quick > gpurun_out/r05_c_2048_noshare.log 2>&1; timeout -k 10 200 python3 tools/bench2048.py - quick > gpurun_out/r05_c_2048_share.log 2>&1
tail -5 gpurun_out/r05_c_2048_noshare.log gpurun_out/r05_c_2048_share.log
for i in 1 2; do timeout -k 10 200 python3 tools/ln_prof.py >> gpurun_out/r05_c_ln.log 2>&1; done; cat gpurun_out/r05_c_ln.log
