for i in 1 2; do
python3 tools/ab_native.py -
python3 tools/ab_native.py tools/bin/lib_noslp_rf_k_col_plain.so
python3 tools/ab_native.py tools/bin/lib_noslp_rf_k_row.so
done
for i in 1 2; do
python3 tools/cfg_workload.py f64 4; python3 tools/cfg_workload.py f64 4 tools/bin/lib_noslp_rf_k_col_gen64.so
python3 tools/cfg_workload.py ref 4; python3 tools/cfg_workload.py ref 4 tools/bin/lib_noslp_rf_k_mt.so
python3 tools/cfg_workload.py 2048 4; python3 tools/cfg_workload.py rank0 4
done
python -m pytest tests -x -q -m gpu -k "distributed_generator or r2c" 2>&1 | tail -3
