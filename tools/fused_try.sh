run() { echo "$*"; env "$@" timeout -k 10 120 python bench.py --steps 5 --warmup 1 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ', d['ms_per_step'], d['config']['rms_last'], d['pipeline']['kernel_ms'])"; }
run RANDOMFIELD_FUSED=0
run RANDOMFIELD_FUSED_G=2
run RANDOMFIELD_FUSED_G=4
run RANDOMFIELD_FUSED_G=8
run RANDOMFIELD_FUSED_G=16
run RANDOMFIELD_FUSED_G=32
run RANDOMFIELD_FUSED_G=8 RANDOMFIELD_FUSED_SLACK=1000000
run RANDOMFIELD_FUSED_G=8 RANDOMFIELD_FUSED_SLACK=0
