run() { echo -n "$* : "; env "$@" python bench.py --workload c4 --no-cpu-baseline --steps 10 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('ms/step %.3f kernel %.3f'%(d['ms_per_step'], d['roofline']['kernel_ms']))"; }
run FMX_FRONTIER_ROUNDS=8 FMX_FRONTIER_CHAIN=8
run FMX_FRONTIER_ROUNDS=12 FMX_FRONTIER_CHAIN=8
run FMX_FRONTIER_ROUNDS=16 FMX_FRONTIER_CHAIN=8
run FMX_FRONTIER_ROUNDS=24 FMX_FRONTIER_CHAIN=6
run FMX_FRONTIER_ROUNDS=16 FMX_FRONTIER_CHAIN=8 FMX_FRONTIER_WGS=4
run FMX_FRONTIER_ROUNDS=16 FMX_FRONTIER_CHAIN=8 FMX_FRONTIER_WGS=3
run FMX_FRONTIER_ROUNDS=16 FMX_FRONTIER_CHAIN=8 FMX_FRONTIER_WGS=2
