run() { echo -n "$* : "; env "$@" python bench.py --workload c4 --no-cpu-baseline --steps 10 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('ms/step %.3f kernel %.3f'%(d['ms_per_step'], d['roofline']['kernel_ms']))"; }
run FMX_FRONTIER_WGS=5
run FMX_FRONTIER_WGS=10
run FMX_FRONTIER_WGS=20
run FMX_FRONTIER_WGS=40
run FMX_FRONTIER_WGS=20 FMX_FRONTIER_ROUNDS=32
run FMX_FRONTIER_WGS=20 FMX_FRONTIER_ROUNDS=64 FMX_FRONTIER_CHAIN=6
run FMX_FRONTIER_WGS=40 FMX_FRONTIER_ROUNDS=64 FMX_FRONTIER_CHAIN=6
run FMX_FRONTIER_WGS=80 FMX_FRONTIER_ROUNDS=128 FMX_FRONTIER_CHAIN=4
