#!/bin/bash
# round 3: the frontier's in-lane row loop (FMX_FAN) A/B on C4 + parity; the host-pointer path's chunk count
O=gpurun_out/${1:-r03y}; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { echo build failed; exit 1; }
bash tools/build_variant.sh nofan -DFMX_FAN=0 > $O/build_nofan.log 2>&1 || { echo variant build failed; exit 1; }
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "regex or c4 or frontier" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for rep in 1 2; do
  timeout -k 10 300 python tools/c4_quick.py 40 2>&1 | grep -v amdgpu.ids | tail -2 > $O/c4_fan_$rep.txt; tail -1 $O/c4_fan_$rep.txt
  FMX_LIB=findex_amd/lib/variants/libfmx_nofan.so timeout -k 10 300 python tools/c4_quick.py 40 2>&1 | grep -v amdgpu.ids | tail -2 > $O/c4_nofan_$rep.txt; tail -1 $O/c4_nofan_$rep.txt
done
if [ "$2" = host ]; then
for ch in 8 1 2 4; do
  FMX_PIPE_CHUNKS=$ch timeout -k 10 300 python tools/measure_host_path.py c3 2>&1 | grep "pinned\|pageable" > $O/host_chunks_$ch.txt; echo "chunks=$ch"; cat $O/host_chunks_$ch.txt
done
FMX_JUMP=0 timeout -k 10 300 python tools/measure_host_path.py c3 2>&1 | grep "pinned\|pageable" > $O/host_nojump.txt; echo "no jump table"; cat $O/host_nojump.txt
HSA_ENABLE_SDMA=0 timeout -k 10 300 python tools/measure_host_path.py c3 2>&1 | grep "pinned\|pageable" > $O/host_nosdma.txt; echo "no sdma"; cat $O/host_nosdma.txt
fi
