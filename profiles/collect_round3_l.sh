#!/bin/bash
# round 3: the tracked frame — small blob tier alone, nearest-neighbour search from LDS, host-side time split, spin wait
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3l
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R && timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "track or lockstep or sequence" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
cd /tmp
for w in 0 1 0 1; do timeout 200 python $R/bench_streams.py --streams 1 --frames 400 --wait-spin $w 2>/dev/null | tail -1 > $O/streams1_spin${w}_$RANDOM.json; done
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -o s -- python $R/bench_streams.py --streams 1 --frames 400 > $O/stats1.log 2>&1
find $O -name "*kernel_trace.csv" -delete
ls $O
