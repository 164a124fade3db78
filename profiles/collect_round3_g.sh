#!/bin/bash
# round 3: leaner scan kernel (split sweep again), tracked path with zero-copy mailbox + small blob tier first.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3g
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R && timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
cd /tmp
for pct in 25 30 35 40; do for blk in 2 3; do
  python $R/bench.py --no-cpu --no-host-leg --steps 20 --warmup 5 --scan-split-pct $pct --side-scan-blocks $blk 2>/dev/null > $O/sweep_${pct}_${blk}.json
done; done
python $R/bench_streams.py --streams 1 --frames 400 2>/dev/null | tail -1 > $O/streams1.json
python $R/bench_streams.py --streams 8 --frames 300 --lockstep 2>/dev/null | tail -1 > $O/lockstep8.json
python $R/bench_streams.py --streams 64 --frames 300 --lockstep 2>/dev/null | tail -1 > $O/lockstep64.json
python $R/bench_streams.py --streams 256 --frames 300 --lockstep --groups 4 --group-threads 4 2>/dev/null | tail -1 > $O/lockstep256g4t4.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_lockstep -o s -- python $R/bench_streams.py --streams 64 --frames 300 --lockstep > $O/stats_lockstep.log 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
ls $O
