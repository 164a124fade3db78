#!/bin/bash
# round 3: new defaults (32768-frame sub-batches, 25 % side scan), the 16-lane refinement kernel on the tracked path.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3f
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R && timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
cd /tmp
python $R/bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
python $R/bench.py --no-cpu --no-host-leg --steps 40 --warmup 5 > $O/bench40.json 2>/dev/null
python $R/bench.py --no-cpu --no-host-leg --steps 20 --pipeline 12 > $O/bench_p12.json 2>/dev/null
python $R/bench.py --no-cpu --no-host-leg --steps 20 --pipeline 10 > $O/bench_p10.json 2>/dev/null
python $R/bench_streams.py --streams 1 --frames 400 2>/dev/null | tail -1 > $O/streams1.json
python $R/bench_streams.py --streams 8 --frames 300 --lockstep 2>/dev/null | tail -1 > $O/lockstep8.json
python $R/bench_streams.py --streams 64 --frames 300 --lockstep 2>/dev/null | tail -1 > $O/lockstep64.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_lockstep -o s -- python $R/bench_streams.py --streams 64 --frames 300 --lockstep > $O/stats_lockstep.log 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
ls -la $O
