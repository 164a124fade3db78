#!/bin/bash
# round 3: whole GPU suite after the tracked-frame changes (small blob tier alone, single and lock-step) + tracked benches
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3m
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R && timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
cd /tmp
timeout 200 python $R/bench_streams.py --streams 1 --frames 400 2>/dev/null | tail -1 > $O/streams1.json
for n in 8 64; do timeout 200 python $R/bench_streams.py --streams $n --frames 300 --lockstep 2>/dev/null | tail -1 > $O/lockstep$n.json; done
ls $O
