#!/bin/bash
# round 3, second GPU call: the streaming entry (submit / collect + prefetch of the next batch's first scan), the
# side-stream probe, and a sweep of the scan split now that the voting kernel is lighter.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3b
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R && timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
cd /tmp
python $R/bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?" >> $O/bench.err
python $R/bench.py --no-cpu --no-host-leg --no-streaming > $O/bench_nostream.json 2> $O/bench_nostream.err
python $R/bench.py --no-cpu --no-host-leg --no-records-to-host > $O/bench_norec.json 2> $O/bench_norec.err
for pct in 25 30 35 40 45; do for blk in 2 3 4; do
  python $R/bench.py --no-cpu --no-host-leg --steps 10 --scan-split-pct $pct --side-scan-blocks $blk 2>/dev/null > $O/sweep_${pct}_${blk}.json
done; done
ls -la $O
