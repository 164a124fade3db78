#!/bin/bash
# round 3, fourth GPU call: follow-up blob tiers beside the voting kernel, mono8 decode, sub-batch size sweep.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3e
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R && timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
cd /tmp
for p in 4 6 8 16; do for pct in 25 30; do
  python $R/bench.py --no-cpu --no-host-leg --steps 12 --pipeline $p --scan-split-pct $pct 2>/dev/null > $O/sweep_p${p}_${pct}.json
done; done
python $R/bench.py --no-cpu --no-host-leg --pipeline 8 --no-records-to-host 2>/dev/null > $O/bench_p8_norec.json
ls -la $O
