#!/bin/bash
# round 3, fourth GPU call: follow-up blob tiers beside the voting kernel, mono8 decode, sub-batch size sweep.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3d
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R && timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
cd /tmp
for p in 4 6 8 16; do for pct in 25 30; do
  python $R/bench.py --no-cpu --no-host-leg --steps 12 --pipeline $p --scan-split-pct $pct 2>/dev/null > $O/sweep_p${p}_${pct}.json
done; done
python $R/bench.py --no-cpu --no-host-leg --pipeline 8 --no-records-to-host 2>/dev/null > $O/bench_p8_norec.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python $R/bench.py --no-cpu --no-host-leg --steps 4 --warmup 2 --pipeline 8 --frames 131072 > $O/trace.log 2>&1
python3 - <<'PY'
import csv,glob,json,os
O=os.environ.get('O','/tmp')
PY
ls -la $O
