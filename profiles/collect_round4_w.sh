#!/bin/bash
# Round 4, last GPU call: what the driver does at round end, on the committed tree — GPU suite, smoke(), the default bench.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4w
mkdir -p $O
cd $R && timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/smoke.log; tail -2 $O/smoke.log
cd /tmp && export TMPDIR=/tmp
timeout 600 python $R/bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"; tail -c 600 $O/bench_default.json | head -c 10; python -c "
import json; d = json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['traffic_source'][-12:], d['cpu_baseline']['value'])"
