#!/bin/bash
# tools/leg_trace.sh NAME bench-args...: rocprofv3 --kernel-trace --stats of one bench.py leg, the kernel table printed
# and left in gpurun_out/NAME/ (run on the GPU box: gpurun -- 'tools/leg_trace.sh salt --clutter salt --frames 16384')
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
name=$1; shift
O=$R/gpurun_out/$name; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o s -- python $R/bench.py --no-cpu --no-host-leg \
  --no-false-hint-leg --headline-only "$@" > $O/run.log 2>&1
python - "$O" <<PY
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
if not f: sys.exit("no kernel_stats.csv under " + sys.argv[1])
for r in list(csv.DictReader(open(f[0])))[:16]:
    print("%-70s calls %6s total %12s ns avg %10s ns %6s %%" % (r["Name"][:70], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]))
PY
tail -1 $O/run.log | cut -c1-160
