#!/bin/bash
# tools/leg_pmc.sh NAME "COUNTERS" bench-args...: one rocprofv3 counter pass (--kernel-trace + --pmc only, mpe:: kernels)
# of one bench.py leg; per-kernel means of the counters printed and left in gpurun_out/NAME_summary.csv
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
name=$1; ctr=$2; shift 2
O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --kernel-trace --kernel-include-regex mpe:: --pmc $ctr --output-format csv -d $O/$name -o p -- \
  python $R/bench.py --no-cpu --no-host-leg --no-false-hint-leg --headline-only --no-isolated "$@" > $O/$name.log 2>&1
timeout 60 python $R/profiles/summarize_pmc_clock.py $O/$name $O/${name}_summary.csv
find $O/$name -name "*.csv" -delete
cat $O/${name}_summary.csv
