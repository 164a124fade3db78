#!/bin/bash
# the bench line once more, now that profiles/round3_pmc.json (same source fingerprint) is in the tree: `traffic` filled in
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final3
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 python $R/bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?" >> $O/bench.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python $R/bench.py --no-cpu --no-host-leg --steps 20 --warmup 5 > $O/stats.log 2>&1
find $O -name "*kernel_trace.csv" -delete
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_seq -o s -- python $R/bench.py --no-cpu --no-host-leg --pipeline 1 --frames 16384 --steps 20 --warmup 3 > $O/stats_seq.log 2>&1
find $O -name "*kernel_trace.csv" -delete
