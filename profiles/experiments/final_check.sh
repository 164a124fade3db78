#!/bin/bash
# Round 5, last GPU call: the bench line of the committed tree once more (bench.py gained two read-only fields after the
# collection), the bench tests of the GPU suite, and the headline leg three times on ONE box (run-to-run spread).
#   gpurun --timeout 420 -- 'bash profiles/experiments/final_check.sh'
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final_check
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
QB="--no-cpu --no-host-leg --no-false-hint-leg --headline-only"
for i in 1 2 3; do
  timeout 100 python $R/bench.py $QB --steps 20 --warmup 5 2>/dev/null > $O/headline_$i.json
done
( time timeout 200 python $R/bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?" >> $O/bench.err ) 2>> $O/bench.err
cd $R && timeout 150 python -m pytest tests -m gpu -q -k "bench" > $O/pytest_bench.log 2>&1; echo "pytest rc $?" >> $O/pytest_bench.log
ls $O
