#!/bin/bash
# round 3, third GPU call: timeline of the streaming step (kernel trace), sub-batch size with streaming, C3 with the
# register-resident prefilter.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3c
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R && timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
cd /tmp
python $R/bench.py --no-cpu --no-host-leg > $O/bench.json 2> $O/bench.err
python $R/bench.py --no-cpu --no-host-leg --pipeline 8 > $O/bench_p8.json 2>/dev/null
python $R/bench.py --no-cpu --no-host-leg --pipeline 8 --scan-split-pct 25 > $O/bench_p8_25.json 2>/dev/null
python $R/bench.py --no-cpu --no-host-leg --pipeline 8 --scan-split-pct 35 > $O/bench_p8_35.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python $R/bench.py --no-cpu --no-host-leg --steps 4 --warmup 2 > $O/trace.log 2>&1
python $R/bench.py --no-cpu --no-host-leg --steps 5 --config C3 --frames 65536 2>/dev/null > $O/bench_C3.json
SQ="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE"
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/pmc3_sq -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu --no-host-leg --frames 16384 --pipeline 1 --config C3 > $O/pmc3_sq.log 2>&1
python $R/profiles/summarize_pmc_clock.py $O/pmc3_sq $O/pmc3_sq_summary.csv
find $O/pmc3_sq -name "*.csv" -delete
find $O/trace -name "*agent_info.csv" -delete
ls -la $O $O/trace/*
