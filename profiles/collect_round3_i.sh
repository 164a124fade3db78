#!/bin/bash
# round 3: C3 with LDS-resident table slices (after the pointer fix); every step under its own timeout
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3i
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R && timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
cd /tmp
timeout 120 python $R/bench.py --no-cpu --no-host-leg --steps 20 --warmup 5 > $O/bench.json 2>/dev/null
timeout 200 python $R/bench.py --no-cpu --no-host-leg --steps 5 --config C3 --frames 65536 2>/dev/null > $O/bench_C3.json
SQ="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE"
timeout 200 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/pmc3_sq -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu --no-host-leg --frames 16384 --pipeline 1 --config C3 > $O/pmc3_sq.log 2>&1
timeout 60 python $R/profiles/summarize_pmc_clock.py $O/pmc3_sq $O/pmc3_sq_summary.csv
find $O/pmc3_sq -name "*.csv" -delete
ls $O
