#!/bin/bash
# Round 4, third GPU call: after the SGPR diet of the voting kernel and with the cell-sum contour phase in the blob
# kernel — the GPU suite, a same-box A/B against the round-3 tree (ab_r3/, commit 14e6180, built here), kernel stats and
# one SQ counter pass of the fused kernel, the vote soaks.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4c
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R && timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
cd /tmp
for i in 1 2; do
(cd $R/ab_r3 && timeout 300 python bench.py --no-cpu --no-host-leg --steps 20 --warmup 5 2>/dev/null > $O/bench_r3_$i.json)
timeout 300 python $R/bench.py --no-cpu --no-host-leg --steps 20 --warmup 5 --vote-arith 2 --no-false-hint-leg 2>/dev/null > $O/bench_arith2_$i.json
timeout 300 python $R/bench.py --no-cpu --no-host-leg --steps 20 --warmup 5 --no-false-hint-leg 2>/dev/null > $O/bench_arith1_$i.json
done
timeout 300 python $R/bench.py --no-cpu --no-host-leg --steps 5 --config C3 --frames 65536 2>$O/bench_C3.err > $O/bench_C3.json
timeout 300 python $R/bench.py --no-cpu --no-host-leg --steps 5 --config C3 --frames 65536 --vote-arith 2 2>/dev/null > $O/bench_C3_arith2.json
(cd $R/ab_r3 && timeout 300 python bench.py --no-cpu --no-host-leg --steps 5 --config C3 --frames 65536 2>/dev/null > $O/bench_C3_r3.json)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python $R/bench.py --no-cpu --no-host-leg --steps 20 --warmup 5 --no-false-hint-leg > $O/stats.log 2>&1
SQ="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE"
timeout 300 rocprofv3 --kernel-trace --kernel-include-regex mpe:: --pmc $SQ --output-format csv -d $O/pmc_sq -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu --no-host-leg --no-false-hint-leg --frames 65536 > $O/pmc_sq.log 2>&1
timeout 60 python $R/profiles/summarize_pmc_clock.py $O/pmc_sq $O/pmc_sq_summary.csv
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
cd $R
timeout 900 python tests/soak_votes.py 65536 C2 gpurun_out/r4c/soak_votes_C2 > $O/soak_votes_C2.log 2>&1; echo "rc $?" >> $O/soak_votes_C2.log
timeout 900 python tests/soak_votes.py 4096 C3 gpurun_out/r4c/soak_votes_C3 > $O/soak_votes_C3.log 2>&1; echo "rc $?" >> $O/soak_votes_C3.log
ls $O
