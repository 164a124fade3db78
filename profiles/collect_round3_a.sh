#!/bin/bash
# round 3, first GPU call: tests, a bench line, the voting kernel's instruction counts and effective clock after the
# instruction-cutting pass, and the parity soaks with forensics.   gpurun --timeout 1500 -- 'bash profiles/collect_round3_a.sh'
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3a
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R && timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
cd /tmp
python $R/bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?" >> $O/bench.err
SQ="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE"
ARGS="--steps 3 --warmup 1 --no-cpu --no-host-leg --frames 32768 --pipeline 2"
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/pmc_sq -o p -- python $R/bench.py $ARGS > $O/pmc_sq.log 2>&1
ARGS1="--steps 3 --warmup 1 --no-cpu --no-host-leg --frames 16384 --pipeline 1"
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/pmc1_sq -o p -- python $R/bench.py $ARGS1 > $O/pmc1_sq.log 2>&1
for n in pmc_sq pmc1_sq; do python $R/profiles/summarize_pmc_clock.py $O/$n $O/${n}_summary.csv; done
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
cd $R
python tests/soak_votes.py 131072 C2 gpurun_out/r3a/soak_votes_C2 2> $O/soak_votes.err | tail -1 > $O/soak_votes.json
MPE_VOTE_ARITH=1 python tests/soak_parity.py 262144 C2 65536 gpurun_out/r3a/soak_parity_C2 2> $O/soak_parity.err | tail -1 > $O/soak_parity.json
ls -la $O
