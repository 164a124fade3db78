#!/bin/bash
# Collects everything under profiles/round2_* in ONE gpurun call on a 1xMI355X box:
#   gpurun --timeout 2400 -- 'bash profiles/collect_round2.sh'
# Raw output goes to gpurun_out/final2/; profiles/install_round2.py condenses it into the committed files.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final2
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $O/bench.json 2> $O/bench.err
for c in C1 C3 C4; do python $R/bench.py --no-cpu --no-host-leg --steps 5 --config $c --frames $([ $c = C4 ] && echo 16384 || echo 65536) 2>/dev/null > $O/bench_$c.json; done
python $R/bench.py --no-cpu --no-host-leg --steps 10 --pipeline-mode 3 2>/dev/null > $O/bench_mode3.json
python $R/bench_streams.py --streams 1 --frames 400 2>/dev/null | tail -1 > $O/streams1.json
python $R/bench_streams.py --streams 8 --frames 400 2>/dev/null | tail -1 > $O/streams8.json
for n in 8 64 128 256; do python $R/bench_streams.py --streams $n --frames 300 --lockstep 2>/dev/null | tail -1 > $O/lockstep$n.json; done
python $R/bench_streams.py --streams 64 --frames 300 --lockstep --groups 2 2>/dev/null | tail -1 > $O/lockstep64g2.json
python $R/bench_streams.py --streams 256 --frames 300 --lockstep --groups 4 2>/dev/null | tail -1 > $O/lockstep256g4.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python $R/bench.py --no-cpu --no-host-leg > $O/stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_seq -o s -- python $R/bench.py --no-cpu --no-host-leg --pipeline 1 --frames 16384 --steps 20 --warmup 3 > $O/stats_seq.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c3 -o s -- python $R/bench.py --no-cpu --no-host-leg --config C3 --pipeline 1 --frames 16384 --steps 3 --warmup 1 > $O/stats_c3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_lockstep -o s -- python $R/bench_streams.py --streams 64 --frames 300 --lockstep > $O/stats_lockstep.log 2>&1
SQ="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_INSTS_SALU"
ARGS="--steps 3 --warmup 1 --no-cpu --no-host-leg --frames 32768 --pipeline 2"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- python $R/bench.py $ARGS > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- python $R/bench.py $ARGS > $O/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/pmc_sq -o p -- python $R/bench.py $ARGS > $O/pmc_sq.log 2>&1
# the stand-alone scan kernel and the plain voting kernel at ONE launch shape (16384 frames, one launch per step)
ARGS1="--steps 3 --warmup 1 --no-cpu --no-host-leg --frames 16384 --pipeline 1"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc1_fetch -o p -- python $R/bench.py $ARGS1 > $O/pmc1_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc1_write -o p -- python $R/bench.py $ARGS1 > $O/pmc1_write.log 2>&1
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/pmc1_sq -o p -- python $R/bench.py $ARGS1 > $O/pmc1_sq.log 2>&1
# C3 (8 markers / 12 detections): the plain voting kernel, one launch shape
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/pmc3_sq -o p -- python $R/bench.py $ARGS1 --config C3 > $O/pmc3_sq.log 2>&1
for n in pmc_fetch pmc_write pmc_sq pmc1_fetch pmc1_write pmc1_sq pmc3_sq; do
  f=$(find $O/$n -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $R/profiles/summarize_pmc.py $f $O/${n}_summary.csv
done
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
# end-to-end parity soaks against the oracle, both voting arithmetics
MPE_VOTE_ARITH=1 python $R/tests/soak_parity.py 1048576 C2 2>/dev/null | tail -1 > $O/soak_fast.json
MPE_VOTE_ARITH=0 python $R/tests/soak_parity.py 262144 C2 2>/dev/null | tail -1 > $O/soak_strict.json
MPE_VOTE_ARITH=1 python $R/tests/soak_parity.py 2048 C3 2048 2>/dev/null | tail -1 > $O/soak_fast_c3.json
python $R/tests/soak_votes.py 131072 C2 2>/dev/null | tail -1 > $O/soak_votes.json
ls $O
