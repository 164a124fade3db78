#!/bin/bash
# Collects everything under profiles/ in ONE gpurun call on a 1xMI355X box:
#   gpurun --timeout 1800 -- 'bash profiles/collect_round1.sh'
# Raw output goes to gpurun_out/final/; profiles/install_round1.py condenses it into the committed files.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --host-frames > $O/bench.json 2> $O/bench.err
for c in C1 C3 C4; do python $R/bench.py --no-cpu --steps 5 --config $c --frames $([ $c = C4 ] && echo 16384 || echo 65536) 2>/dev/null > $O/bench_$c.json; done
python $R/bench_streams.py --streams 1 --frames 400 2>/dev/null | tail -1 > $O/streams1.json
python $R/bench_streams.py --streams 8 --frames 400 --no-cpu 2>/dev/null | tail -1 > $O/streams8.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python $R/bench.py --no-cpu > $O/stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_seq -o s -- python $R/bench.py --no-cpu --pipeline 1 --frames 16384 --steps 20 --warmup 3 > $O/stats_seq.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_two_stream -o s -- python $R/bench.py --no-cpu --pipeline-mode 0 > $O/stats_two_stream.log 2>&1
ARGS="--steps 3 --warmup 1 --no-cpu --frames 32768 --pipeline 2"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- python $R/bench.py $ARGS > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- python $R/bench.py $ARGS > $O/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_INSTS_SALU --output-format csv -d $O/pmc_sq -o p -- python $R/bench.py $ARGS > $O/pmc_sq.log 2>&1
# the stand-alone scan kernel and the plain voting kernel at ONE launch shape (16384 frames, one launch per step)
ARGS1="--steps 3 --warmup 1 --no-cpu --frames 16384 --pipeline 1"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc1_fetch -o p -- python $R/bench.py $ARGS1 > $O/pmc1_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc1_write -o p -- python $R/bench.py $ARGS1 > $O/pmc1_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_INSTS_SALU --output-format csv -d $O/pmc1_sq -o p -- python $R/bench.py $ARGS1 > $O/pmc1_sq.log 2>&1
for n in pmc_fetch pmc_write pmc_sq pmc1_fetch pmc1_write pmc1_sq; do
  f=$(find $O/$n -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $R/profiles/summarize_pmc.py $f $O/${n}_summary.csv
done
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
ls $O
