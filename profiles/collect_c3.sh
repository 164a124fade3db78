R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c3a; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
ARGS="--config C3 --frames 16384 --pipeline 1 --steps 3 --warmup 1 --no-cpu"
python $R/bench.py $ARGS > $O/bench.json 2>$O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python $R/bench.py $ARGS > $O/stats.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_INSTS_SALU --output-format csv -d $O/pmc_sq -o p -- python $R/bench.py $ARGS > $O/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM --output-format csv -d $O/pmc_sq2 -o p -- python $R/bench.py $ARGS > $O/pmc_sq2.log 2>&1
for n in pmc_sq pmc_sq2; do f=$(find $O/$n -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python $R/profiles/summarize_pmc.py $f $O/${n}_summary.csv; done
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete
ls $O; tail -3 $O/pmc_sq2.log
