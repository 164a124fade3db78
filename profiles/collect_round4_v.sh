#!/bin/bash
# Round 4, GPU call v: the plain voting kernel of <= 5 markers (small batches, stage-level vote entry, tracker
# initialisation) with the single-precision head + deferred exact evaluation as well (k2_defers: np >= 1).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4v
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R && timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
timeout 600 python tests/soak_votes.py 131072 C2 gpurun_out/r4v/soak_votes_C2 > $O/soak_votes_C2.log 2>&1; echo "rc $?" >> $O/soak_votes_C2.log
MPE_SOAK_ORACLE=0 timeout 600 python tests/soak_votes.py 1048576 C2 gpurun_out/r4v/soak_dvs_C2 > $O/soak_dvs_C2.log 2>&1; echo "rc $?" >> $O/soak_dvs_C2.log
MPE_SOAK_ORACLE=0 timeout 600 python tests/soak_votes.py 262144 C1 gpurun_out/r4v/soak_dvs_C1 > $O/soak_dvs_C1.log 2>&1; echo "rc $?" >> $O/soak_dvs_C1.log
cd /tmp
timeout 200 python $R/bench.py --no-cpu --no-host-leg --pipeline 1 --frames 16384 --steps 20 --warmup 3 --no-false-hint-leg 2>/dev/null > $O/bench_seq.json
(cd $R/ab_r3 && timeout 200 python bench.py --no-cpu --no-host-leg --pipeline 1 --frames 16384 --steps 20 --warmup 3 2>/dev/null > $O/bench_seq_r3.json)
timeout 200 python $R/bench_streams.py --streams 1 --frames 400 2>/dev/null > $O/streams1.json
for f in soak_votes_C2 soak_dvs_C2 soak_dvs_C1; do tail -2 $O/$f.log | cut -c1-900; done
python -c "
import json
for n in ('bench_seq', 'bench_seq_r3'):
    d = json.loads(open('$O/' + n + '.json').read().strip().splitlines()[-1]); print(n, d['ms_per_step'], d['kernel_ms'])"
