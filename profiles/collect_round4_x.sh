#!/bin/bash
# Round 4, GPU call x (blur: the dilated columns compute only their r outputs next to the bright ones):  the blob kernel's contour phase (experiment builds with -DK1B_STOP_AFTER=4 / 41 .. 44 /
# 5), kernels back to back, 65 536 C2 frames; and the step with the new side-scan defaults.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4x
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for n in 3 4 full; do
  LIB=$R/build_variants/libmpe_hip_stop$n.so
  [ $n = full ] && LIB=$R/rpg_monocular_pose_estimator_amd/libmpe_hip.so
  MPE_LIB=$LIB timeout 200 python $R/bench.py --no-cpu --no-host-leg --pipeline 1 --frames 65536 --steps 10 --warmup 3 2>$O/err_$n.log | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$n', json.dumps(d.get('kernel_ms')), d.get('ms_per_step'))" >> $O/phases.log 2>&1
done
timeout 300 python $R/bench.py --no-cpu --no-host-leg --steps 20 --warmup 5 --no-false-hint-leg 2>/dev/null > $O/bench_C2.json
cd $R && timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
cd /tmp
timeout 200 python $R/bench_streams.py --streams 1 --frames 400 2>/dev/null > $O/streams1.json
timeout 200 python $R/bench_streams.py --streams 64 --frames 300 --lockstep 2>/dev/null > $O/lockstep64.json
cat $O/phases.log; python -c "
import json; d = json.loads(open('$O/bench_C2.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['ms_per_step_median'], d['config'].get('scan_split_pct'), d['config'].get('side_scan_blocks'))"
