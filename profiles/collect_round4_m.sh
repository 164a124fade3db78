#!/bin/bash
# Round 4, thirteenth GPU call: same-box A/B of the step, alternating: this tree / its kernels of the commit before
# (49db044: double-precision scan-carrying voting loop, wide island windows) / this tree with the wide island windows /
# the round-3 tree.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4m
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() {  # name, lib
  MPE_LIB=$2 timeout 200 python $R/bench.py --no-cpu --no-host-leg --steps 15 --warmup 5 --no-false-hint-leg 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernel_ms']
print('$1', round(d['ms_per_step'], 3), round(d['ms_per_step_median'], 3), 'vote_in_region', d['roofline'].get('avg_launch_ms'), 'blobs', round(k['blobs'], 3), 'vote', round(k['vote'], 3), 'tail', round(k['tail'], 3))" >> $O/ab.log 2>&1
}
for rep in 1 2; do
  run new $R/rpg_monocular_pose_estimator_amd/libmpe_hip.so
  run k $R/build_variants/libmpe_hip_k.so
  run wide $R/build_variants/libmpe_hip_wide.so
  (cd $R/ab_r3 && timeout 200 python bench.py --no-cpu --no-host-leg --steps 15 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernel_ms']
print('r3', round(d['ms_per_step'], 3), round(d['ms_per_step_median'], 3), 'vote_in_region', d['roofline'].get('avg_launch_ms'), 'blobs', round(k['blobs'], 3), 'vote', round(k['vote'], 3), 'tail', round(k['tail'], 3))" >> $O/ab.log 2>&1)
done
cat $O/ab.log
