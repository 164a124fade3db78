#!/bin/bash
# Round 4, GPU call y: the small blob tier at 5 / 6 waves per SIMD with smaller LDS pools (experiment builds:
# occ1 = 5 waves, PIX 3072, BM 128; occ2 = 6 waves, PIX 2560, BM 96; occ3 = 5 waves, PIX 2560, BM 96).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4y
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() {  # name, lib, args
  local name=$1 lib=$2; shift 2
  MPE_LIB=$lib timeout 200 python $R/bench.py --no-cpu --no-host-leg --steps 15 --warmup 5 --no-false-hint-leg "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernel_ms']; o = d['config']['blob_tier_overflow']
print('$name', round(d['ms_per_step'], 3), round(d['ms_per_step_median'], 3), 'vote_in_region', d['roofline'].get('avg_launch_ms'), 'blobs', round(k['blobs'], 3), 'vote', round(k['vote'], 3), 'tail', round(k['tail'], 3), 'overflow', o['frames'], 'iso_blobs', round(d['kernel_ms_isolated']['blobs'], 3))" >> $O/ab.log 2>&1
}
L=$R/rpg_monocular_pose_estimator_amd/libmpe_hip.so
for rep in 1 2; do
  run base $L
  run occ1 $R/build_variants/libmpe_hip_occ1.so
  run occ2 $R/build_variants/libmpe_hip_occ2.so
  run occ3 $R/build_variants/libmpe_hip_occ3.so
done
cat $O/ab.log
