#!/bin/bash
# Round 4, GPU call t: where does the strict re-evaluation cost the step 6 % (vote_arith 1 vs 2)?  Timing-only experiment
# builds: the block's suspect list is never moved to the launch's list (-DK2_EXP_NO_SUS_FLUSH) / the launch's list is
# never worked off (-DK2_EXP_NO_FIXUP).  Wrong results on purpose.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4t
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() {  # name, lib, args
  local name=$1 lib=$2; shift 2
  MPE_LIB=$lib timeout 200 python $R/bench.py --no-cpu --no-host-leg --steps 15 --warmup 5 --no-false-hint-leg "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernel_ms']
print('$name', round(d['ms_per_step'], 3), round(d['ms_per_step_median'], 3), 'vote_in_region', d['roofline'].get('avg_launch_ms'), 'blobs', round(k['blobs'], 3), 'vote', round(k['vote'], 3), 'tail', round(k['tail'], 3))" >> $O/ab.log 2>&1
}
L=$R/rpg_monocular_pose_estimator_amd/libmpe_hip.so
for rep in 1 2; do
  run default $L
  run arith2 $L --vote-arith 2
  run no_sus_flush $R/build_variants/libmpe_hip_K2_EXP_NO_SUS_FLUSH.so
  run no_fixup $R/build_variants/libmpe_hip_K2_EXP_NO_FIXUP.so
done
cat $O/ab.log
