#!/bin/bash
# Experiment (round 5): a blob kernel capped at 96 VGPRs (5 waves per SIMD possible: -DK1B_SMALL_MIN_WAVES=5, pools 3072 /
# 128) leaves room for TWO side-scan waves per SIMD beside four of its own — does a two-block side scan then pay?
O=gpurun_out/r5m; mkdir -p $O
Q="--no-cpu --no-host-leg --no-false-hint-leg --headline-only --steps 15 --warmup 4"
run() {
  local lib=$1; shift
  MPE_LIB=$lib python bench.py $Q "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['timed_region_by_slot']; print('$(basename $lib) $*', round(d['ms_per_step'],3), round(d['ms_per_step_median'],3), 'gaps', round(sum(x['gap_before_ms'] for x in s)/8,3), 'launch', round(sum(x['launch_ms'] for x in s)/8,3), 'overflow', d['blob_tier_overflow']['frames'])" >> $O/out.txt
}
B=rpg_monocular_pose_estimator_amd/libmpe_hip.so
V=build_variants/libmpe_hip_occ1.so
for rep in 1 2; do
run $B
run $V
run $V --side-scan-blocks 2 --scan-split-pct 34
run $V --side-scan-blocks 2 --scan-split-pct 40
run $V --side-scan-blocks 3 --scan-split-pct 44
run $B --side-scan-blocks 2 --scan-split-pct 40
done
cat $O/out.txt
