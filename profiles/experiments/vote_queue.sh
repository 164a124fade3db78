#!/bin/bash
# Experiment (round 5): the scan-carrying voting kernel's per-wave queue of deferred exact votes at 60 instead of 28
# entries (worked off with 52 - 60 lanes busy instead of 20 - 28) — for frames with many detections (clutter legs).
O=gpurun_out/r5q; mkdir -p $O
run() {
  local lib=$1; shift
  MPE_LIB=$lib python bench.py --no-host-leg --no-false-hint-leg --headline-only --cpu-sample 256 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$(basename $lib) $*', round(d['value']), round(d['ms_per_step'],3), 'vote', round(d['kernel_ms']['vote'],3), 'tail', round(d['kernel_ms']['tail'],3), 'parity', d['parity']['status_mismatches'], d['parity']['mismatches_unexplained'], 'fix', round(d['vote_arith']['hypotheses_re_evaluated_strictly_per_step']), d['vote_arith']['frames_voted_again'])" >> $O/out.txt
}
B=rpg_monocular_pose_estimator_amd/libmpe_hip.so
V=build_variants/libmpe_hip_vq60.so
for rep in 1 2; do
for L in $B $V; do
run $L --clutter d16 --frames 16384 --steps 4 --warmup 1
run $L --clutter d4 --frames 32768 --steps 5 --warmup 2
run $L --steps 15 --warmup 4
done
done
cat $O/out.txt
