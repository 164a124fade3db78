#!/bin/bash
# Experiment (round 5): k2_vote_relost as 64 one-wave blocks instead of 32 four-wave blocks (it waits 0.4 - 0.5 ms for
# four-wave slots beside the voting launch, in every tail chain, to read two words).
O=gpurun_out/r5n; mkdir -p $O
Q="--no-cpu --no-host-leg --no-false-hint-leg --headline-only --steps 20 --warmup 5"
run() {
  local lib=$1; shift
  MPE_LIB=$lib python bench.py $Q "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['timed_region_by_slot']; k=d['kernel_ms']['per_sub_batch'][1:7]; print('$(basename $lib) $*', round(d['ms_per_step'],3), round(d['ms_per_step_median'],3), 'gaps', round(sum(x['gap_before_ms'] for x in s)/8,3), 'launch', round(sum(x['launch_ms'] for x in s)/8,3), 'tail', round(sum(x['tail'] for x in k)/6,3))" >> $O/out.txt
}
B=rpg_monocular_pose_estimator_amd/libmpe_hip.so
V=build_variants/libmpe_hip_r64.so
for rep in 1 2 3 4; do run $B; run $V; done
MPE_LIB=$V timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "full_suspect" 2>&1 | tail -1 >> $O/out.txt
cat $O/out.txt
