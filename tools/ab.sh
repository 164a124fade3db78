#!/bin/bash
# One parametrised A/B runner (replaces the per-experiment collect_round*_?.sh scripts of rounds 3 - 5):
#   tools/ab.sh OUTDIR REPS "LIB1 LIB2 ..." "bench args of case 1" ["bench args of case 2" ...]
# runs bench.py (quick flags) for every case x library, REPS times, interleaved, and appends one summary line per run
# to gpurun_out/OUTDIR/out.txt.  LIB = a path to a libmpe_hip.so, or "base" for the tree's own.
# A case may start with ENV:NAME=VALUE,NAME=VALUE to set environment variables for it.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$1; REPS=$2; LIBS=$3; shift 3
mkdir -p $O
cd $R
QB="--no-cpu --no-host-leg --no-false-hint-leg --headline-only"
for rep in $(seq 1 $REPS); do
  for c in "$@"; do
    for L in $LIBS; do
      envs=""; args="$c"
      if [[ "$c" == ENV:* ]]; then envs=$(echo "${c#ENV:}" | cut -d' ' -f1 | tr ',' ' '); args=$(echo "$c" | cut -d' ' -f2-); fi
      lib=${L%%@*}; lenv=""; [[ "$L" == *@* ]] && lenv=$(echo "${L#*@}" | tr ',' ' ')   # LIB@VAR=VAL,VAR=VAL
      [ "$lib" = base ] && lib=$R/rpg_monocular_pose_estimator_amd/libmpe_hip.so
      env $envs $lenv MPE_LIB=$lib timeout 300 python bench.py $QB $args 2>$O/last.err | tail -1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read().strip().splitlines()[-1])
    k = d.get('kernel_ms', {}); r = d.get('roofline', {}); p = d.get('parity', {}); v = d.get('vote_arith', {})
    print('$L | $c |', round(d['value']), 'fps', round(d['ms_per_step'], 3), 'ms med', round(d.get('ms_per_step_median', 0), 3),
          '| blobs', round(k.get('blobs', 0), 3), 'vote', round(k.get('vote', 0), 3), 'tail', round(k.get('tail', 0), 3),
          '| frac', r.get('frac'), '| parity', p.get('status_mismatches'), p.get('mismatches_unexplained'),
          '| fix', v.get('hypotheses_re_evaluated_strictly_per_step'), 'relost', v.get('frames_voted_again'))
except Exception as e:
    print('$L | $c | FAILED', e)
" >> $O/out.txt
    done
  done
done
cat $O/out.txt
