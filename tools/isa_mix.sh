#!/bin/bash
# static instruction mix of kernels of one TU: tools/isa_mix.sh mpe_k2 'k2_voteILb0ELb0ELi3' 'k2_voteILb1ELb0ELi0' [-- extra flags]
R=$(cd "$(dirname "$0")/.." && pwd)
F=$1; shift
SYMS=(); FLAGS=()
while [ $# -gt 0 ]; do if [ "$1" = "--" ]; then shift; FLAGS=("$@"); break; fi; SYMS+=("$1"); shift; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off --cuda-device-only -S "${FLAGS[@]}" -I$R/include -I$R/rpg_monocular_pose_estimator_amd/csrc $R/rpg_monocular_pose_estimator_amd/csrc/$F.hip -o /tmp/$F.s
python3 $R/profiles/experiments/isa_histogram.py /tmp/$F.s "${SYMS[@]}" | python3 -c "
import sys, json
d = json.load(sys.stdin)
for k, v in d.items():
    c = v.get('counts', v)
    print(k[:60], {x: c[x] for x in c if c[x]} if isinstance(c, dict) else c, v.get('resources'))
"
