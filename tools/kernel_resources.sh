#!/bin/bash
# Resource usage of every kernel of one translation unit (VGPRs, SGPRs, scratch, LDS, occupancy), one line each:
#   tools/kernel_resources.sh mpe_k2 [extra hipcc flags]
R=$(cd "$(dirname "$0")/.." && pwd)
F=$1; shift
cd $R/rpg_monocular_pose_estimator_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off "$@" -Rpass-analysis=kernel-resource-usage -c $F.hip -o /dev/null 2>&1 |
python3 -c "
import sys, re, subprocess
cur = {}
def flush():
    if cur:
        name = subprocess.run(['c++filt', cur.get('name', '?')], capture_output=True, text=True).stdout.strip()
        name = re.sub(r'\(.*', '', name)
        print('%-44s VGPR %3s AGPR %3s SGPR %3s scratch %4s B  LDS %6s B  occupancy %s' % (name[:44], cur.get('VGPRs'), cur.get('AGPRs'), cur.get('SGPRs'), cur.get('ScratchSize [bytes/lane]'), cur.get('LDS Size [bytes/block]'), cur.get('Occupancy [waves/SIMD]')))
for line in sys.stdin:
    m = re.search(r'remark: [^:]*:\d+:\d+: +(.*?): (.*?) \[-Rpass', line) or re.search(r'remark: +(.*?): (.*?) \[-Rpass', line)
    if not m: continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == 'Function Name':
        flush(); cur = {'name': v}
    else:
        cur[k] = v
flush()
"
