#!/bin/bash
mkdir -p gpurun_out; L=gpurun_out/dot4.log; : > $L
timeout -s KILL 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 >> $L
for extra in "--pipeline 1 --frames 16384 --steps 20" "--steps 10" "--steps 10"; do
timeout -s KILL 200 python bench.py --no-cpu --no-host-leg $extra --warmup 3 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d.get('kernel_ms'); k.pop('per_sub_batch',None); print({a:round(b,4) for a,b in k.items()}, round(d['value']), d['ms_per_step'])
" >> $L
done
cat $L
