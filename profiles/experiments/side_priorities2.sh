#!/bin/bash
O=gpurun_out/r5f; mkdir -p $O
python -c "
import ctypes
h=ctypes.CDLL('/opt/rocm/lib/libamdhip64.so'); a=ctypes.c_int(); b=ctypes.c_int(); print('priority range (least, greatest):', h.hipDeviceGetStreamPriorityRange(ctypes.byref(a), ctypes.byref(b)), a.value, b.value)" >> $O/out.txt 2>&1
Q="--no-cpu --no-host-leg --no-false-hint-leg --headline-only --steps 15 --warmup 4"
run() {
  python bench.py $Q "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms']['per_sub_batch'][1:7]; s=d['roofline']['timed_region_by_slot']; print('$*', round(d['ms_per_step'],3), round(d['ms_per_step_median'],3), 'gap0', round(s[0]['gap_before_ms'],3), 'gaps', round(sum(x['gap_before_ms'] for x in s[1:])/7,3), 'launch', round(sum(x['launch_ms'] for x in s)/8,3), 'tail', round(sum(x['tail'] for x in k)/6,3))" >> $O/out.txt
}
for rep in 1 2; do
run
run --opt tail_priority=2 --opt scan_priority=2
run --opt tail_priority=-1 --opt scan_priority=-1
run --opt tail_priority=1 --opt scan_priority=-1
run --opt tail_priority=-1 --opt scan_priority=1
run --opt tail_priority=1 --opt scan_priority=1
run --opt tail_priority=2 --opt scan_priority=1
done
echo GPU_MAX_HW_QUEUES=8 >> $O/out.txt
export GPU_MAX_HW_QUEUES=8
run
run --opt tail_priority=-1 --opt scan_priority=1
cat $O/out.txt
