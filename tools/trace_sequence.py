#!/usr/bin/env python3
"""One TTA step out of a rocprofv3 --kernel-trace CSV as the ordered launch sequence: start offset, duration, gap to the previous
kernel's end (negative = overlap with another stream), kernel name, grid.  Works for graph replay and kernel-by-kernel launches."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
adam = [i for i, n in enumerate(names) if n.startswith('adam_multi_kernel')]
if len(adam) < 2:          # MSG_CHN 1layer since round 5: Adam runs inside the weight gradient's reduction -- that launch ends a step
    adam = [i for i, n in enumerate(names) if 'gwgrad_mfma_reduce_kernel' in n]
# which step: argv[2] = index into the list of Adam launches (default: the last one -- in bench.py that is the kernel-by-kernel
# profiling leg; e.g. 8 = a replayed step of the first timed block)
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(adam) - 1
step = rows[adam[k - 1] + 1:adam[k] + 1]
t0 = int(step[0]['Start_Timestamp'])
prev_end = t0
tot = 0.0
for r in step:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    tot += (e - s) / 1e3
    print('%8.1f  dur %7.1f  gap %6.1f  %-64s grid %8s wg %4s' % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r['Kernel_Name'][:64], r['Grid_Size_X'], r['Workgroup_Size_X']))
    prev_end = max(prev_end, e)
print('step wall us %.1f  kernels %d  sum kernel us %.1f' % ((prev_end - t0) / 1e3, len(step), tot))
