#!/usr/bin/env python3
"""Summarise one TTA step out of a rocprofv3 --kernel-trace CSV (per kernel name and grid)."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
adam = [i for i, n in enumerate(names) if n.startswith('adam_multi_kernel')]
if len(adam) < 2:          # MSG_CHN 1layer since round 5: Adam runs inside the weight gradient's reduction -- that launch ends a step
    adam = [i for i, n in enumerate(names) if 'gwgrad_mfma_reduce_kernel' in n]      # one launch per step
step = rows[adam[-2] + 1:adam[-1] + 1]
t0, t1 = int(step[0]['Start_Timestamp']), int(step[-1]['End_Timestamp'])
ksum = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in step) / 1e3
print('step wall us %.1f  kernels %d  sum kernel us %.1f' % ((t1 - t0) / 1e3, len(step), ksum))
agg = collections.OrderedDict()
for r in step:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    agg.setdefault((r['Kernel_Name'][:70], r['Grid_Size_X']), []).append(d)
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:top]:
    print('%-72s grid %8s n %3d tot %8.1f avg %7.1f max %7.1f' % (k[0], k[1], len(v), sum(v), sum(v) / len(v), max(v)))
