#!/usr/bin/env python3
"""Timeline of one TTA step out of a rocprofv3 --kernel-trace CSV: busy time (union of kernel intervals), overlap
between queues, idle gaps and the kernels around the largest ones.  Works on graph-replay traces too."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
adam = [i for i, n in enumerate(names) if n.startswith('adam_multi_kernel')]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
step = rows[adam[-back - 1] + 1:adam[-back] + 1]
iv = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:60], r['Queue_Id']) for r in step]
t0, t1 = iv[0][0], max(e for _, e, _, _ in iv)
ksum = sum(e - s for s, e, _, _ in iv) / 1e3
busy = 0; cur_s, cur_e = iv[0][0], iv[0][1]
gaps = []
last = iv[0]
for s, e, n, q in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append(((s - cur_e) / 1e3, last[2], n))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    if e >= cur_e: last = (s, e, n, q)
busy += cur_e - cur_s
print('step wall us %.1f  kernels %d  sum kernel us %.1f  busy (union) us %.1f  idle us %.1f  overlapped us %.1f  queues %s'
      % ((t1 - t0) / 1e3, len(iv), ksum, busy / 1e3, (t1 - t0 - busy) / 1e3, ksum - busy / 1e3, sorted(set(q for _, _, _, q in iv))))
print('gaps: n %d  mean %.2f us  total %.1f us' % (len(gaps), sum(g[0] for g in gaps) / max(1, len(gaps)), sum(g[0] for g in gaps)))
for g in sorted(gaps, key=lambda g: -g[0])[:12]:
    print('  %.1f us  after %-60s before %s' % g)
