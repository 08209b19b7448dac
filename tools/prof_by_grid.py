#!/usr/bin/env python3
"""Group a rocprofv3 kernel trace by (kernel name, grid size): calls, total, average.  usage: prof_by_grid.py <dir> [pattern] [top]"""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
pat = sys.argv[2] if len(sys.argv) > 2 else ''
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if pat in r['Kernel_Name']:
        agg[(r['Kernel_Name'][:60], r['Grid_Size_X'], r.get('LDS_Block_Size', ''))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot = sum(sum(v) for v in agg.values())
print('total us %.1f' % tot)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:top]:
    print('%-62s grid %9s lds %6s n %4d tot %9.1f avg %8.1f' % (k[0], k[1], k[2], len(v), sum(v), sum(v) / len(v)))
