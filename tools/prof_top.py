#!/usr/bin/env python3
"""Top kernels of a rocprofv3 --kernel-trace --stats --output-format csv run: python tools/prof_top.py <dir> [n]"""
import csv
import glob
import sys

f = sorted(glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True))[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total ms', tot / 1e6)
for r in rows[:n]:
    print('%-64s calls %6s total %9.2f ms avg %9.1f us %5.1f%%' % (r['Name'][:64], r['Calls'], float(r['TotalDurationNs']) / 1e6,
                                                                   float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
