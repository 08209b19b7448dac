#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counter_collection.csv files (one or more passes)."""
import collections
import csv
import glob
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for pat in sys.argv[1:]:
    for f in glob.glob(pat):
        for r in csv.DictReader(open(f)):
            key = (r['Kernel_Name'][:64], r.get('Grid_Size', r.get('Grid_Size_X', '')))
            agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
rows = []
for key, cs in agg.items():
    rows.append((key, {c: sum(v) / len(v) for c, v in cs.items()}, max(len(v) for v in cs.values())))
rows.sort(key=lambda t: -t[1].get('SQ_BUSY_CYCLES', t[1].get('FETCH_SIZE', 0)))
for key, c, n in rows[:int(40)]:
    print('%-66s grid %8s n %3d' % (key[0], key[1], n))
    print('    ' + '  '.join('%s=%.4g' % (k, v) for k, v in sorted(c.items())))
