#!/usr/bin/env python3
"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` output: VGPRs, scratch (spills), occupancy and LDS per kernel.
  hipcc ... -c x.hip -Rpass-analysis=kernel-resource-usage 2> log; python tools/kernel_resources.py log [filter]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ''
names, rows = [], []
for b in re.split(r'remark: [^\n]*Function Name: ', txt)[1:]:
    name = b.split('\n')[0].strip()
    g = lambda k: int(re.search(k + r': (\d+)', b).group(1)) if re.search(k + r': (\d+)', b) else -1
    names.append(name)
    rows.append((g('VGPRs'), g('AGPRs'), g(r'ScratchSize \[bytes/lane\]'), g(r'Occupancy \[waves/SIMD\]'), g(r'LDS Size \[bytes/block\]')))
dem = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout.splitlines()
print('vgpr agpr scratch occ lds  kernel')
for r, d in zip(rows, dem):
    if flt in d:
        print('%4d %4d %5d %3d %6d  %s' % (r + (d[:150],)))
