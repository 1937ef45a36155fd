#!/usr/bin/env python3
"""Dev tool: list the kernels of a -Rpass-analysis=kernel-resource-usage log that spill (or all with -a)."""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
blocks = re.split(r'remark: Function Name: ', txt)[1:]
rows = []
for b in blocks:
    name = b.split()[0]
    g = lambda k: int(re.search(k + r': (\d+)', b).group(1))  # noqa: E731
    rows.append((name, g('VGPRs'), g('AGPRs'), g('VGPRs Spill'), g(r'ScratchSize \[bytes/lane\]'), g(r'Occupancy \[waves/SIMD\]')))
names = subprocess.run(['c++filt'] + [r[0] for r in rows], capture_output=True, text=True).stdout.split('\n')
print(len(rows), 'kernels;', sum(1 for r in rows if r[3] or r[4]), 'with spills / scratch')
for r, n in zip(rows, names):
    if r[3] or r[4] or '-a' in sys.argv:
        print(re.sub(r'\(anonymous namespace\)::|\(.*', '', n), 'vgpr', r[1], 'agpr', r[2], 'spill', r[3], 'scratch', r[4], 'occ', r[5])
