#!/usr/bin/env python3
"""Dev tool: per-kernel instruction statistics of a hipcc -S --cuda-device-only listing (register copies, full waits ...)."""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else 'dconv_kernel'
parts = re.split(r'\n(_Z[A-Za-z0-9_]+):\s*; @', txt)
rows = []
for i in range(1, len(parts), 2):
    name, body = parts[i], parts[i + 1].split('s_endpgm')[0]
    if pat not in name:
        continue
    c = lambda k: len(re.findall(k, body))  # noqa: E731
    rows.append((name, c(r'\n'), c('v_mfma'), c('v_mov_b64'), c(r'v_mov_b32'), c(r'vmcnt\(0\)'), c('s_cbranch'), c('scratch_')))
names = subprocess.run(['c++filt'] + [r[0] for r in rows], capture_output=True, text=True).stdout.split('\n')
for r, n in zip(rows, names):
    n = re.sub(r'\(anonymous namespace\)::|void |\(.*', '', n)
    print(f'{n:60s} lines {r[1]:6d} mfma {r[2]:4d} mov64 {r[3]:4d} mov32 {r[4]:4d} vmcnt0 {r[5]:3d} branches {r[6]:4d} scratch {r[7]:3d}')
