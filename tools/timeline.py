#!/usr/bin/env python3
"""Where one steady-state train step's wall time goes (dev tool): from a rocprofv3 --kernel-trace database of bench.py, for the
LAST complete step: idle time (no kernel running), time with exactly one kernel running (and which kernels: the serial
sections), mean concurrency, and the phases between the optimizer launches.
    python tools/timeline.py <r_results.db> [out.txt]"""
import collections
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    return re.sub(r'\(.*$', '', n)[:70]


def main():
    c = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in c.execute('pragma table_info(kernels)')]
    rows = c.execute('select name, start, end from kernels order by start').fetchall()
    ad = [i for i, r in enumerate(rows) if 'adamw_kernel' in r[0]]
    gaps = [(ad[i + 1] - ad[i], ad[i + 1]) for i in range(len(ad) - 1)]
    big = max(g for g, _ in gaps[len(gaps) // 2:])
    ends = [e for g, e in gaps if g >= big - 4]
    ends = [max(a for a in ad if a - e < 8 and a >= e) for e in ends]
    s, e = ends[-2] + 1, ends[-1] + 1
    seg = rows[s:e]
    t0, t1 = min(r[1] for r in seg), max(r[2] for r in seg)
    out = [f'# columns of the kernels table: {cols}', f'step: {len(seg)} launches, wall {(t1 - t0) / 1e6:.3f} ms']
    ev = []
    for i, (n, a, b) in enumerate(seg):
        ev.append((a, 1, i)); ev.append((b, -1, i))
    ev.sort()
    live = set()
    last = t0
    by_n = collections.Counter()
    solo = collections.Counter()
    solo_n = collections.Counter()
    for t, d, i in ev:
        dt = t - last
        by_n[min(len(live), 6)] += dt
        if len(live) == 1:
            k = short(seg[next(iter(live))][0])
            solo[k] += dt
        last = t
        if d > 0:
            live.add(i)
        else:
            live.discard(i)
    tot = t1 - t0
    # idle gaps: intervals with no kernel in flight, largest first, with the launch that ended before and the one that starts after
    order = sorted(range(len(seg)), key=lambda i: seg[i][1])
    gaps_, end_max, last_i = [], seg[order[0]][2], order[0]
    for i in order[1:]:
        if seg[i][1] > end_max:
            gaps_.append((seg[i][1] - end_max, end_max, last_i, i))
        if seg[i][2] > end_max:
            end_max, last_i = seg[i][2], i
    out.append(f'idle gaps: {len(gaps_)} of {sum(g[0] for g in gaps_) / 1e6:.3f} ms in all; by size: ' +
               ', '.join(f'>= {th} us: {sum(1 for g in gaps_ if g[0] >= th * 1e3)} ({sum(g[0] for g in gaps_ if g[0] >= th * 1e3) / 1e6:.3f} ms)' for th in (1, 2, 4, 8, 16)))
    for g, t, a_, b_ in sorted(gaps_, reverse=True)[:12]:
        out.append(f'  {g / 1e3:6.1f} us at {(t - t0) / 1e6:7.3f} ms  after {short(seg[a_][0])[:40]}  before {short(seg[b_][0])[:40]}')
    out.append('time by number of kernels in flight (6 = six or more):')
    for k in sorted(by_n):
        out.append(f'  {k}: {by_n[k] / 1e6:7.3f} ms  {100 * by_n[k] / tot:5.1f} %')
    out.append('kernels running ALONE (serial sections), by time:')
    for k, v in solo.most_common(40):
        out.append(f'  {v / 1e6:7.3f} ms  {k}')
    # phases: between the adamw launches of the step
    marks = [(i, seg[i][1]) for i in range(len(seg)) if 'adamw_kernel' in seg[i][0]]
    out.append('optimizer launches at (ms from step start): ' + ', '.join(f'{(t - t0) / 1e6:.2f}' for _, t in marks))
    # the launches around the optimizer steps and the step start: start (ms), duration (us), kernels in flight at its start, stream
    import os
    win = float(os.environ.get('TL_WIN_MS', '0.7'))
    centers = [t0 + int(float(x) * 1e6) for x in os.environ.get('TL_AT_MS', '').split(',') if x] or [m[1] for m in marks[1::2]]
    srows = c.execute('select name, start, end, stream_id from kernels order by start').fetchall()[s:e]
    for cen in centers:
        out.append(f'---- launches within {win} ms of {(cen - t0) / 1e6:.2f} ms')
        for n, a, b, sid in srows:
            if abs(a - cen) <= win * 1e6 or abs(b - cen) <= win * 1e6:
                conc = sum(1 for _, a2, b2, _ in srows if a2 <= a < b2)
                out.append(f'  {(a - t0) / 1e6:8.3f} ms {(b - a) / 1e3:7.1f} us  x{conc}  s{sid:<3d} {short(n)}')
    text = '\n'.join(out)
    print(text)
    if len(sys.argv) > 2:
        open(sys.argv[2], 'w').write(text + '\n')


if __name__ == '__main__':
    main()
