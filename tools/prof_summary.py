#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace result database (rocpd sqlite) of bench.py into a per-kernel CSV restricted to
the steady-state train steps (the block-shape tuner's trial launches of the first step are left out).

    python tools/prof_summary.py gpurun_out/<dir>/prof/r_results.db profiles/<name>_kernel_stats.csv [--skip-steps 3]

A step ends at the last `adamw_kernel` launch of the generator update; steps are found from the gaps between adamw
launches."""
import csv
import sqlite3
import sys


def main():
    db, out = sys.argv[1], sys.argv[2]
    skip = int(sys.argv[sys.argv.index('--skip-steps') + 1]) if '--skip-steps' in sys.argv else 3
    c = sqlite3.connect(db)
    rows = c.execute('select name, start, end from kernels order by start').fetchall()
    # step boundaries: an adamw launch followed by a long run (> 250 launches) of non-adamw launches starts the G
    # forward of the NEXT step only after the generator's update, i.e. after the adamw that follows the longest gap
    ad = [i for i, r in enumerate(rows) if 'adamw' in r[0]]
    gaps = [(ad[i + 1] - ad[i], ad[i + 1]) for i in range(len(ad) - 1)]
    big = max(g for g, _ in gaps[len(gaps) // 2:])            # the G-update backward (longest launch run of a step)
    ends = [e for g, e in gaps if g >= big - 4]               # index of the adamw that closes each G update
    # the generator's optimizer may issue more than one launch: extend to the last adamw of the cluster
    ends = [max(a for a in ad if a - e < 8 and a >= e) for e in ends]
    steps = [(ends[i] + 1, ends[i + 1] + 1) for i in range(len(ends) - 1)]
    steps = steps[skip - 1:] if skip >= 1 else steps
    agg = {}
    wall = busy = 0.0
    for s, e in steps:
        seg = rows[s:e]
        wall += max(r[2] for r in seg) - min(r[1] for r in seg)
        cur_s = cur_e = None
        for n, a, b in seg:
            v = agg.setdefault(n, [0, 0.0, 1e30, 0.0])
            v[0] += 1
            v[1] += b - a
            v[2] = min(v[2], b - a)
            v[3] = max(v[3], b - a)
            if cur_e is None or a > cur_e:
                if cur_e is not None:
                    busy += cur_e - cur_s
                cur_s, cur_e = a, b
            else:
                cur_e = max(cur_e, b)
        busy += cur_e - cur_s
    n = len(steps)
    tot = sum(v[1] for v in agg.values())
    with open(out, 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['# steady-state steps', n, 'wall_ms_per_step', round(wall / n / 1e6, 3), 'gpu_busy_ms_per_step',
                    round(busy / n / 1e6, 3), 'sum_kernel_ms_per_step', round(tot / n / 1e6, 3), 'launches_per_step',
                    sum(v[0] for v in agg.values()) // n])
        w.writerow(['Name', 'Calls', 'CallsPerStep', 'TotalDurationNs', 'AverageNs', 'MinNs', 'MaxNs', 'Percentage'])
        for name, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            w.writerow([name, v[0], round(v[0] / n, 2), int(v[1]), round(v[1] / v[0], 1), int(v[2]), int(v[3]),
                        round(100 * v[1] / tot, 2)])
    print(open(out).readline().strip())


if __name__ == '__main__':
    main()
