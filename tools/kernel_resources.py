"""Table of the compiler's per-kernel resource usage for the last build of librtg.so (registers, scratch, occupancy), by kernel
family:  python tools/kernel_resources.py > profiles/rNN_kernel_resources.txt
The numbers are the `-Rpass-analysis=kernel-resource-usage` remarks build.py keeps next to every object (csrc/<unit>.o.res);
build.py itself refuses to link when any kernel touches scratch memory (build.check_no_scratch)."""
import collections
import os
import re
import subprocess
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'transtacos-retunegan_amd'))
import build  # noqa: E402


def main():
    rows = build.kernel_resources()
    names = subprocess.run(['c++filt'] + [r['name'] for r in rows], capture_output=True, text=True).stdout.split('\n')
    fam = collections.OrderedDict()
    for r, n in zip(rows, names):
        n = re.sub(r'\(anonymous namespace\)::', '', n)
        n = re.sub(r'^void ', '', n)
        r['pretty'] = re.sub(r'\(.*$', '', n)
        f = re.sub(r'<.*$', '', r['pretty'])
        fam.setdefault((re.sub(r'(_t\d+|_m\d+|_io\d+|_bf)$', '', r['file']), f), []).append(r)
    total = len(rows)
    spilled = [r for r in rows if r.get('scratch', 0) > 0]
    print(f'# librtg.so: {total} kernels in {len(set(r["file"] for r in rows))} translation units; {len(spilled)} use scratch memory')
    print('# build.py refuses to link when any kernel uses scratch memory (build.check_no_scratch)')
    print(f'# {"unit":14s} {"kernel family":28s} {"instances":>9s} {"VGPRs":>9s} {"AGPRs":>6s} {"occupancy":>9s} {"scratch":>8s}')
    for (unit, f), rs in fam.items():
        v = [r.get('vgpr', 0) for r in rs]
        occ = [r.get('occupancy', 0) for r in rs]
        print(f'  {unit:14s} {f:28s} {len(rs):9d} {min(v):4d}-{max(v):<4d} {max(r.get("agpr", 0) for r in rs):6d} '
              f'{min(occ):4d}-{max(occ):<4d} {sum(1 for r in rs if r.get("scratch", 0) > 0):8d}')
    if spilled:
        print('# kernels with scratch (bytes per lane, spilled VGPRs):')
        for r in spilled:
            print(f'  {r["file"]:18s} {r["scratch"]:5d} B {r.get("vgpr_spill", 0):4d}  {r["pretty"]}')
    print('# dense-layer conv kernel instances (rtg_dconv_kernel.h: dc_built) by translation unit:')
    for u in sorted(set(r['file'] for r in rows if 'dconv_kernel' in r['pretty'])):
        print(f'  {u:18s} {sum(1 for r in rows if r["file"] == u and "dconv_kernel" in r["pretty"])}')
    print(f'  total              {sum(1 for r in rows if "dconv_kernel" in r["pretty"])}')


if __name__ == '__main__':
    main()
