"""Build librtg.so (all HIP kernels + the C ABI of include/rtg.h) for gfx950 with hipcc, in-tree.

    python transtacos-retunegan_amd/build.py [--force]

hipcc cross-compiles without a GPU; the .so sits next to this file so that it travels with the source tree.  Objects are
rebuilt only when their source, a header or the flags changed (a per-object digest next to each .o), at most JOBS
compilations at a time.
"""
import glob
import hashlib
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'librtg.so')
STAMP = os.path.join(HERE, 'csrc', '.build_stamp')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-comment', '-Wno-unused-result',
         '-Rpass-analysis=kernel-resource-usage'] + os.environ.get('RTG_EXTRA_FLAGS', '').split()
# No kernel of the library may touch scratch memory (a spilled instance is a slower duplicate of a block shape that fits: round 5
# shipped 70 of them): the compiler's per-kernel resource remarks are kept next to each object (<obj>.res), summarised by
# kernel_resources() (tools/kernel_resources.py writes the table under profiles/), and check_no_scratch() fails the build.
JOBS = max(1, min(8, os.cpu_count() or 1))
# ablation / diagnostic hooks of the kernel headers (-DRTG_EXP_*, -DRTG_STAMPS) change what the kernels compute or write:
# they belong to the side libraries of tools/dev_build.sh (librtg_dev*.so, loaded through RTG_DEV_LIB), never to librtg.so
if any(f.startswith(('-DRTG_EXP_', '-DRTG_STAMPS')) for f in FLAGS):
    raise SystemExit('build.py: RTG_EXTRA_FLAGS carries an ablation / diagnostic define; build those with tools/dev_build.sh')


def _headers_digest():
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(CSRC, '*.h')) + [os.path.join(HERE, '..', 'include', 'rtg.h')]):
        h.update(os.path.basename(f).encode())
        h.update(open(f, 'rb').read())
    h.update(' '.join(FLAGS).encode())
    return h


def _obj_digest(src, hd):
    h = hd.copy()
    h.update(open(src, 'rb').read())
    return h.hexdigest()


# rough compile cost of the heavy translation units (seconds on this image), for the build order
_COST = {'rtg_dconv': 60, 'rtg_dconv_bf': 60, 'rtg_dconv_io': 40, 'rtg_dwgrad': 60, 'rtg_conv1d_t': 50, 'rtg_wgrad_m': 40, 'rtg_gconv': 30,
         'rtg_gmfma': 30}


def _without_remarks(text):
    """compiler output minus the kernel-resource remarks (with the source excerpts and include traces printed around them)"""
    keep, inc, skip = [], [], 0
    for ln in text.splitlines(keepends=True):
        if '[-Rpass-analysis=kernel-resource-usage]' in ln:
            skip, inc = (2 if 'Function Name' in ln else 0), []
            continue
        if skip and (ln.lstrip()[:1].isdigit() or ln.lstrip().startswith('|')):
            skip -= 1
            continue
        skip = 0
        if ln.startswith('In file included from'):
            inc.append(ln)
            continue
        keep += inc + [ln]
        inc = []
    return ''.join(keep)


def kernel_resources():
    """[{file, name (mangled), sgpr, vgpr, agpr, scratch, occupancy, vgpr_spill, sgpr_spill}] of every kernel of the last build"""
    import re
    rows = []
    for res in sorted(glob.glob(os.path.join(CSRC, '*.o.res'))):
        cur = None
        for ln in open(res, errors='replace'):
            m = re.search(r'remark: Function Name: (\S+)', ln)
            if m:
                cur = {'file': os.path.basename(res)[:-6], 'name': m.group(1)}
                rows.append(cur)
                continue
            m = re.search(r'remark:\s+([A-Za-z \[\]/]+): (\S+) \[-Rpass', ln)
            if m and cur is not None:
                key = {'TotalSGPRs': 'sgpr', 'VGPRs': 'vgpr', 'AGPRs': 'agpr', 'ScratchSize [bytes/lane]': 'scratch',
                       'Occupancy [waves/SIMD]': 'occupancy', 'VGPRs Spill': 'vgpr_spill', 'SGPRs Spill': 'sgpr_spill',
                       'LDS Size [bytes/block]': 'lds'}.get(m.group(1).strip())
                if key:
                    cur[key] = int(m.group(2))
    return rows


def check_no_scratch():
    bad = [r for r in kernel_resources() if r.get('scratch', 0) > 0]
    if bad:
        raise RuntimeError('kernels that use scratch memory (prune the instance or fix its registers):\n' +
                           '\n'.join(f"  {r['file']}: {r['name']}: {r['scratch']} B/lane, {r.get('vgpr_spill', 0)} VGPRs spilled" for r in bad))


def build(force=False, verbose=True):
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    srcs = sorted(glob.glob(os.path.join(CSRC, '*.hip')))
    hd = _headers_digest()
    objs, todo = [], []
    for s in srcs:
        o = os.path.join(CSRC, os.path.basename(s)[:-4] + '.o')
        objs.append(o)
        dig = _obj_digest(s, hd)
        st = o + '.stamp'
        if force or not os.path.exists(o) or not os.path.exists(st) or open(st).read().strip() != dig:
            todo.append((s, o, st, dig))
    # stale objects of sources that no longer exist must not be linked
    for o in glob.glob(os.path.join(CSRC, '*.o')):
        if o not in objs:
            os.remove(o)
            for ext in ('.res', '.stamp'):
                if os.path.exists(o + ext):
                    os.remove(o + ext)
    all_dig = hashlib.sha256('\n'.join(_obj_digest(s, hd) for s in srcs).encode()).hexdigest()
    if not todo and os.path.exists(LIB) and os.path.exists(STAMP) and open(STAMP).read().strip() == all_dig:
        return LIB
    running = []

    def reap(block_until):
        # whichever compilation finishes first frees its slot (output goes to a file: the resource remarks of a
        # many-instance translation unit are megabytes, a pipe nobody drains would stall the compiler)
        while len(running) > block_until:
            done = [r for r in running if r[4].poll() is not None]
            if not done:
                time.sleep(0.2)
                continue
            for r in done:
                running.remove(r)
                s, o, st, dig, p, t0, log = r
                text = open(log, errors='replace').read()
                os.remove(log)
                if verbose:
                    print(f'  {os.path.basename(s)}: {time.time() - t0:.0f} s', flush=True)
                if p.returncode != 0:
                    sys.stderr.write(_without_remarks(text))
                    for q in running:
                        q[4].kill()
                    raise RuntimeError(f'hipcc failed on {s}')
                open(o + '.res', 'w').write(text)
                rest = _without_remarks(text)
                if rest.strip() and verbose:
                    sys.stderr.write(rest)
                open(st, 'w').write(dig)

    # the translation units with the most instances first: the tail of the build is then made of short ones
    todo.sort(key=lambda t: -_COST.get(os.path.basename(t[0])[:-4].rstrip('0123456789'), 1))
    for s, o, st, dig in todo:
        reap(JOBS - 1)
        cmd = [hipcc, *FLAGS, '-c', s, '-o', o]
        if verbose:
            print(' '.join(cmd), flush=True)
        if os.path.exists(st):
            os.remove(st)
        log = o + '.log'
        running.append((s, o, st, dig, subprocess.Popen(cmd, stdout=open(log, 'w'), stderr=subprocess.STDOUT), time.time(), log))
    reap(0)
    check_no_scratch()
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB, *objs]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    open(STAMP, 'w').write(all_dig)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print('built', LIB)
