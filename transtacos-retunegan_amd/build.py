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

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'librtg.so')
STAMP = os.path.join(HERE, 'csrc', '.build_stamp')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-comment', '-Wno-unused-result'] + os.environ.get('RTG_EXTRA_FLAGS', '').split()
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
    all_dig = hashlib.sha256('\n'.join(_obj_digest(s, hd) for s in srcs).encode()).hexdigest()
    if not todo and os.path.exists(LIB) and os.path.exists(STAMP) and open(STAMP).read().strip() == all_dig:
        return LIB
    running = []

    def reap(block_until):
        while len(running) > block_until:
            s, o, st, dig, p = running.pop(0)
            out, _ = p.communicate()
            if p.returncode != 0:
                sys.stderr.write(out.decode())
                for r in running:
                    r[4].kill()
                raise RuntimeError(f'hipcc failed on {s}')
            open(st, 'w').write(dig)

    for s, o, st, dig in todo:
        reap(JOBS - 1)
        cmd = [hipcc, *FLAGS, '-c', s, '-o', o]
        if verbose:
            print(' '.join(cmd), flush=True)
        if os.path.exists(st):
            os.remove(st)
        running.append((s, o, st, dig, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    reap(0)
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB, *objs]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    open(STAMP, 'w').write(all_dig)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print('built', LIB)
