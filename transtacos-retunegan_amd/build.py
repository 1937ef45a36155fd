"""Build librtg.so (all HIP kernels + the C ABI of include/rtg.h) for gfx950 with hipcc, in-tree.

    python transtacos-retunegan_amd/build.py [--force]

hipcc cross-compiles without a GPU; the .so sits next to this file so that it travels with the source tree.
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
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-comment', '-Wno-unused-result']


def _digest():
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(CSRC, '*.hip')) + glob.glob(os.path.join(CSRC, '*.h')) +
                   [os.path.join(HERE, '..', 'include', 'rtg.h')])
    for f in files:
        h.update(f.encode())
        h.update(open(f, 'rb').read())
    h.update(' '.join(FLAGS).encode())
    return h.hexdigest()


def build(force=False, verbose=True):
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(STAMP) and open(STAMP).read().strip() == dig:
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objs = []
    srcs = sorted(glob.glob(os.path.join(CSRC, '*.hip')))
    procs = []
    for s in srcs:
        o = os.path.join(CSRC, os.path.basename(s)[:-4] + '.o')
        objs.append(o)
        cmd = [hipcc, *FLAGS, '-c', s, '-o', o]
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError(f'hipcc failed on {s}')
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB, *objs]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    open(STAMP, 'w').write(dig)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print('built', LIB)
