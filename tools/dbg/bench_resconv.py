"""dev: the resconv kernel against the general kernel's best block shape on the ResBlock3 layers of config 2"""
import os, sys, ctypes as C
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, 'transtacos-retunegan_amd')); sys.path.insert(0, os.path.join(REPO, 'tests'))
import numpy as np, torch
import packref
from rtg.lib import lib, Conv1dDesc, current_stream_ptr
from rtg.ops import _desc
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
dev = 'cuda'
for B, Cc, L in ((32, 32, 8192), (32, 64, 2048)):
    for K, dil in ((3, 9), (3, 1), (5, 3), (7, 9)):
        gen = torch.Generator().manual_seed(1)
        x = torch.randn(B, Cc, L, generator=gen).to(dev)
        w = torch.randn(Cc, Cc, K, generator=gen) / np.sqrt(Cc * K)
        wp = torch.from_numpy(packref.pack_logical(packref.logical_fwd(w.numpy(), 1), 32)).to(dev)
        bias = torch.randn(Cc, generator=gen).to(dev)
        out = torch.empty_like(x)
        pad = (K * dil - dil) // 2
        desc = _desc(B=B, C1=Cc, L_in=L, Cg=Cc, Mg=Cc, K=K, dil=dil, pad=pad, Q=L, out_C=Cc, out_L=L, pre_mode=1, pre_slope=0.15, tile_m=32)
        cands = (C.c_int * 32)()
        n = lib.rtg_conv1d_tile_candidates(C.byref(desc), cands, 32)
        res = {}
        for c in cands[:n]:
            desc.tile_cfg = c
            run = lambda: lib.rtg_conv1d(C.byref(desc), p(x), None, None, p(wp), p(bias), None, p(x), p(out), None, current_stream_ptr())
            if run() != 0:
                continue
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(3): run()
            e0.record()
            for _ in range(20): run()
            e1.record(); e1.synchronize()
            res[c] = e0.elapsed_time(e1) * 1000 / 20
        flop = 2.0 * B * L * Cc * Cc * K
        old = min((v, k) for k, v in res.items() if k < 7000)
        new = sorted((v, k) for k, v in res.items() if k > 7000)
        print(f'B{B} C{Cc} L{L} k{K} d{dil}: general best {old[0]:7.1f} us (cfg {old[1]}, {flop/old[0]/1e6:5.1f} TF/s) | resconv ' +
              ', '.join(f'{k}: {v:7.1f} us {flop/v/1e6:5.1f} TF/s' for v, k in new))
