#!/usr/bin/env python3
"""dev: MSD convs.4 (512 -> 512, k41, stride 4, 64 groups of 8 x 8) forward at the three scales: the vector-ALU kernel
(rtg_gconv.hip) against the position-pair matrix-core kernel (rtg_gmfma.hip, gmfma_pair_kernel)."""
import ctypes as C
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, 'transtacos-retunegan_amd'))
import torch  # noqa: E402
from rtg import tune  # noqa: E402
from rtg import lib as L  # noqa: E402
from rtg.lib import lib, GconvDesc  # noqa: E402

P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
B, g, K, s, pad = 64, 64, 41, 4, 20
tune.REPS = 20
for Lin in (512, 256, 128):
    Lo = (Lin + 2 * pad - (K - 1) - 1) // s + 1
    d = GconvDesc(B, g, 8, 8, K, s, pad, Lin, Lo, 0.15)
    x = torch.randn(B, 512, Lin, device='cuda')
    v = torch.randn(512, 8, K, device='cuda') * 0.2
    scale = torch.cat([torch.ones(512, device='cuda'), torch.ones(512, device='cuda')])
    bias = torch.randn(512, device='cuda')
    out = torch.empty(B, 512, Lo, device='cuda')
    fl = 2.0 * B * 512 * Lo * 8 * K
    res = []
    # vector-ALU kernel: weights [group][ci][tap][oc]
    n_g = lib.rtg_gconv_workspace(C.byref(d))
    wg = torch.zeros(n_g + 256, device='cuda')
    jobg = L.PackJob(0, 0, 0, 512 * 8 * K, L.PACK_GCONV_FWD, g, 8, 8, K, K, 8, 1, 16, 0, 0, 0, 0)
    blocks, lds = L.assign_pack_blocks([jobg])
    tab = torch.frombuffer(bytearray(bytes(jobg)), dtype=torch.uint8).cuda()
    assert lib.rtg_weights_pack(P(tab), 1, blocks, lds, P(v.flatten()), P(scale), P(wg), None) == 0
    t = tune._time(lambda: lib.rtg_gconv_forward(C.byref(d), P(x), P(wg), P(bias), P(out), None))
    ref = out.clone()
    res.append(f'gconv {t * 1e3:6.1f} us {fl / t / 1e9:5.1f} TF')
    n_w = lib.rtg_gmfma_workspace(C.byref(d))
    wm = torch.zeros(n_w + 64, device='cuda')
    job = L.PackJob(0, 0, 0, n_w, L.PACK_GMFMA_FWD, g, 8, 8, K, K, 8, 48, 16, s, 0, 0, 0)
    blocks, lds = L.assign_pack_blocks([job])
    tab2 = torch.frombuffer(bytearray(bytes(job)), dtype=torch.uint8).cuda()
    assert lib.rtg_weights_pack(P(tab2), 1, blocks, lds, P(v.flatten()), P(scale), P(wm), None) == 0
    t = tune._time(lambda: lib.rtg_gmfma_forward(C.byref(d), P(x), P(wm), P(bias), P(out), None))
    err = (out - ref).abs().max().item()
    res.append(f'pair  {t * 1e3:6.1f} us {fl / t / 1e9:5.1f} TF  max diff {err:.2e}')
    print(f'L_in {Lin:4d} L_out {Lo:4d}: ' + ' | '.join(res), flush=True)
