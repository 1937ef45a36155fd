"""dev: time the thin-shape launches of the train step through the C ABI"""
import os, sys, ctypes as C
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, 'transtacos-retunegan_amd')); sys.path.insert(0, os.path.join(REPO, 'tests'))
import numpy as np, torch
import packref
from rtg.lib import lib, Conv1dDesc, current_stream_ptr
from rtg.ops import _desc
CASES = [  # name, B, Cin, Cout, L, K, s, d, p, pre, mask
    ('G conv_post fwd', 32, 32, 1, 8192, 7, 1, 1, 3, 1, 0),
    ('G conv_post dgrad', 32, 1, 32, 8192, 7, 1, 1, 3, 0, 1),
    ('G conv_pre fwd', 32, 1, 16, 8192, 7, 1, 1, 3, 0, 0),
    ('MSD0 conv0 fwd', 64, 1, 32, 8192, 15, 1, 1, 7, 0, 0),
    ('MSD1 conv0 fwd', 64, 1, 32, 4096, 15, 1, 1, 7, 0, 0),
    ('MSD0 conv0 dgrad', 32, 32, 1, 8192, 15, 1, 1, 7, 0, 0),
    ('MSD0 conv_post fwd', 64, 512, 1, 128, 3, 1, 1, 1, 1, 0),
    ('MSD2 conv_post fwd', 64, 512, 1, 32, 3, 1, 1, 1, 1, 0),
    ('MSD0 conv_post dgrad', 64, 1, 512, 128, 3, 1, 1, 1, 0, 1),
    ('MPD0 conv0 fwd', 192, 1, 32, 2731, 5, 3, 1, 2, 0, 0),
    ('MPD3 conv0 fwd', 704, 1, 32, 745, 5, 3, 1, 2, 0, 0),
    ('MPD0 conv_post fwd', 192, 512, 1, 34, 3, 1, 1, 1, 1, 0),
    ('MPD3 conv_post fwd', 704, 512, 1, 10, 3, 1, 1, 1, 1, 0),
    ('MPD0 conv_post dgrad', 192, 1, 512, 34, 3, 1, 1, 1, 0, 1),
    ('MPD3 conv_post dgrad', 704, 1, 512, 10, 3, 1, 1, 1, 0, 1),
]
dev = 'cuda'
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
for name, B, Cin, Cout, L, K, s, d, pad, pre, msk in CASES:
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(B, Cin, L, generator=gen)
    w = torch.randn(Cout, Cin, K, generator=gen) / np.sqrt(Cin * K)
    TM = 32 if Cout >= 32 else 16
    wp = torch.from_numpy(packref.pack_logical(packref.logical_fwd(w.numpy(), 1), TM)).to(dev)
    Lo = (L + 2 * pad - d * (K - 1) - 1) // s + 1
    bias = torch.randn(Cout, generator=gen).to(dev)
    mask = torch.randn(B, Cout, Lo, generator=gen).to(dev) if msk else None
    out = torch.empty(B, Cout, Lo, device=dev)
    xd = x.to(dev)
    desc = _desc(B=B, C1=Cin, L_in=L, Cg=Cin, Mg=Cout, K=K, stride=s, dil=d, pad=pad, Q=Lo, out_C=Cout, out_L=Lo,
                 pre_mode=1 if pre else 0, pre_slope=0.15, mask_slope=0.15, tile_m=TM)
    var = lib.rtg_conv1d_variant(C.byref(desc))
    def run():
        return lib.rtg_conv1d(C.byref(desc), p(xd), None, None, p(wp), p(bias), p(mask), None, p(out), None, current_stream_ptr())
    assert run() == 0
    torch.cuda.synchronize()
    ref = torch.nn.functional.conv1d(torch.nn.functional.leaky_relu(x, 0.15) if pre else x, w, bias.cpu(), s, pad, d)
    if msk: ref = ref * torch.where(mask.cpu() > 0, 1.0, 0.15)
    err = (out.cpu() - ref).abs().max().item()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5): run()
    e0.record()
    for _ in range(50): run()
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) * 1000 / 50
    nbytes = 4 * (x.numel() + out.numel() + (mask.numel() if msk else 0))
    print(f'{name:24s} variant {var} {us:8.1f} us  {nbytes / us / 1e3:8.1f} GB/s  maxerr {err:.2e}')
