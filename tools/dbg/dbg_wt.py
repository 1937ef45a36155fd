import os, sys, ctypes as C
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, 'transtacos-retunegan_amd'))
import torch, torch.nn.functional as F
from rtg.lib import lib, WgradDesc, check
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
for (B, Cin, Cout, L, K, s, d, pad) in ((2, 1, 16, 1024, 7, 1, 1, 3), (2, 32, 1, 2048, 7, 1, 1, 3)):
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(B, Cin, L, generator=gen)
    w = torch.randn(Cout, Cin, K, generator=gen, dtype=torch.float64, requires_grad=True)
    bias = torch.zeros(Cout, dtype=torch.float64, requires_grad=True)
    y = F.conv1d(x.double(), w, bias, s, pad, d)
    dy = torch.randn(y.shape, generator=gen)
    y.backward(dy.double())
    Lo = y.shape[-1]
    for cfg in (0, 7):
        wd = WgradDesc(B=B, C1=Cin, C2=0, L_in=L, groups=1, Cg=Cin, Mg=Cout, K=K, stride=s, dil=d, pad=pad, Q=Lo, dy_L=Lo,
                       pre_mode=0, pre_slope=1.0, gy_mode=0, gy_slope=1.0, gy_scale=1.0, splits=1, part_stride=0, shape_cfg=cfg)
        splits = lib.rtg_wgrad_splits(C.byref(wd))
        need = Cout * (Cin * K + 1)
        part = torch.full((splits * need,), float('nan'), device='cuda')
        wd.splits, wd.part_stride = splits, need
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        xd, dyd = x.cuda(), dy.cuda()
        check(lib.rtg_conv1d_wgrad(C.byref(wd), p(xd), None, p(dyd), None, p(part), st))
        torch.cuda.synchronize()
        tot = part.view(splits, need).double().sum(0).cpu()
        dW = tot[:Cout * Cin * K].view(Cout, Cin, K); db = tot[Cout * Cin * K:]
        print(cfg, 'splits', splits, 'dW err', (dW - w.grad).abs().max().item(), 'ref max', w.grad.abs().max().item(), 'db err', (db - bias.grad).abs().max().item(), 'nan', torch.isnan(tot).sum().item())
        if cfg == 7: print(dW.flatten()[:8], w.grad.flatten()[:8])
