import ctypes as C, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, 'transtacos-retunegan_amd')); sys.path.insert(0, os.path.join(REPO, 'tests'))
import numpy as np, torch, torch.nn.functional as F
import packref
from rtg.lib import lib, Conv1dDesc
B, Cin, Cout, H, W = 2, 2, 32, 65, 35
kh, kw, sh, sw, ph, pw = 3, 3, 2, 1, 1, 1
g = torch.Generator().manual_seed(0)
x = torch.randn(B, Cin, H, W, generator=g)
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def run(w, bias):
    ref = F.conv2d(F.leaky_relu(x, 0.15).double(), w.double(), bias.double(), (sh, sw), (ph, pw)).float()
    Ho, Wo = ref.shape[-2:]
    wp = torch.from_numpy(packref.pack_logical(w.numpy().reshape(1, Cout, Cin * kh, kw), 32)).cuda()
    d = Conv1dDesc(B=B * Ho, C1=Cin * kh, C2=0, L_in=W, groups=1, Cg=Cin * kh, Mg=Cout, K=kw, stride=sw, dil=1, pad=pw, Q=Wo,
                   out_C=Cout, out_L=Wo, shuf_S=1, shuf_P=0, pre_mode=1, pre_slope=0.15, mask_slope=1.0, out_scale=1.0, act=0,
                   act_slope=1.0, accumulate=0, tile_m=32, out_split=0, h_in=H, h_k=kh, h_stride=sh, h_pad=ph, h_n=Ho, h_mode=0)
    out = torch.full((B, Cout, Ho, Wo), float('nan'), device='cuda')
    rc = lib.rtg_conv1d(C.byref(d), P(x.cuda()), None, None, P(wp), P(bias.cuda()), None, None, P(out), None, st)
    torch.cuda.synchronize()
    return out.cpu(), ref
w0 = torch.zeros(Cout, Cin, kh, kw); b0 = torch.arange(Cout).float()
o, r = run(w0, b0); print('bias only: err', (o - r).abs().max().item(), o[0, :4, 0, 0], o[1, 31, 32, 34])
for (co, ci, a, b_) in ((0, 0, 1, 1), (5, 1, 0, 2), (31, 0, 2, 0)):
    w1 = torch.zeros(Cout, Cin, kh, kw); w1[co, ci, a, b_] = 1.0
    o, r = run(w1, torch.zeros(Cout))
    e = (o - r).abs()
    print('delta', (co, ci, a, b_), 'err', e.max().item(), 'nonzero channels', (o.abs().amax(dim=(0, 2, 3)) > 0).nonzero().flatten().tolist()[:8],
          'ref nz', (r.abs().amax(dim=(0, 2, 3)) > 0).nonzero().flatten().tolist())
    print('   out', o[0, co, 3, 3:7], 'ref', r[0, co, 3, 3:7])
w1 = torch.zeros(Cout, Cin, kh, kw); w1[0, 0, 1, 1] = 1.0
o, r = run(w1, torch.zeros(Cout))
e = (o - r).abs()
idx = (e == e.max()).nonzero()[0].tolist()
print('worst at', idx, o[tuple(idx)].item(), r[tuple(idx)].item(), 'rows with err', (e.amax(dim=(0, 1, 3)) > 1e-4).nonzero().flatten().tolist()[:10],
      'cols', (e.amax(dim=(0, 1, 2)) > 1e-4).nonzero().flatten().tolist()[:10], 'items', (e.amax(dim=(1, 2, 3)) > 1e-4).nonzero().flatten().tolist())
