"""dev: time the fused ResidualStack launches (forward / backward) for the two served shapes"""
import os, sys, ctypes as C
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, 'transtacos-retunegan_amd'))
import torch
from rtg import lib as L
from rtg.lib import lib
for Cc, Lx in ((128, 32), (64, 256), (32, 2048)):
    B = 32
    d = L.ResStackDesc(B, Cc, Lx, (C.c_int * 6)(1, 1, 3, 1, 9, 1), 0.01, 1, 0.15)
    x = torch.randn(B, Cc, Lx, device='cuda')
    n_frag = (Cc // 32) * (Cc // 16) * 3 * 8 * 64
    wps = [torch.randn(n_frag, device='cuda') * 0.05 for _ in range(6)]
    bs = [torch.randn(Cc, device='cuda') for _ in range(6)]
    outs = [torch.empty_like(x) for _ in range(6)]
    wp = L.PtrArray6(*[t.data_ptr() for t in wps]); bp = L.PtrArray6(*[t.data_ptr() for t in bs]); op = L.PtrArray6(*[t.data_ptr() for t in outs])
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    fwd = lambda: lib.rtg_resstack_forward(C.byref(d), C.c_void_p(x.data_ptr()), C.byref(wp), C.byref(bp), C.byref(op), st)
    masks = L.PtrArray6(*[t.data_ptr() for t in outs])
    gouts = [torch.empty_like(x) for _ in range(6)]; gp = L.PtrArray6(*[t.data_ptr() for t in gouts])
    bwd = lambda: lib.rtg_resstack_backward(C.byref(d), C.c_void_p(x.data_ptr()), C.c_void_p(outs[5].data_ptr()), C.byref(wp), C.byref(masks), C.byref(gp), st)
    for name, fn in (('fwd', fwd), ('bwd', bwd)):
        assert fn() == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3): fn()
        e0.record()
        for _ in range(20): fn()
        e1.record(); e1.synchronize()
        print(f'C{Cc} L{Lx} {name}: {e0.elapsed_time(e1) * 50:7.1f} us')
