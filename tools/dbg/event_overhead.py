"""dev: what a HIP event pair adds to a bracketed launch (bench.py's per-launch timing)"""
import os, sys, ctypes as C
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, 'transtacos-retunegan_amd'))
import torch
from rtg.lib import lib, current_stream_ptr
x = torch.zeros(4, device='cuda'); big = torch.zeros(1 << 24, device='cuda')
p = lambda t: C.c_void_p(t.data_ptr())
def bracket(fn, n=200):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    torch.cuda.synchronize()
    for e0, e1 in ev:
        e0.record(); fn(); e1.record()
    torch.cuda.synchronize()
    ts = sorted(e0.elapsed_time(e1) * 1000 for e0, e1 in ev)
    return ts[len(ts) // 2], ts[len(ts) // 10], ts[-len(ts) // 10]
print('empty bracket            median/p10/p90 us', bracket(lambda: None))
print('1-element axpby          median/p10/p90 us', bracket(lambda: lib.rtg_axpby(p(x), None, p(x), 1, 1.0, 0.0, 0, current_stream_ptr())))
print('64 MB axpby (~25 us)     median/p10/p90 us', bracket(lambda: lib.rtg_axpby(p(big), None, p(big), big.numel(), 1.0, 0.0, 0, current_stream_ptr())))
# the same with a busy queue ahead (GPU never idles: launches queued back to back)
def bracket_busy(fn, n=200):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    torch.cuda.synchronize()
    for _ in range(20): lib.rtg_axpby(p(big), None, p(big), big.numel(), 1.0, 0.0, 0, current_stream_ptr())
    for e0, e1 in ev:
        e0.record(); fn(); e1.record()
    torch.cuda.synchronize()
    ts = sorted(e0.elapsed_time(e1) * 1000 for e0, e1 in ev)
    return ts[len(ts) // 2], ts[len(ts) // 10], ts[-len(ts) // 10]
print('busy: empty bracket      ', bracket_busy(lambda: None))
print('busy: 1-element axpby    ', bracket_busy(lambda: lib.rtg_axpby(p(x), None, p(x), 1, 1.0, 0.0, 0, current_stream_ptr())))
print('busy: 64 MB axpby        ', bracket_busy(lambda: lib.rtg_axpby(p(big), None, p(big), big.numel(), 1.0, 0.0, 0, current_stream_ptr())))
