"""The multi-tensor loss kernels of rtg_elem.hip (rtg_loss_fwd / rtg_loss_bwd: loss.py:51-52,121-122,142,154) through the C
ABI against torch in float64 — every kind, jobs whose length is no multiple of 4 and whose tensors start at 4-, 8- and
12-byte offsets (the 16-byte path, its tail and the 4-byte fallback of a job)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _term(kind, a, b, target):
    from rtg import lib as L
    if kind == L.LOSS_L1:
        return (a - b).abs()
    if kind == L.LOSS_L1_L1LOG:
        return (a - b).abs() + (a.log() - b.log()).abs()
    if kind == L.LOSS_MSE_REL:
        return (target - (a - b)) ** 2
    return (target - a) ** 2


@pytest.mark.parametrize('kind_name', ['LOSS_L1', 'LOSS_L1_L1LOG', 'LOSS_MSE_TARGET', 'LOSS_MSE_REL'])
def test_loss_kernels_against_torch(kind_name):
    from rtg import lib as L
    from rtg.lib import lib
    kind = getattr(L, kind_name)
    gen = torch.Generator().manual_seed(3)
    target = 1.0
    jobs, keep = [], []
    ref_loss = torch.zeros((), dtype=torch.float64)
    refs = []
    for n, off_a, off_b, w in ((4096 * 5 + 3, 0, 0, 0.7), (1001, 1, 1, 1.3), (4 * 777, 2, 3, 0.5), (7, 0, 1, 2.0),
                                (256 * 64 * 4 * 3, 0, 0, 1.0)):
        def mk(off):
            base = torch.rand(n + 8, generator=gen) + 0.25 if kind == L.LOSS_L1_L1LOG else torch.randn(n + 8, generator=gen)
            d = base.cuda()
            return d, d[off:off + n]
        (abase, a), (bbase, b) = mk(off_a), mk(off_b)
        use_b = kind != L.LOSS_MSE_TARGET
        da, db = torch.full((n + 8,), float('nan'), device='cuda'), torch.full((n + 8,), float('nan'), device='cuda')
        dav, dbv = da[off_a:off_a + n], db[off_b:off_b + n]
        keep += [abase, bbase, da, db]
        a64 = a.cpu().double().requires_grad_(True)
        b64 = b.cpu().double().requires_grad_(True)
        term = _term(kind, a64, b64 if use_b else None, target) if use_b else _term(kind, a64, None, target)
        lj = w * term.mean()
        ref_loss = ref_loss + lj.detach()
        lj.backward()
        refs.append((a64.grad, b64.grad if (use_b and b64.grad is not None) else None, dav, dbv, use_b))
        jobs.append(L.LossJob(a.data_ptr(), b.data_ptr() if use_b else None, dav.data_ptr(), dbv.data_ptr() if use_b else None,
                              n, w, target))
    arr = (L.LossJob * len(jobs))(*jobs)
    ws = torch.empty(64 * L.MAX_LOSS_JOBS, device='cuda')
    loss = torch.zeros(1, device='cuda')
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.rtg_loss_fwd(kind, arr, len(jobs), _ptr(ws), _ptr(loss), st) == 0
    gscale = torch.full((1,), 0.5, device='cuda')
    assert lib.rtg_loss_bwd(kind, arr, len(jobs), _ptr(gscale), st) == 0
    torch.cuda.synchronize()
    np.testing.assert_allclose(loss.item(), ref_loss.item(), rtol=2e-5)
    for ga, gb, dav, dbv, use_b in refs:
        np.testing.assert_allclose(dav.cpu().double().numpy(), 0.5 * ga.numpy(), rtol=1e-5, atol=1e-9)
        if use_b and kind in (L.LOSS_L1, L.LOSS_L1_L1LOG):
            np.testing.assert_allclose(dbv.cpu().double().numpy(), 0.5 * gb.numpy(), rtol=1e-5, atol=1e-9)
