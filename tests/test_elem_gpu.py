"""Small launches of rtg_elem.hip added in round 4 (ABI 8): the weighted sum of the loss terms and the GaussianNoise
backward that accumulates the shared scalar's gradient itself.  GPU only."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_weighted_sum_is_one_node_with_torch_values_and_gradients():
    """ops.weighted_sum (train.py:137-158, 170-189 of the reference: the loss totals): value and gradients of
    sum_i w_i * t_i against the ATen expression it replaces, terms of every producer kind (leaf, computed, no grad)."""
    from rtg import ops
    gen = torch.Generator().manual_seed(2)
    vals = torch.randn(6, generator=gen).tolist()
    ws = [8.0, 1.0, 2.0, 0.5, 1.0, 3.0]

    def terms():
        leaves = [torch.tensor(v, device='cuda', requires_grad=(i != 4)) for i, v in enumerate(vals)]
        ts = [leaves[0], leaves[1] * 2.0, leaves[2].reshape(1), leaves[3], leaves[4], leaves[5] ** 2]
        return leaves, ts
    la, ta = terms()
    total = ops.weighted_sum(ta, ws)
    assert total.shape == () and total.grad_fn is not None and type(total.grad_fn).__name__.startswith('WSumFn')
    lb, tb = terms()
    ref = sum(t.reshape(()) * w for t, w in zip(tb, ws))
    np.testing.assert_allclose(total.item(), ref.item(), rtol=1e-6)
    (total * 1.5).backward()
    (ref * 1.5).backward()
    for i, (a, b) in enumerate(zip(la, lb)):
        if b.grad is None:
            assert a.grad is None, i
        else:
            np.testing.assert_allclose(a.grad.item(), b.grad.item(), rtol=1e-6)
    # more terms than the table holds: the plain expression
    many = [torch.tensor(float(i), device='cuda') for i in range(20)]
    assert ops.weighted_sum(many).item() == pytest.approx(sum(range(20)))


@pytest.mark.parametrize('shape', [(3, 32, 1000), (2, 16, 4096)])
def test_noise_backward_accumulates_the_scalar_gradient_in_place(shape):
    """NoiseFn.backward (generator.py:19-30): inside ops.noise_grad_accumulate() (train.Trainer.g_step's backward) with a
    live w.grad buffer the launch pair rtg_noise_lrelu_bwd_acc adds sum(dy * lrelu' * u) to it and autograd receives no
    gradient for w; the result equals the part.sum() + accumulation path (any other backward) and the closed form on a
    replayed noise field, and two calls accumulate.  torch.autograd.grad outside the context returns dw and leaves w.grad
    alone (round-4 advisor finding)."""
    from rtg import ops
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(*shape, generator=gen).cuda()
    u = torch.rand(*shape, generator=gen).cuda()
    dy = torch.randn(*shape, generator=gen).cuda()
    w0 = 0.3

    def run(acc, calls):
        import contextlib
        w = torch.nn.Parameter(torch.tensor([w0], device='cuda'))
        w.grad = torch.full((1,), 0.25, device='cuda')            # a live buffer with a value to accumulate onto
        xs = []
        for _ in range(calls):
            xi = x.clone().requires_grad_(True)
            out = ops.NoiseFn.apply(xi, w, u, 0.15, 1234, None)
            with (ops.noise_grad_accumulate() if acc else contextlib.nullcontext()):
                out.backward(dy)
            xs.append(xi.grad)
        torch.cuda.synchronize()
        return w.grad.item(), xs
    ga, xa = run(True, 2)
    gb, xb = run(False, 2)
    pre = x.double() + u.double() * w0
    d = dy.double() * torch.where(pre > 0, 1.0, 0.15)
    want = 0.25 + 2 * (d * u.double()).sum().item()
    assert ga == pytest.approx(want, rel=2e-5) and gb == pytest.approx(want, rel=2e-5)
    for a, b in zip(xa, xb):
        assert torch.equal(a, b)
        np.testing.assert_allclose(a.cpu().numpy(), d.float().cpu().numpy(), rtol=1e-6, atol=1e-7)
    # torch.autograd.grad with a live w.grad: dw comes back, w.grad is not touched
    w = torch.nn.Parameter(torch.tensor([w0], device='cuda'))
    w.grad = torch.full((1,), 0.25, device='cuda')
    xi = x.clone().requires_grad_(True)
    out = ops.NoiseFn.apply(xi, w, u, 0.15, 1234, None)
    dx, dw = torch.autograd.grad(out, (xi, w), dy)
    assert w.grad.item() == 0.25
    assert dw.item() == pytest.approx((d * u.double()).sum().item(), rel=2e-5)
    (dx_only,) = torch.autograd.grad(ops.NoiseFn.apply(xi, w, u, 0.15, 1234, None), (xi,), dy)
    assert w.grad.item() == 0.25 and torch.equal(dx_only, dx)
