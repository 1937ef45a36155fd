"""rtg_resstack.hip: a whole ResidualStack (generator.py:33-77) in one launch per direction, against the CPU oracle's
ResidualStack (float64) and against the six-launch path it replaces.  GPU only."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _net(C):
    from models.generator import ResidualStack
    from models.layers import BankedModel

    class Net(BankedModel):
        def __init__(self):
            super().__init__()
            self.stack = ResidualStack(C)

        def forward(self, x, final=None):
            return self.stack.run(self.token(), x, final_act_slope=final)
    return Net()


@pytest.mark.parametrize('final', [None, 0.15])
@pytest.mark.parametrize('C,L,B', [(128, 32, 5), (64, 256, 3), (64, 200, 2), (32, 2048, 2), (32, 500, 3)])
def test_fused_residual_stack_matches_oracle_and_unfused(oracle, C, L, B, final, monkeypatch):
    from rtg import ops
    monkeypatch.setattr(ops, 'RESSTACK_KINDS', 7)        # the C = 64 / 32 instances are not used by default (not faster)
    torch.manual_seed(C + L)
    net = _net(C)
    ref = oracle.ResidualStack(C).double()
    ref.load_state_dict({k[len('stack.'):]: v.double() for k, v in net.state_dict().items()})
    net.to(DEV)
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(B, C, L, generator=gen)
    dy = torch.randn(B, C, L, generator=gen)

    def run(mode):
        # 'fused': one launch per direction; 'node': one autograd node, six launches per direction (residual gradient in
        # the backward-data epilogue, grouped weight gradients); 'legacy': one autograd node per conv
        ops.RESSTACK = mode != 'legacy'
        ops.RESSTACK_KINDS = 7 if mode == 'fused' else 0
        try:
            net.zero_grad()
            xg = x.to(DEV).requires_grad_(True)
            y = net(xg, final)
            y.backward(dy.to(DEV))
            torch.cuda.synchronize()
            return y.detach().cpu(), xg.grad.cpu(), {n: p.grad.detach().cpu().clone() for n, p in net.named_parameters()}
        finally:
            ops.RESSTACK = True
            ops.RESSTACK_KINDS = 7

    lys = [getattr(blk, n)._layer for blk in (net.stack.res_1, net.stack.res_2, net.stack.res_3) for n in ('1', '3')] \
        if net._bank is not None else None
    net.bank()                                            # builds the layers
    lys = [getattr(blk, n)._layer for blk in (net.stack.res_1, net.stack.res_2, net.stack.res_3) for n in ('1', '3')]
    assert ops.resstack_ok(lys, x.to(DEV))
    y1, dx1, g1 = run('fused')
    y2, dx2, g2 = run('node')
    y0, dx0, g0 = run('legacy')
    assert torch.equal(y2, y0)                            # the same six forward launches
    xr = x.double().requires_grad_(True)
    yr = ref(xr)
    if final is not None:
        yr = F.leaky_relu(yr, final)
    yr.backward(dy.double())
    rp = dict(ref.named_parameters())
    for got, name in ((y1, 'fused'), (y2, 'node'), (y0, 'legacy')):
        np.testing.assert_allclose(got.numpy(), yr.detach().float().numpy(), rtol=1e-4, atol=2e-5, err_msg=name)
    for got, name in ((dx1, 'fused'), (dx2, 'node'), (dx0, 'legacy')):
        err = (got.double() - xr.grad).norm().item() / xr.grad.norm().item()
        assert err < 2e-4, (name, err)       # relative L2: a leaky-relu flip of an activation within rounding of 0 is local
    for n, g in g1.items():
        r = rp[n[len('stack.'):]].grad
        err = (g.double() - r).norm().item() / (r.norm().item() + 1e-30)
        assert err < 2e-4, (n, err)
        for other in (g0, g2):
            err0 = (g - other[n]).norm().item() / (other[n].norm().item() + 1e-30)
            assert err0 < 2e-4, (n, err0)


def test_stack_shapes_that_are_not_served_fall_back(oracle):
    """clips shorter than a tile (C = 32 below 128 samples) keep the six-launch path"""
    from rtg import ops
    net = _net(32).to(DEV)
    net.bank()
    lys = [getattr(blk, n)._layer for blk in (net.stack.res_1, net.stack.res_2, net.stack.res_3) for n in ('1', '3')]
    assert not ops.resstack_ok(lys, torch.zeros(2, 32, 64, device=DEV))
    y = net(torch.randn(2, 32, 64, device=DEV))
    assert y.shape == (2, 32, 64) and torch.isfinite(y).all()
