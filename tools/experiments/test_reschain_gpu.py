"""rtg_reschain.hip: a whole ResBlock3 branch of the UNet-G decoder (generator.py:133-155) in one launch per direction, against
the three-launch path it replaces — bit for bit: forward output, input gradient, parameter gradients (the weight gradients run on
the same tensors either way) — and against float64 torch.  GPU only."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _net(C, k, dil):
    from models.generator import ResBlock3
    from models.layers import BankedModel

    class Net(BankedModel):
        def __init__(self):
            super().__init__()
            self.blk = ResBlock3(C, k, dil)

        def forward(self, x):
            return self.blk.run(self.token(), x)
    return Net()


@pytest.mark.parametrize('k', [3, 5, 7])
@pytest.mark.parametrize('B,L', [(2, 8192), (3, 1000), (1, 64), (5, 332)])
def test_fused_branch_is_bit_identical_to_the_layers_one_by_one(k, B, L, monkeypatch):
    from rtg import ops
    torch.manual_seed(10 * k + B)
    net = _net(32, k, (9, 3, 1)).to(DEV)
    net.bank()
    lys = [c._layer for c in net.blk.convs]
    gen = torch.Generator().manual_seed(L)
    x = torch.randn(B, 32, L, generator=gen)
    dy = torch.randn(B, 32, L, generator=gen)
    assert ops.reschain_ok(lys, x.to(DEV))

    def run(fused):
        monkeypatch.setattr(ops, 'RESCHAIN', fused)
        net.zero_grad()
        xg = x.to(DEV).requires_grad_(True)
        y = net(xg)
        y.backward(dy.to(DEV))
        torch.cuda.synchronize()
        return y.detach().cpu(), xg.grad.cpu(), {n: p.grad.detach().cpu().clone() for n, p in net.named_parameters()}

    y1, dx1, g1 = run(True)
    y0, dx0, g0 = run(False)
    assert torch.equal(y1, y0), (y1 - y0).abs().max().item()
    assert torch.equal(dx1, dx0), (dx1 - dx0).abs().max().item()
    for n in g0:
        assert torch.equal(g1[n], g0[n]), (n, (g1[n] - g0[n]).abs().max().item())
    # ... and both are the block: float64 torch on the effective (weight-normed) weights
    xr = x.double().requires_grad_(True)
    cur = xr
    sd = {n: p.detach().cpu().double() for n, p in net.named_parameters()}
    for i, d in enumerate((9, 3, 1)):
        v, g, bias = sd[f'blk.convs.{i}.weight_v'], sd[f'blk.convs.{i}.weight_g'], sd[f'blk.convs.{i}.bias']
        w = g * v / v.flatten(1).norm(dim=1).view(-1, 1, 1)
        cur = cur + F.conv1d(F.leaky_relu(cur, 0.15), w, bias, 1, (k - 1) // 2 * d, d)
    cur.backward(dy.double())
    np.testing.assert_allclose(y1.numpy(), cur.detach().float().numpy(), rtol=1e-4, atol=5e-5)
    err = (dx1.double() - xr.grad).norm().item() / xr.grad.norm().item()
    assert err < 2e-4, err


def test_shapes_the_fused_branch_does_not_serve_fall_back():
    from rtg import ops
    net = _net(64, 3, (9, 3, 1)).to(DEV)
    net.bank()
    assert not ops.reschain_ok([c._layer for c in net.blk.convs], torch.zeros(2, 64, 256, device=DEV))
    net32 = _net(32, 3, (9, 3, 1)).to(DEV)
    net32.bank()
    assert not ops.reschain_ok([c._layer for c in net32.blk.convs], torch.zeros(2, 32, 62, device=DEV))     # L % 4, L < 64
    y = net(torch.randn(2, 64, 256, device=DEV))
    assert torch.isfinite(y).all()
