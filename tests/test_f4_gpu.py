"""SURVEY.md 8 f3/f4 rows on the HIP path against the fixtures the reference produced (oracle/gen_golden_f4.py): the
full-size Generator_RefineGAN, the losses switched off by default (envelope, strip-mirror, relative LSGAN) and the
inference path (batch 1, 37 frames, remove_weight_norm).  Tolerances as in DESIGN.md 5 (fp32): waves atol 1e-4,
losses rel 1e-4, gradients rel 2e-3 (relative L2 where a leaky-relu kink can flip single elements).  GPU only."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def stats(t):
    t = t.detach().double().cpu()
    return np.array([t.sum().item(), t.abs().mean().item()])


def rel_l2(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30)


@pytest.fixture(scope='module')
def full(oracle):
    from models import Generator_RefineGAN
    g = Generator_RefineGAN()
    oracle.det_fill(g)
    return g.to(DEV).train()


def test_full_generator_forward(full, oracle, gold4):
    x, y_tmpl, _ = oracle.golden_inputs()
    with torch.no_grad():
        y_hat = full(x.to(DEV), y_tmpl.to(DEV))
    assert y_hat.shape == (2, 1, 8192)
    np.testing.assert_allclose(y_hat.cpu().numpy(), gold4['full_yhat'], atol=1e-4, rtol=0)
    assert full.bank().n_params == int(gold4['full_count'])


def test_full_generator_backward(full, oracle, gold4):
    from models import dynamic_loss
    x, y_tmpl, y = oracle.golden_inputs()
    full.zero_grad()
    y_hat = full(x.to(DEV), y_tmpl.to(DEV))
    yd = y.to(DEV)
    # |y_hat - y|.mean() through the L1 kernel + the dynamic loss, as in the fixture
    from rtg import ops
    from rtg.lib import LOSS_L1
    loss = ops.multi_loss(LOSS_L1, [y_hat], [yd]) + dynamic_loss(yd, y_hat)
    loss.backward()
    torch.cuda.synchronize()
    np.testing.assert_allclose(loss.item(), gold4['full_loss'], rtol=1e-4)
    pd = dict(full.named_parameters())
    for n, ref in zip(gold4['full_grad_names'], gold4['full_grad_stats']):
        np.testing.assert_allclose(stats(pd[str(n)].grad)[1], ref[1], rtol=2e-3, atol=1e-7, err_msg=str(n))
    assert rel_l2(pd['conv_post.weight_v'].grad.cpu().numpy(), gold4['full_grad_conv_post_v']) < 2e-3


def test_envelope_and_strip_mirror_losses(oracle, gold, gold4):
    from models import envelope_loss, strip_mirror_loss
    _, _, y = oracle.golden_inputs()
    yh = torch.from_numpy(gold['y_hat']).to(DEV).requires_grad_(True)
    env, sm = envelope_loss(y.to(DEV), yh), strip_mirror_loss(yh)
    np.testing.assert_allclose(env.item(), gold4['loss_env'], rtol=1e-4)
    np.testing.assert_allclose(sm.item(), gold4['loss_sm'], rtol=1e-4)
    (4 * env + 0.01 * sm).backward()
    got, ref = yh.grad.cpu().numpy(), gold4['grad_env_sm_yhat']
    # -1/(|d| + 1e-9) of the strip-mirror gradient amplifies the rounding of d = u - mean(u) where |d| is tiny: compare
    # every element at rel 1e-3 of max(|ref|, typical size), and the whole vector in L2
    np.testing.assert_allclose(got, ref, rtol=2e-3, atol=2e-3 * np.abs(ref).mean())
    assert rel_l2(got, ref) < 1e-3
    yo = (torch.rand(2, 1, 4097, generator=torch.Generator().manual_seed(3)) * 2 - 1).to(DEV).requires_grad_(True)
    smo = strip_mirror_loss(yo)
    smo.backward()
    np.testing.assert_allclose(smo.item(), gold4['loss_sm_odd'], rtol=1e-4)
    ref = gold4['grad_sm_odd']
    np.testing.assert_allclose(yo.grad.cpu().numpy(), ref, rtol=2e-3, atol=2e-3 * np.abs(ref).mean())
    assert (yo.grad[:, :, -1] == 0).all()


def test_relative_gan_losses(oracle, gold, gold4):
    import hparam as hp
    from models import MultiScaleDiscriminator, discriminator_loss, generator_loss
    msd = MultiScaleDiscriminator()
    oracle.det_fill(msd)
    msd.to(DEV).train()
    _, _, y = oracle.golden_inputs()
    y = y.to(DEV)
    yh = torch.from_numpy(gold['y_hat']).to(DEV).requires_grad_(True)
    hp.relative_gan_loss = True
    try:
        dr, dg, _, _ = msd(y, yh.detach())
        dl = discriminator_loss(dr, dg)
        dl.backward()
        np.testing.assert_allclose(dl.item(), gold4['rel_d_loss'], rtol=1e-4)
        pm = dict(msd.named_parameters())
        for n, ref in zip(('discriminators.0.conv_post.weight_v', 'discriminators.2.convs.1.weight_g'),
                          gold4['rel_d_grad_stats']):
            np.testing.assert_allclose(stats(pm[n].grad)[1], ref[1], rtol=2e-3, err_msg=n)
        msd.zero_grad()
        dr, dg, _, _ = msd(y, yh)
        gl = generator_loss(dg, dr)
        gl.backward()
        np.testing.assert_allclose(gl.item(), gold4['rel_g_loss'], rtol=1e-4)
        np.testing.assert_allclose(stats(yh.grad)[1], gold4['rel_g_grad_yhat_stats'][1], rtol=2e-3)
    finally:
        hp.relative_gan_loss = False


def test_inference_path(oracle, gold4):
    """server.py:33-81: eval mode, weight norm removed, batch 1, a length that is not the training segment."""
    from models import Generator_RefineGAN_small
    g = Generator_RefineGAN_small()
    oracle.det_fill(g)
    g.to(DEV).eval()
    xi, yi = torch.from_numpy(gold4['infer_x']).to(DEV), torch.from_numpy(gold4['infer_y']).to(DEV)
    with torch.no_grad():
        a = g(xi, yi)
        g.remove_weight_norm()
        b = g(xi, yi)
    assert a.shape == (1, 1, 37 * 256)
    np.testing.assert_allclose(a.cpu().numpy(), gold4['infer_out'], atol=1e-4, rtol=0)
    assert (a - b).abs().max().item() == 0.0
