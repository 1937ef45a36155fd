"""Pins the oracle's restatement of the SURVEY.md 8 f3/f4 rows (full-size Generator_RefineGAN, the losses switched off
by default, the inference path) to fixtures produced by running the reference (oracle/gen_golden_f4.py), and checks
the host-side construction of the product's full-size generator.  CPU only."""
import numpy as np
import torch


def stats(t):
    t = t.detach().double()
    return np.array([t.sum().item(), t.abs().mean().item()])


def test_full_generator_construction(oracle, gold4):
    torch.manual_seed(114514)
    g = oracle.GeneratorFull()
    assert sum(p.numel() for p in g.parameters()) == int(gold4['full_count'])
    assert sorted(g.state_dict().keys()) == list(gold4['full_keys'])
    st = np.stack([stats(p) for _, p in sorted(g.named_parameters())])
    np.testing.assert_allclose(st, gold4['full_init_stats'], rtol=1e-6, atol=1e-7)


def test_product_full_generator_construction(gold4):
    """same key set and bit-identical initial parameters under the reference's seed (CPU construction, no kernels)"""
    from models import Generator_RefineGAN
    torch.manual_seed(114514)
    g = Generator_RefineGAN()
    assert sum(p.numel() for p in g.parameters()) == int(gold4['full_count'])
    assert sorted(g.state_dict().keys()) == list(gold4['full_keys'])
    st = np.stack([stats(p) for _, p in sorted(g.named_parameters())])
    np.testing.assert_allclose(st, gold4['full_init_stats'], rtol=1e-6, atol=1e-7)
    import models
    assert getattr(models, 'Generator_RefineGAN') is Generator_RefineGAN       # train.py:48 looks the class up by name


def test_full_generator_forward_backward(oracle, gold4):
    g = oracle.GeneratorFull()
    oracle.det_fill(g)
    g.train()
    x, y_tmpl, y = oracle.golden_inputs()
    y_hat = g(x, y_tmpl)
    np.testing.assert_allclose(y_hat.detach().numpy(), gold4['full_yhat'], atol=2e-5)
    loss = (y_hat - y).abs().mean() + oracle.dynamic_loss(y, y_hat)
    loss.backward()
    np.testing.assert_allclose(loss.item(), gold4['full_loss'], rtol=1e-5)
    pd = dict(g.named_parameters())
    for n, ref in zip(gold4['full_grad_names'], gold4['full_grad_stats']):
        np.testing.assert_allclose(stats(pd[str(n)].grad)[1], ref[1], rtol=2e-3, atol=1e-7)
    np.testing.assert_allclose(pd['conv_post.weight_v'].grad.numpy(), gold4['full_grad_conv_post_v'], rtol=2e-3, atol=2e-6)


def test_disabled_losses(oracle, gold, gold4):
    x, y_tmpl, y = oracle.golden_inputs()
    # the reference's own y_hat (bit-identical input: 1/|d| in the strip-mirror gradient amplifies any input noise)
    yh = torch.from_numpy(gold['y_hat']).clone().requires_grad_(True)
    env, sm = oracle.envelope_loss(y, yh), oracle.strip_mirror_loss(yh)
    np.testing.assert_allclose(env.item(), gold4['loss_env'], rtol=1e-5)
    np.testing.assert_allclose(sm.item(), gold4['loss_sm'], rtol=1e-5)
    (4 * env + 0.01 * sm).backward()
    np.testing.assert_allclose(yh.grad.numpy(), gold4['grad_env_sm_yhat'], rtol=1e-3, atol=1e-7)
    yo = (torch.rand(2, 1, 4097, generator=torch.Generator().manual_seed(3)) * 2 - 1).requires_grad_(True)
    smo = oracle.strip_mirror_loss(yo)
    smo.backward()
    np.testing.assert_allclose(smo.item(), gold4['loss_sm_odd'], rtol=1e-5)
    np.testing.assert_allclose(yo.grad.numpy(), gold4['grad_sm_odd'], rtol=1e-3, atol=1e-8)


def test_relative_gan_losses(oracle, gold4):
    gs, msd = oracle.Generator(), oracle.MSD()
    oracle.det_fill(gs)
    oracle.det_fill(msd)
    x, y_tmpl, y = oracle.golden_inputs()
    yh = gs(x, y_tmpl).detach().requires_grad_(True)
    dr, dg, _, _ = msd(y, yh.detach())
    dl = oracle.discriminator_loss(dr, dg, relative=True)
    dl.backward()
    np.testing.assert_allclose(dl.item(), gold4['rel_d_loss'], rtol=1e-5)
    pm = dict(msd.named_parameters())
    for n, ref in zip(('discriminators.0.conv_post.weight_v', 'discriminators.2.convs.1.weight_g'), gold4['rel_d_grad_stats']):
        np.testing.assert_allclose(stats(pm[n].grad)[1], ref[1], rtol=2e-3)
    msd.zero_grad()
    dr, dg, _, _ = msd(y, yh)
    gl = oracle.generator_loss(dg, dr, relative=True)
    gl.backward()
    np.testing.assert_allclose(gl.item(), gold4['rel_g_loss'], rtol=1e-5)
    np.testing.assert_allclose(stats(yh.grad)[1], gold4['rel_g_grad_yhat_stats'][1], rtol=2e-3)


def test_inference_path(oracle, gold4):
    gs = oracle.Generator()
    oracle.det_fill(gs)
    gs.eval()
    with torch.no_grad():
        out = gs(torch.from_numpy(gold4['infer_x']), torch.from_numpy(gold4['infer_y']))
    np.testing.assert_allclose(out.numpy(), gold4['infer_out'], atol=2e-5)
    assert float(gold4['infer_out_nown_maxdiff']) < 1e-5        # the reference's own remove_weight_norm changes nothing
