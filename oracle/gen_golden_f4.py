#!/usr/bin/env python3
"""Generate tests/golden/retunegan_f4_b2_t8192.npz — SURVEY.md 8 f3/f4 rows — by importing and RUNNING the reference
(build container only; same stand-in modules and rules as oracle/gen_golden.py: TEST INFRASTRUCTURE, nothing of the
reference is copied, the fixture holds inputs-by-recipe + expected outputs).

    cd /tmp && python /root/repo/oracle/gen_golden_f4.py

Covers: the full-size `Generator_RefineGAN` (construction under the reference's seed, forward, gradients), the losses
switched off by default (envelope, strip-mirror, relative LSGAN: values and gradients), and the inference path
(`remove_weight_norm`, batch 1, a length that is not the training segment)."""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, '/root/reference/retunegan')
sys.path.insert(0, os.path.join(HERE, 'stubs'))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import hparam as hp  # noqa: E402  (reference)
import models as M  # noqa: E402  (reference)
import models.loss as RL  # noqa: E402  (reference)

import importlib.util  # noqa: E402

_spec = importlib.util.spec_from_file_location('rtg_oracle', os.path.join(HERE, 'rtg_oracle.py'))
O = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(O)
torch.set_num_threads(8)


def stats(t):
    t = t.detach().double()
    return np.array([t.sum().item(), t.abs().mean().item()], dtype=np.float64)


def main():
    gold = {}
    # ---------------------------------------------------------------- full-size generator: construction
    torch.manual_seed(hp.randseed)
    g = M.Generator_RefineGAN()
    gold['full_count'] = np.array(sum(p.numel() for p in g.parameters()))
    gold['full_keys'] = np.array(sorted(g.state_dict().keys()))
    gold['full_init_stats'] = np.stack([stats(p) for _, p in sorted(g.named_parameters())])
    gold['full_param_order'] = np.array([n for n, _ in g.named_parameters()])      # optimizer state index order
    # ---------------------------------------------------------------- forward / backward on the golden inputs
    O.det_fill(g)
    g.train()
    x, y_tmpl, y = O.golden_inputs()
    y_hat = g(x, y_tmpl)
    gold['full_yhat'] = y_hat.detach().numpy()
    loss = (y_hat - y).abs().mean() + RL.dynamic_loss(y, y_hat)
    loss.backward()
    gold['full_loss'] = np.array(loss.item())
    names = ['conv_pre_y.weight_v', 'conv_pre.weight_g', 'downs.2.bias', 'ups.0.weight_v', 'ups.0.weight_g',
             'resblock.1.convs.1.weight_v', 'resblocks.4.convs.0.weight_v', 'merge.1.weight_v', 'conv_post.weight_v']
    pd = dict(g.named_parameters())
    gold['full_grad_names'] = np.array(names)
    gold['full_grad_stats'] = np.stack([stats(pd[n].grad) for n in names])
    gold['full_grad_conv_post_v'] = pd['conv_post.weight_v'].grad.numpy()

    # ---------------------------------------------------------------- disabled losses: values and gradients w.r.t. y_hat
    gs = M.Generator_RefineGAN_small()
    O.det_fill(gs)
    yh = gs(x, y_tmpl).detach().requires_grad_(True)
    env = RL.envelope_loss(y, yh)
    sm = RL.strip_mirror_loss(yh)
    (4 * env + 0.01 * sm).backward()
    gold['loss_env'], gold['loss_sm'] = np.array(env.item()), np.array(sm.item())
    gold['grad_env_sm_yhat'] = yh.grad.numpy()
    yo = torch.rand(2, 1, 4097, generator=torch.Generator().manual_seed(3)) * 2 - 1      # odd length: last sample dropped
    yo.requires_grad_(True)
    smo = RL.strip_mirror_loss(yo)
    smo.backward()
    gold['loss_sm_odd'], gold['grad_sm_odd'] = np.array(smo.item()), yo.grad.numpy()

    # ---------------------------------------------------------------- relative LSGAN on MSD logits
    msd = M.MultiScaleDiscriminator()
    O.det_fill(msd)
    yh2 = yh.detach().requires_grad_(True)
    hp.relative_gan_loss = True
    try:
        dr, dg, _, _ = msd(y, yh2.detach())
        dl = RL.discriminator_loss(dr, dg)
        dl.backward()
        gold['rel_d_loss'] = np.array(dl.item())
        pm = dict(msd.named_parameters())
        gold['rel_d_grad_stats'] = np.stack([stats(pm[n].grad) for n in ('discriminators.0.conv_post.weight_v',
                                                                         'discriminators.2.convs.1.weight_g')])
        msd.zero_grad()
        dr, dg, _, _ = msd(y, yh2)
        gl = RL.generator_loss(dg, dr)
        gl.backward()
        gold['rel_g_loss'] = np.array(gl.item())
        gold['rel_g_grad_yhat_stats'] = stats(yh2.grad)
    finally:
        hp.relative_gan_loss = False

    # ---------------------------------------------------------------- inference path: B=1, 37 frames, weight norm removed
    torch.manual_seed(1)
    xi, yi = torch.randn(1, 80, 37).abs(), torch.rand(1, 1, 37 * 256) * 2 - 1
    gs.noise.w.data.zero_()
    gs.eval()
    with torch.no_grad():
        a = gs(xi, yi)
        gs.remove_weight_norm()
        b = gs(xi, yi)
    gold['infer_x'], gold['infer_y'] = xi.numpy(), yi.numpy()
    gold['infer_out'] = a.numpy()
    gold['infer_out_nown_maxdiff'] = np.array((a - b).abs().max().item())
    gold['infer_keys_nown'] = np.array(sorted(gs.state_dict().keys()))

    out = os.path.join(REPO, 'tests', 'golden', 'retunegan_f4_b2_t8192.npz')
    np.savez_compressed(out, **gold)
    print('wrote', out, os.path.getsize(out), 'bytes;', {k: (v.shape if hasattr(v, 'shape') else v) for k, v in gold.items()
                                                          if v.size < 4})


if __name__ == '__main__':
    main()
