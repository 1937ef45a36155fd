#!/usr/bin/env python3
"""Generate tests/golden/*.npz by importing and RUNNING the reference itself (build container only).

    cd /tmp && python /root/repo/oracle/gen_golden.py

*** TEST INFRASTRUCTURE. ***  Needs /root/reference (absent on the GPU box); nothing of the reference is copied: the
fixtures are inputs-by-recipe + expected outputs.  The reference is imported with three stand-in modules from
oracle/stubs (seaborn, tensorboardX: unused on the path; librosa.filters.mel: published Slaney formula) exactly as
SURVEY.md Appendix B describes.  The step sequence below calls the reference's own functions in the order of
retunegan/train.py:121-193.
"""
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
from audio import get_stft_torch  # noqa: E402  (reference)

sys.path.insert(0, HERE)
import importlib.util  # noqa: E402

_spec = importlib.util.spec_from_file_location('rtg_oracle', os.path.join(HERE, 'rtg_oracle.py'))
O = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(O)

OUT = os.path.join(REPO, 'tests', 'golden')
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(8)


def stats(t):
    t = t.detach().double()
    return np.array([t.sum().item(), t.abs().mean().item()], dtype=np.float64)


def sample_idx(n, k=257, seed=7):
    return np.random.RandomState(seed).choice(n, size=min(k, n), replace=False)


def build_ref():
    g = M.Generator_RefineGAN_small()
    msd, mpd, mtd = M.MultiScaleDiscriminator(), M.MultiPeriodDiscriminator(), M.MultiStftDiscriminator()
    return g, msd, mpd, mtd


def main():
    gold = {}
    # ------------------------------------------------------------------ construction under the reference's seed
    torch.manual_seed(hp.randseed)
    g, msd, mpd, mtd = build_ref()
    for tag, m in (('g', g), ('msd', msd), ('mpd', mpd), ('mtd', mtd)):
        gold[f'init_{tag}_count'] = np.array(sum(p.numel() for p in m.parameters()))
        gold[f'init_{tag}_keys'] = np.array(sorted(m.state_dict().keys()))
        gold[f'init_{tag}_stats'] = np.stack([stats(p) for _, p in sorted(m.named_parameters())])
        # registration order of the parameters: torch.optim state dicts index their entries by it (train.py:263-273
        # saves optim_g / optim_d that way), so a drop-in must reproduce it, not only the key set
        gold[f'init_{tag}_param_order'] = np.array([n for n, _ in m.named_parameters()])

    # ------------------------------------------------------------------ deterministic fill + golden inputs
    for m in (g, msd, mpd, mtd):
        O.det_fill(m)
        m.train()
    x, y_tmpl, y = O.golden_inputs()

    # ------------------------------------------------------------------ mel basis + STFT of y
    from audio import mel_basis_torch
    for n_fft, win, hop in hp.multi_stft_params:
        S, Mel, P = get_stft_torch(y.squeeze(1), n_fft, win, hop)
        gold[f'stft{n_fft}_mel'] = Mel.numpy()
        idx = sample_idx(S.numel())
        gold[f'stft{n_fft}_idx'] = idx
        gold[f'stft{n_fft}_S'] = S.flatten().numpy()[idx]
        gold[f'stft{n_fft}_P'] = P.flatten().numpy()[idx]
        gold[f'stft{n_fft}_S_stats'] = stats(S)
        gold[f'stft{n_fft}_logS_stats'] = stats(torch.log(S))
        gold[f'melbasis{n_fft}_sum'] = np.array(mel_basis_torch[n_fft].double().sum().item())
        gold[f'melbasis{n_fft}_rowsum'] = mel_basis_torch[n_fft].double().sum(1).numpy()

    # ------------------------------------------------------------------ generator forward (noise.w == 0)
    y_hat = g(x, y_tmpl)
    gold['y_hat'] = y_hat.detach().numpy()

    # ------------------------------------------------------------------ discriminators on (y, y_hat)
    yd = y_hat.detach()
    S, Sg = M.multi_stft_loss(y, yd, ret_specs=True)
    for tag, d, a, b in (('msd', msd, y, yd), ('mpd', mpd, y, yd), ('mtd', mtd, S, Sg)):
        lr, lg, fr, fg = d(a, b)
        for i, (r, gg) in enumerate(zip(lr, lg)):
            gold[f'{tag}_logit_r{i}'] = r.detach().numpy()
            gold[f'{tag}_logit_g{i}'] = gg.detach().numpy()
        gold[f'{tag}_fmap_r_stats'] = np.stack([stats(f) for fl in fr for f in fl])
        gold[f'{tag}_fmap_g_stats'] = np.stack([stats(f) for fl in fg for f in fl])
        gold[f'{tag}_fmap_shapes'] = np.array([list(f.shape) + [1] * (4 - f.dim()) for fl in fr for f in fl])
        gold[f'{tag}_d_loss'] = np.array(M.discriminator_loss(lr, lg).item())
        gold[f'{tag}_g_loss'] = np.array(M.generator_loss(lg, lr).item())
        gold[f'{tag}_fm_loss'] = np.array(M.feature_loss(fr, fg).item())

    # frame 0 of the centred, reflect-padded STFT is symmetric: its spectrum is real up to rounding and the sign of the
    # reference's angle() (+pi or -pi) there is rounding noise of THIS run.  The MTD values above were computed from it,
    # so the branch the reference took is part of the fixture (tests put GPU phases that sit on the cut on this branch).
    for j, spec_list in enumerate(zip(S, Sg)):
        n_fft = hp.multi_stft_params[j][0]
        gold[f'stft{n_fft}_frame0_phase_r'] = spec_list[0][:, 1, :, 0].numpy().copy()
        gold[f'stft{n_fft}_frame0_phase_g'] = spec_list[1][:, 1, :, 0].numpy().copy()

    # ------------------------------------------------------------------ waveform / spectral losses
    gold['loss_mstft'] = np.array(M.multi_stft_loss(y, yd, ret_loss=True).item())
    gold['loss_dyn'] = np.array(M.dynamic_loss(y, yd).item())
    gold['loss_env'] = np.array(M.envelope_loss(y, yd).item())
    gold['loss_sm'] = np.array(M.strip_mirror_loss(yd).item())

    # ------------------------------------------------------------------ D backward (train.py:139-158), full stack
    for m in (g, msd, mpd, mtd):
        m.zero_grad()
    S, Sg = M.multi_stft_loss(y, yd, ret_specs=True)
    r1, g1, _, _ = msd(y, yd)
    r2, g2, _, _ = mpd(y, yd)
    r3, g3, _, _ = mtd(S, Sg)
    ld = M.discriminator_loss(r1, g1) + M.discriminator_loss(r2, g2) + M.discriminator_loss(r3, g3)
    ld.backward()
    gold['loss_disc_all'] = np.array(ld.item())
    for tag, m in (('msd', msd), ('mpd', mpd), ('mtd', mtd)):
        gold[f'dgrad_{tag}_stats'] = np.stack([stats(p.grad) for _, p in sorted(m.named_parameters())])

    # ------------------------------------------------------------------ G backward (train.py:163-191), full stack
    for m in (g, msd, mpd, mtd):
        m.zero_grad()
    torch.manual_seed(4321)          # pins the six rand_like draws of this forward (d loss / d noise.w depends on them)
    y_hat = g(x, y_tmpl)
    y_hat.retain_grad()
    lm, (S, Sgh) = M.multi_stft_loss(y, y_hat, ret_loss=True, ret_specs=True)
    ldyn = M.dynamic_loss(y, y_hat)
    tot = lm * hp.w_loss_mstft + ldyn * hp.w_loss_dyn
    for d, a, b in ((msd, y, y_hat), (mpd, y, y_hat), (mtd, S, Sgh)):
        lr, lg, fr, fg = d(a, b)
        tot = tot + M.generator_loss(lg, lr) + M.feature_loss(fr, fg) * hp.w_loss_fm
    tot.backward()
    gold['loss_gen_all'] = np.array(tot.item())
    gold['ggrad_yhat'] = y_hat.grad.numpy()
    gold['ggrad_g_stats'] = np.stack([stats(p.grad) for _, p in sorted(g.named_parameters())])
    gold['ggrad_g_names'] = np.array([n for n, _ in sorted(g.named_parameters())])
    # gradient of the mstft / dyn losses alone w.r.t. y_hat (pins the STFT backward kernel)
    yh = yd.clone().requires_grad_(True)
    M.multi_stft_loss(y, yh, ret_loss=True).backward()
    gold['grad_mstft_yhat'] = yh.grad.numpy()
    yh = yd.clone().requires_grad_(True)
    M.dynamic_loss(y, yh).backward()
    gold['grad_dyn_yhat'] = yh.grad.numpy()
    # gradient of the MTD generator-side losses w.r.t. y_hat through log|D| and angle(D)
    yh = yd.clone().requires_grad_(True)
    S, Sgh = M.multi_stft_loss(y, yh, ret_specs=True)
    lr, lg, fr, fg = mtd(S, Sgh)
    (M.generator_loss(lg, lr) + M.feature_loss(fr, fg) * hp.w_loss_fm).backward()
    gold['grad_mtd_yhat'] = yh.grad.numpy()

    # ------------------------------------------------------------------ two complete train steps per configuration
    def run_steps(use_msd, use_mpd, use_mtd, d_times, n_steps=2):
        torch.manual_seed(1234)
        g, msd, mpd, mtd = build_ref()
        for m in (g, msd, mpd, mtd):
            O.det_fill(m)
            m.train()
        ds = [d for d, u in ((msd, use_msd), (mpd, use_mpd), (mtd, use_mtd)) if u]
        import itertools
        og = torch.optim.AdamW(g.parameters(), hp.learning_rate_g, betas=[hp.adam_b1, hp.adam_b2])
        od = torch.optim.AdamW(itertools.chain(*[d.parameters() for d in ds]), hp.learning_rate_d,
                               betas=[hp.adam_b1, hp.adam_b2])
        x, y_tmpl, y = O.golden_inputs()
        rec = []
        after_first = {}

        def snapshot():
            out = {'g_stats': np.stack([stats(p) for _, p in sorted(g.named_parameters())])}
            for tag, d, u in (('msd', msd, use_msd), ('mpd', mpd, use_mpd), ('mtd', mtd, use_mtd)):
                if u:
                    out[f'{tag}_stats'] = np.stack([stats(p) for _, p in sorted(d.named_parameters())])
            return out

        for step_i in range(n_steps):
            y_hat = g(x, y_tmpl)
            yd = y_hat.detach()
            for _ in range(d_times):
                od.zero_grad()
                tot = 0
                if use_mtd:
                    S, Sg = M.multi_stft_loss(y, yd, ret_specs=True)
                if use_msd:
                    r, gg, _, _ = msd(y, yd)
                    tot = tot + M.discriminator_loss(r, gg)
                if use_mpd:
                    r, gg, _, _ = mpd(y, yd)
                    tot = tot + M.discriminator_loss(r, gg)
                if use_mtd:
                    r, gg, _, _ = mtd(S, Sg)
                    tot = tot + M.discriminator_loss(r, gg)
                if not torch.isnan(tot):
                    tot.backward()
                od.step()
            og.zero_grad()
            lm, (S, Sgh) = M.multi_stft_loss(y, y_hat, ret_loss=True, ret_specs=True)
            gt = lm * hp.w_loss_mstft + M.dynamic_loss(y, y_hat) * hp.w_loss_dyn
            for d, u, a, b in ((msd, use_msd, y, y_hat), (mpd, use_mpd, y, y_hat), (mtd, use_mtd, S, Sgh)):
                if u:
                    lr, lg, fr, fg = d(a, b)
                    gt = gt + M.generator_loss(lg, lr) + M.feature_loss(fr, fg) * hp.w_loss_fm
            if not torch.isnan(gt):
                gt.backward()
            og.step()
            rec.append([tot.item(), gt.item()])
            if step_i == 0:
                # (round 4) the parameters after the FIRST step: with MTD a one-step comparison can be made on the reference
                # run's side of the frame-0 phase cut (stft<n>_frame0_phase_*), a two-step one cannot (the second step's
                # generated wave is new) — tests/test_step_gpu.py::test_first_train_step_with_mtd_on_the_reference_branch
                after_first = {k + '1': v for k, v in snapshot().items()}
        res = {'losses': np.array(rec), **after_first,
               'g_stats': np.stack([stats(p) for _, p in sorted(g.named_parameters())])}
        for tag, d, u in (('msd', msd, use_msd), ('mpd', mpd, use_mpd), ('mtd', mtd, use_mtd)):
            if u:
                res[f'{tag}_stats'] = np.stack([stats(p) for _, p in sorted(d.named_parameters())])
        return res

    for name, cfg in (('cfg1', (True, False, False, 1)), ('cfg2', (True, True, False, 2)),
                      ('cfg4', (True, True, True, 2))):
        for k, v in run_steps(*cfg).items():
            gold[f'step_{name}_{k}'] = v

    np.savez_compressed(os.path.join(OUT, 'retunegan_b2_t8192.npz'), **gold)
    sz = os.path.getsize(os.path.join(OUT, 'retunegan_b2_t8192.npz'))
    print('wrote', len(gold), 'arrays,', sz, 'bytes')
    for k in ('loss_mstft', 'loss_dyn', 'loss_env', 'loss_sm', 'msd_d_loss', 'mpd_d_loss', 'mtd_d_loss',
              'loss_gen_all'):
        print(k, gold[k])
    print('sum y_hat', gold['y_hat'].astype(np.float64).sum())


if __name__ == '__main__':
    main()
