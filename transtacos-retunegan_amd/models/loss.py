"""Loss functions of retunegan/models/loss.py on the MI355X kernels (same names, signatures and return values).

multi_stft_loss runs the fused STFT kernel (mel, log-magnitude and phase/PI epilogues in one pass); all L1 / LSGAN
reductions of a whole list of tensors are ONE multi-tensor launch each."""
import torch
import torch.nn as nn  # noqa: F401
import torch.nn.functional as F  # noqa: F401

import hparam as hp
from audio import get_stft_torch, stft_mel_spec, multi_stft_mel_spec  # noqa: F401
from utils import PI  # noqa: F401
from rtg import ops
from rtg.lib import LOSS_L1, LOSS_L1_L1LOG, LOSS_L1_ENC, LOSS_MSE_TARGET, LOSS_MSE_REL, MAX_LOSS_JOBS, RtgError  # noqa: F401

_real_cache = {}


class stft_cache:
    """Context manager: inside it, the spectra of the REAL wave are computed once and reused by every
    multi_stft_loss call (the reference recomputes them 3x per step: its own TODO at loss.py:32)."""

    def __enter__(self):
        _real_cache.clear()
        _real_cache['on'] = True
        return self

    def __exit__(self, *a):
        _real_cache.clear()


def _real_key(y):
    return (y.data_ptr(), tuple(y.shape), y._version)


def _real_hit(y, want_spec):
    """the cached spectra of the real wave (inside a stft_cache context), or None"""
    if _real_cache.get('on'):
        hit = _real_cache.get(_real_key(y))
        if hit is not None and (not want_spec or hit[1][0] is not None):
            return hit
    return None


def multi_stft_loss(y, y_g, ret_loss=False, ret_specs=False):
    """loss.py:22-62.  y, y_g: [B,1,T] or [B,T].  Returns loss, (stft_r, stft_g) per the two flags; spec tensors are
    [B,2,F,frames] = stack([log S, P/PI], dim=1) (phd_input == 'stft', hparam.py:83)."""
    if not (ret_loss or ret_specs):
        raise ValueError('multi_stft_loss: nothing requested')
    if hp.phd_input != 'stft':
        raise RtgError("only phd_input == 'stft' (hparam.py:83) is on the path")
    if y.dim() == 3:
        y, y_g = y.squeeze(1), y_g.squeeze(1)
    if y.requires_grad:
        raise RtgError('multi_stft_loss: the real wave is treated as a constant (as in retunegan/train.py)')
    # every resolution of the generated wave — and of the real one unless its spectra are cached — in ONE launch
    # (audio.multi_stft_mel_spec; rounds 1-5: a launch per resolution and wave, six of 14-17 us per loss call)
    hit = _real_hit(y, ret_specs)
    if hit is not None:
        (mels_r, specs_r), (mels_g, specs_g) = hit, multi_stft_mel_spec(y_g, hp.multi_stft_params, ret_specs)
    else:
        (mels_g, specs_g), (mels_r, specs_r) = multi_stft_mel_spec(y_g, hp.multi_stft_params, ret_specs, y_real=y)
        if _real_cache.get('on'):
            _real_cache[_real_key(y)] = (mels_r, specs_r)
    loss = None
    if ret_loss:
        n = len(hp.multi_stft_params)
        loss = ops.multi_loss(LOSS_L1_L1LOG, mels_r, mels_g, [1.0 / n] * n)
    if ret_loss and ret_specs:
        return loss, (specs_r, specs_g)
    if ret_loss:
        return loss
    return specs_r, specs_g


def envelope_loss(y, y_g):
    """loss.py:66-72: mean |max160(y) - max160(y_g)| + mean |max160(-y) - max160(-y_g)| (off by default, hparam.py:88)."""
    return ops.DynLossFn.apply(y, y_g, hp.envelope_pool_k, True)


def dynamic_loss(y, y_g):
    """loss.py:76-82."""
    return ops.DynLossFn.apply(y, y_g, hp.envelope_pool_k)


def strip_mirror_loss(y):
    """loss.py:86-98 (off by default, hparam.py:87)."""
    return ops.StripMirrorFn.apply(y)


def _base(t):
    return getattr(t, '_rtg_base', t)


def discriminator_loss(disc_r, disc_g):
    """loss.py:102-125: sum_k mean((1 - dr_k)^2) + mean(dg_k^2); with hparam.relative_gan_loss the real term is
    mean((1 - (dr_k - dg_k.detach()))^2) (loss.py:116)."""
    pr, pg = [getattr(d, '_rtg_pair', None) for d in disc_r], [getattr(d, '_rtg_pair', None) for d in disc_g]
    if len(pr) == len(pg) and 2 * len(pr) <= MAX_LOSS_JOBS and \
            all(a is not None and b is not None and a[0] is b[0] and (a[1], b[1]) == (0, 1) for a, b in zip(pr, pg)):
        return ops.pair_loss([a[0] for a in pr], relative=hp.relative_gan_loss)
    rs, gs = [_base(d) for d in disc_r], [_base(d) for d in disc_g]
    if hp.relative_gan_loss:
        real = ops.multi_loss(LOSS_MSE_REL, rs, [g.detach() for g in gs], target=1.0)
    else:
        real = ops.multi_loss(LOSS_MSE_TARGET, rs, target=1.0)
    return real + ops.multi_loss(LOSS_MSE_TARGET, gs, target=0.0)


def generator_loss(disc_g, disc_r):
    """loss.py:129-145: sum_k mean((1 - dg_k)^2); with hparam.relative_gan_loss mean((dg_k - dr_k.detach())^2)
    (loss.py:136)."""
    gs = [_base(d) for d in disc_g]
    if hp.relative_gan_loss:
        return ops.multi_loss(LOSS_MSE_REL, gs, [_base(d).detach() for d in disc_r], target=0.0)
    return ops.multi_loss(LOSS_MSE_TARGET, gs, target=1.0)


def feature_loss(fmap_r, fmap_g):
    """loss.py:149-156: sum over every feature map of mean |r - g| (an element-order-invariant reduction, so the
    kernel-side layout of MPD maps is used directly)."""
    rs = [_base(r) for dr in fmap_r for r in dr]
    gs = [_base(g) for dg in fmap_g for g in dg]
    # bf16 feature maps (hparam.bf16_maps) are stored leaky-relu encoded: their pairs go through the decoding loss kind
    if any((r.dtype == torch.bfloat16) != (g.dtype == torch.bfloat16) for r, g in zip(rs, gs)):
        # (a pair of one encoded and one plain map — e.g. the real half still encoded, the generated one already decoded for
        # conv_post: everything through the fp32 feature maps)
        rs, gs = [ops.decode(r) for r in rs], [ops.decode(g) for g in gs]
    enc = [i for i, (r, g) in enumerate(zip(rs, gs)) if r.dtype == torch.bfloat16 and g.dtype == torch.bfloat16]
    if not enc:
        return ops.multi_loss(LOSS_L1, rs, gs)
    plain = [i for i in range(len(rs)) if i not in set(enc)]
    terms = [ops.multi_loss(LOSS_L1_ENC, [rs[i] for i in enc], [gs[i] for i in enc], target=ops.ENC_SLOPE)]
    if plain:
        terms.append(ops.multi_loss(LOSS_L1, [rs[i] for i in plain], [gs[i] for i in plain]))
    return ops.weighted_sum(terms)
