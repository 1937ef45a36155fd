"""MSD / MPD discriminator stacks (retunegan/models/discrminator.py — the reference's file name, typo included, is
kept so that `from models.discrminator import ...` keeps working) on the MI355X kernels.

Differences in execution, not in results:
  * D(real) and D(fake) share weights (discrminator.py:120-121): when both need the same treatment they run as ONE
    batch of 2B clips per layer; when the discriminator is frozen (generator update) the real half runs without
    autograd bookkeeping.
  * DiscriminatorP's [B,1,T/p,p] view + (k,1) Conv2d is executed as Conv1d over T/p with the period folded into the
    batch ([B*p, C, T/p]); feature maps are handed back as [B,C,T/p,p] views, logits in the reference's order.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F  # noqa: F401

import hparam as hp
from utils import *  # noqa: F401,F403
from utils import LRELU_SLOPE
from rtg import ops
from .layers import WNConv, BankedModel, conv, fork_join

PI = 3.14159265358979


def _frozen(model):
    return not (torch.is_grad_enabled() and any(p.requires_grad for p in model.parameters()))


def _run_stack(tok, convs, conv_post, x):
    """conv -> (feature map, before activation) -> lrelu(0.15) fused into the next conv's prologue."""
    fmap = []
    h = conv(tok, convs[0], x)
    fmap.append(h)
    for c in convs[1:]:
        h = conv(tok, c, h, pre_slope=LRELU_SLOPE)
        fmap.append(h)
    return conv(tok, conv_post, h, pre_slope=LRELU_SLOPE), fmap


# GROUPED: sibling sub-discriminators run layer by layer in grouped launches (rtg_conv1d_group) instead of on forked
# streams.  Measured on MI355X (config 2, batch 32): a grouped launch beats its members run back to back by 11-34 %, but
# the forked streams already overlap them as well — 42.7 ms/step grouped vs 42.3 forked — so forking it is (tests flip it).
GROUPED = False
PROLOGUES_FORKED = True      # run_stacks: the stacks' weight refresh and input preparation on forked streams


def _groupable(c):
    ly = c._layer
    return ly is not None and ly.kind == 'conv' and ly.cin // ly.groups > 1 and ly.cout // ly.groups > 1


def _run_stacks_lockstep(tok, subs, xs):
    """The sibling sub-discriminators `subs` (same architecture, own weights) on their inputs `xs`, layer by layer: every
    layer position is ONE grouped launch (rtg_conv1d_group) forward and ONE backward-data instead of len(subs) small
    ones; the 1-channel first / last layers run on the bandwidth kernels as before.  Returns [(logit, fmap)] per sub."""
    n = len(subs)
    hs = list(xs)
    fmaps = [[] for _ in range(n)]
    for li in range(len(subs[0].convs)):
        cs = [d.convs[li] for d in subs]
        slope = LRELU_SLOPE if li > 0 else 1.0
        if all(_groupable(c) for c in cs):
            hs = ops.group_conv(tok, [c._layer for c in cs], hs, pre_slope=slope)
        else:
            hs = [conv(tok, c, h, pre_slope=slope) for c, h in zip(cs, hs)]
        for f, h in zip(fmaps, hs):
            f.append(h)
    logits = [conv(tok, d.conv_post, h, pre_slope=LRELU_SLOPE) for d, h in zip(subs, hs)]
    return list(zip(logits, fmaps))


class DiscriminatorS(nn.Module):
    """discrminator.py:17-101, active branch 'MelGAN_small' (:36-45).  `use_sn` is ignored by the reference too."""

    def __init__(self, use_sn=False):
        super().__init__()
        spec = [(1, 32, 15, 1, 7, 1), (32, 64, 41, 2, 20, 4), (64, 128, 41, 2, 20, 8), (128, 512, 41, 4, 20, 32),
                (512, 512, 41, 4, 20, 64), (512, 512, 5, 1, 2, 1)]
        self.convs = nn.ModuleList([WNConv('conv', ci, co, k, stride=s, pad=p, groups=g) for ci, co, k, s, p, g in spec])
        self.conv_post = WNConv('conv', 512, 1, 3, pad=1)

    def pre(self, x):
        return x

    def post(self, logit, fmap, B):
        return torch.flatten(logit, 1, -1), fmap

    def run(self, tok, x):
        logit, fmap = _run_stack(tok, self.convs, self.conv_post, x)
        return torch.flatten(logit, 1, -1), fmap


class DiscriminatorP(nn.Module):
    """discrminator.py:132-222, active branch 'HiFiGAN_small' (:155-163)."""

    def __init__(self, period):
        super().__init__()
        self.period = period
        ch = [1, 32, 128, 256, 512]
        self.convs = nn.ModuleList([WNConv('conv', ch[i], ch[i + 1], 5, stride=3, pad=2, kdims=2) for i in range(4)]
                                   + [WNConv('conv', 512, 512, 5, stride=1, pad=2, kdims=2)])
        self.conv_post = WNConv('conv', 512, 1, 3, pad=1, kdims=2)

    def run(self, tok, x):
        """x [B,1,T] -> (logits [B, H'*p], feature maps as [B,C,H,p] views whose `_rtg_base` is the contiguous
        [B*p,C,H] tensor the kernels produced)."""
        logit, fmap = _run_stack(tok, self.convs, self.conv_post, self.pre(x))
        return self.post(logit, fmap, x.shape[0])

    def pre(self, x):
        return ops.PeriodFoldFn.apply(x, self.period)

    def post(self, logit, fmap, B):
        p = self.period
        views = []
        for f in fmap:
            v = f.view(B, p, f.shape[1], f.shape[2]).permute(0, 2, 3, 1)
            v._rtg_base = f
            views.append(v)
        lg = logit.view(B, p, logit.shape[2]).permute(0, 2, 1).reshape(B, -1)
        lg._rtg_base = logit
        return lg, views


class _Prepared:
    """a sub-discriminator input that already is in the layout its layers run on (StftDiscriminator.pre)"""

    def __init__(self, t):
        self.t = t
        self.shape = t.shape


def _split(t, B):
    """[2B, ...] batch of (real, fake) -> two halves that remember their contiguous kernel-side storage."""
    base = getattr(t, '_rtg_base', None)
    r, g = t[:B], t[B:]
    if base is not None:
        n = base.shape[0] // 2
        r._rtg_base, g._rtg_base = base[:n], base[n:]
    # the halves remember the tensor they are halves of: discriminator_loss then runs as ONE node over the whole tensors
    whole = base if base is not None else t
    r._rtg_pair, g._rtg_pair = (whole, 0), (whole, 1)
    return r, g


# Generator step (D frozen): real and generated clips of a sub-discriminator run as ONE 2B-clip launch per layer
# (ops.PairConvFn), the backward-data only over the generated half (False: two separate passes; 38.8 -> 38.4 ms in round 1).
PAIRED = True


def _pairable(d):
    return (hasattr(d, 'pre') and hasattr(d, 'post') and
            all(c._layer is not None and c._layer.kind in ('conv', 'conv2d') for c in list(d.convs) + [d.conv_post]))


def _run_pair(d, tok, x_real, x_fake):
    B = x_fake.shape[0]
    with torch.no_grad():
        xr = d.pre(x_real)
    hr, hg = ops.pair_entry(xr, d.pre(x_fake))
    fr, fg = [], []
    hr, hg = ops.pair_conv(tok, d.convs[0]._layer, hr, hg)
    # every feature map has two readers, the next layer and the feature-matching loss: the loss reads the copy the next
    # layer hands through (`tap`), so the two gradients meet in that layer's backward-data epilogue
    for c in list(d.convs[1:]) + [d.conv_post]:
        if ops._is_bf(hg) and not c._layer.maps_bf:
            # a bf16 (encoded) feature map in front of a layer without a bf16 input path (conv_post): ONE decode of the
            # 2B-clip buffer serves the layer and the feature-matching loss
            hr, hg = ops.PairDecodeFn.apply(hr, hg)
        nr, ng, tap = ops.pair_conv(tok, c._layer, hr, hg, pre_slope=LRELU_SLOPE, tap=True)
        fr.append(hr)
        fg.append(tap)
        hr, hg = nr, ng
    lr, lg = hr, hg
    lr, fr = d.post(lr, fr, B)
    lg, fg = d.post(lg, fg, B)
    return lr, lg, fr, fg


def _sub_runner(d, tok, inp, frozen):
    """closure running one sub-discriminator on (real, fake): separately when D is frozen (real half without autograd),
    as one 2B batch otherwise.  Returns (logit_r, logit_g, fmap_r, fmap_g)."""
    def run():
        # The inputs may have been prepared on ANOTHER forked stream (run_stacks forks the stacks' prologues): tell the caching
        # allocator that this branch's stream reads them, so that their memory is not handed out again — to that other stream,
        # which never waits for this one — before the branch (and, later, its backward) is done with it.
        cur = torch.cuda.current_stream()
        for t in inp:
            tt = t.t if isinstance(t, _Prepared) else t
            if torch.is_tensor(tt) and tt.is_cuda:
                tt.record_stream(cur)
        if frozen and PAIRED and _pairable(d):
            return _run_pair(d, tok, inp[0], inp[1])
        if frozen:
            with torch.no_grad():
                lr, fr = d.run(tok, inp[0])
            lg, fg = d.run(tok, inp[1])
            return lr, lg, fr, fg
        B = inp[0].shape[0] // 2
        l2, f2 = d.run(tok, inp[0].t, prepared=True) if isinstance(inp[0], _Prepared) else d.run(tok, inp[0])
        lr, lg = _split(l2, B)
        fs = [_split(f, B) for f in f2]
        return lr, lg, [a for a, _ in fs], [b for _, b in fs]
    return run


def _collect(outs):
    return ([o[0] for o in outs], [o[1] for o in outs], [o[2] for o in outs], [o[3] for o in outs])


class _MultiBase(BankedModel):
    """`branches(a, b)` refreshes the packed weights and returns one closure per sub-discriminator; `forward` runs them
    on forked streams.  `run_stacks` forks the sub-discriminators of SEVERAL stacks in one flat fork (what the trainer
    uses: 7 - 10 independent streams off the current one instead of a fork inside a fork)."""

    def forward(self, a, b):
        return _collect(fork_join(self.branches(a, b)))

    def sub_inputs(self, a, b, frozen):
        """per sub-discriminator: [real, fake] (frozen) or [cat(real, fake)]"""
        raise NotImplementedError

    def run_grouped(self, a, b):
        """the whole stack layer by layer in grouped launches; same return value as forward()"""
        tok = self.token()
        frozen = _frozen(self) and not (a[0] if isinstance(a, (list, tuple)) else a).requires_grad
        subs = list(self.discriminators)
        inputs = self.sub_inputs(a, b, frozen)

        def lockstep(xs, batch):
            res = _run_stacks_lockstep(tok, subs, [d.pre(x) for d, x in zip(subs, xs)])
            return [d.post(lg, fm, batch) for d, (lg, fm) in zip(subs, res)]
        if frozen:
            B = inputs[0][0].shape[0]
            with torch.no_grad():
                real = lockstep([inp[0] for inp in inputs], B)
            fake = lockstep([inp[1] for inp in inputs], B)
            return ([r[0] for r in real], [f[0] for f in fake], [r[1] for r in real], [f[1] for f in fake])
        B2 = inputs[0][0].shape[0]
        both = lockstep([inp[0] for inp in inputs], B2)
        outs = []
        for l2, f2 in both:
            lr, lg = _split(l2, B2 // 2)
            fs = [_split(f, B2 // 2) for f in f2]
            outs.append((lr, lg, [x for x, _ in fs], [y for _, y in fs]))
        return _collect(outs)


def run_stacks(calls, extra=()):
    """calls: [(stack, a, b)] -> [(logits_r, logits_g, fmaps_r, fmaps_g)] per stack.  A stack whose sub-discriminators
    share an architecture of 1-D convs (MSD, MPD) runs layer by layer in grouped launches (one branch of the fork); the
    sub-discriminators of the others (MTD) are forked side by side — one flat fork in all.
    extra: thunks that depend on no stack (the generator step's spectral / dynamic losses): more branches of the same fork,
    their results follow the stacks' in the returned list."""
    brs, spans = [], []
    # The stacks' prologues — weight-norm scales + weight pack of the stack's bank, the input concatenations / pooling pyramid —
    # side by side on forked streams as well (PROLOGUES_FORKED): bandwidth launches that ran one after the other on the main
    # stream in front of the fork, with nothing else on the chip (round 6, kernel trace of the replayed step: 0.2 ms per pass;
    # same-box A/B of the config-2 step 25.97 / 25.79 ms forked against 26.15 / 26.19), and a bank's weight-norm backward runs on
    # the stream of its pack.  The inputs a prologue allocates are read by the sub-discriminators' branches on OTHER streams:
    # every branch records itself as a reader (_sub_runner) — without that they returned to the prologue stream's allocator pool
    # while those readers could still run (the 1-rank RCCL bit-identity test of the full stack caught the first version).
    plain = [c for c in calls if not (GROUPED and getattr(c[0], 'groupable', False))]
    pre = {}
    if PROLOGUES_FORKED and len(plain) > 1:
        for c, bs in zip(plain, fork_join([(lambda st=st, a=a, b=b: st.branches(a, b)) for st, a, b in plain])):
            pre[id(c[0])] = bs
    for stack, a, b in calls:
        if GROUPED and getattr(stack, 'groupable', False):
            bs = [(lambda st=stack, a=a, b=b: st.run_grouped(a, b))]
            spans.append((len(brs), None))
        else:
            bs = pre[id(stack)] if id(stack) in pre else stack.branches(a, b)
            spans.append((len(brs), len(brs) + len(bs)))
        brs += bs
    n_br = len(brs)
    outs = fork_join(brs + list(extra))
    return [outs[lo] if hi is None else _collect(outs[lo:hi]) for lo, hi in spans] + outs[n_br:]


class MultiScaleDiscriminator(_MultiBase):
    """discrminator.py:104-129."""

    def __init__(self):
        super().__init__()
        self.discriminators = nn.ModuleList([DiscriminatorS(use_sn=i == 0) for i in range(hp.msd_layers)])
        assert hp.downsample_pool_k == 4, 'rtg_avgpool4s2 implements AvgPool1d(4, 2, 1) (hparam.py:91)'

    groupable = True

    def branches(self, y, y_hat, tok=None):
        tok = self.token() if tok is None else tok
        frozen = _frozen(self) and not y.requires_grad
        inputs = self.sub_inputs(y, y_hat, frozen)
        return [_sub_runner(d, tok, inp, frozen) for d, inp in zip(self.discriminators, inputs)]

    def sub_inputs(self, y, y_hat, frozen):
        # inputs of the three scales: y, AvgPool(y), AvgPool(AvgPool(y)) (discrminator.py:126-127)
        xs = [y, y_hat] if frozen else [torch.cat([y, y_hat], dim=0)]
        inputs = []
        for i in range(len(self.discriminators)):
            inputs.append(list(xs))
            if i != len(self.discriminators) - 1:
                if frozen:
                    with torch.no_grad():
                        x0 = ops.AvgPoolFn.apply(xs[0])
                    xs = [x0, ops.AvgPoolFn.apply(xs[1])]
                else:
                    xs = [ops.AvgPoolFn.apply(xs[0])]
        return inputs


class MultiPeriodDiscriminator(_MultiBase):
    """discrminator.py:225-244."""

    def __init__(self):
        super().__init__()
        self.discriminators = nn.ModuleList([DiscriminatorP(p) for p in hp.mpd_periods])

    groupable = True

    def sub_inputs(self, y, y_hat, frozen):
        inp = [y, y_hat] if frozen else [torch.cat([y, y_hat], dim=0)]
        return [inp for _ in self.discriminators]

    def branches(self, y, y_hat, tok=None):
        tok = self.token() if tok is None else tok
        frozen = _frozen(self) and not y.requires_grad
        return [_sub_runner(d, tok, inp, frozen) for d, inp in zip(self.discriminators, self.sub_inputs(y, y_hat, frozen))]


# The spectrogram discriminators run along the FREQUENCY axis (round 5): their maps live in HBM as [B, C, frames, F] — rows of
# 1025 .. 8 bins instead of 137 .. 5 frames for the kernels to walk — and the layers are built with WNConv(wt=True): same
# parameters, shapes and state-dict keys, the kernel axes swapped when the weights are packed.  What goes in and comes out
# keeps the reference's [B, C, F, frames] shape (transposed views; logits in the reference's order).
MTD_ALONG_FREQ = True        # (False: the reference's layout [B, C, F, frames]; same-box A/B in DESIGN.md: equal step time)


class StftDiscriminator(nn.Module):
    """discrminator.py:247-308: five strided Conv2d layers + conv_post over [log|D|, phase/PI] ([B,2,F,frames])."""

    def __init__(self, i, ch=2):
        super().__init__()
        spec = [(ch, 32, (3, 3), (2, 1), (1, 1)), (32, 64, (3, 3), (2, 2), (1, 1)), (64, 256, (5, 3), (3, 2), (2, 1)),
                (256, 512, (5, 3), (3, 2), (2, 1)), (512, 512, (3, 3), (1, 1), (1, 1))]
        self.wt = MTD_ALONG_FREQ
        self.convs = nn.ModuleList([WNConv('conv2d', ci, co, k, stride=s, pad=p, wt=self.wt) for ci, co, k, s, p in spec])
        self.conv_post = WNConv('conv2d', 512, 1, (3, 3), stride=(1, 1), pad=(1, 1), wt=self.wt)
        for c in [*self.convs, self.conv_post]:
            c.burn_init_rng()       # self.convs.apply(init_weights); self.conv_post.apply(init_weights) (:264-265)

    def pre(self, x):
        """[B,2,F,frames] map -> the tensor the layers run on: [B,2,frames,F], frequency contiguous (a view of what
        audio.stft_mel_spec made; anything else is copied)"""
        return x.transpose(2, 3).contiguous() if self.wt else x

    def run(self, tok, x, prepared=False):
        logit, fmap = _run_stack(tok, self.convs, self.conv_post, x if prepared else self.pre(x))
        return self.post(logit, fmap, logit.shape[0])

    def post(self, logit, fmap, B):
        """kernel-side logits / feature maps -> what the reference returns (its layout, as views)"""
        if not self.wt:
            return torch.flatten(logit, 1, -1), fmap
        views = []
        for f in fmap:
            v = f.transpose(2, 3)
            v._rtg_base = f
            views.append(v)
        return torch.flatten(logit.transpose(2, 3), 1, -1), views


class MultiStftDiscriminator(_MultiBase):
    """discrminator.py:311-330: one StftDiscriminator per STFT resolution, zipped with the spec lists."""

    def __init__(self):
        super().__init__()
        self.discriminators = nn.ModuleList([StftDiscriminator(i) for i in range(len(hp.multi_stft_params))])

    groupable = False      # 2-D convs: not served by the grouped launch yet

    def branches(self, phs, ph_hats, tok=None):
        tok = self.token() if tok is None else tok
        frozen = _frozen(self) and not phs[0].requires_grad
        # (the real / generated pair is concatenated in the layout the layers run on: a coalesced copy)
        inputs = [[ph, ph_hat] if frozen else [_Prepared(torch.cat([d.pre(ph), d.pre(ph_hat)], dim=0))]
                  for d, ph, ph_hat in zip(self.discriminators, phs, ph_hats)]
        return [_sub_runner(d, tok, inp, frozen) for d, inp in zip(self.discriminators, inputs)]
