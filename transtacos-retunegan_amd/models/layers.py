"""Building blocks shared by the generator and the discriminators: the weight-normed conv parameter holder and the
base class that keeps all of a model's parameters in one flat GPU buffer (rtg/bank.py)."""
import math
import weakref

import numpy as np
import torch
import torch.nn as nn

from rtg import ops, tune, config
from rtg.bank import WeightBank
from rtg.lib import ACT_NONE, ACT_LRELU, ACT_TANH, RtgError, new_stream  # noqa: F401


class WNConv(nn.Module):
    """Parameters of one weight-normed convolution, named like torch.nn.utils.weight_norm names them
    (weight_g, weight_v, bias; call sites retunegan/models/generator.py:682-722, discrminator.py:37-45,156-163).
    kind 'conv'  : Conv1d, or a (k,1)/(k,1)-strided Conv2d whose trailing kernel dim is 1 (v keeps the 4-D shape)
    kind 'convT' : ConvTranspose1d (dim 0 of v is the INPUT channel axis, as in the reference's ups.N.weight_g)
    kind 'conv2d': Conv2d with (rows, cols) kernel / stride / padding pairs (StftDiscriminator, discrminator.py:255-262)."""

    def __init__(self, kind, cin, cout, k, stride=1, pad=0, dil=1, groups=1, out_pad=0, kdims=1, wt=False):
        super().__init__()
        self.kind, self.cin, self.cout, self.k = kind, cin, cout, k
        # wt ('conv2d' only): the layer is run on tensors whose last two axes are swapped — [B, C, W, H] for the reference's
        # [B, C, H, W] — so that the LONG axis (frequency, for the spectrogram discriminators) is the contiguous one the
        # kernels walk; k / stride / pad and the parameter shapes stay the reference's (rows, cols) = (H, W) pairs
        self.wt = bool(wt)
        assert not wt or kind == 'conv2d'
        self.stride, self.pad, self.dil, self.groups, self.out_pad = stride, pad, dil, groups, out_pad
        if kind == 'conv2d':                                   # k, stride, pad are (rows, cols) pairs
            assert groups == 1 and dil == 1
            ks = tuple(k)
        else:
            ks = (k,) if kdims == 1 else (k, 1)
        shape = ((cin, cout // groups) if kind == 'convT' else (cout, cin // groups)) + ks
        v = torch.empty(shape)
        nn.init.kaiming_uniform_(v, a=math.sqrt(5))            # torch's default conv init (reset_parameters)
        bound = 1.0 / math.sqrt(shape[1] * int(np.prod(ks)))
        b = torch.empty(cout).uniform_(-bound, bound)
        # registration order of torch.nn.utils.weight_norm(Conv*): the conv registers (weight, bias), weight_norm deletes
        # `weight` and appends weight_g, weight_v -> bias, weight_g, weight_v.  torch.optim state dicts index their
        # entries by this order (checkpoints `do_*`, train.py:263-273), so it is part of the drop-in contract.
        self.bias = nn.Parameter(b)
        self.weight_g = nn.Parameter(v.flatten(1).norm(dim=1).reshape((shape[0],) + (1,) * (len(shape) - 1)))
        self.weight_v = nn.Parameter(v)
        self._layer = None      # rtg.bank.ConvLayer, set when the owning model builds its bank

    def burn_init_rng(self):
        torch.empty(self.weight_v.shape).normal_(0, 1.0)

    def effective_weight(self):
        v = self.weight_v
        return v * (self.weight_g / v.flatten(1).norm(dim=1).reshape(self.weight_g.shape))

    def extra_repr(self):
        return (f'{self.kind} {self.cin}->{self.cout} k={self.k} s={self.stride} p={self.pad} d={self.dil} '
                f'g={self.groups}')


class BankedModel(nn.Module):
    """nn.Module whose WNConv children (and listed extra parameters) live in one WeightBank on the GPU."""

    _registry = weakref.WeakSet()      # lets train.AdamW(module.parameters(), ...) find the owning models

    def __init__(self):
        super().__init__()
        self._bank = None
        BankedModel._registry.add(self)

    def _extra_bank_params(self):
        return []

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._bank = None           # .to()/.cuda()/.float() re-materialise the parameters: rebuild the flat views
        return r

    def bank(self):
        dev = next(self.parameters()).device
        if dev.type != 'cuda':
            raise RtgError(f'{type(self).__name__}: parameters are on {dev}; the RetuneGAN hot path runs on the HIP '
                           'kernels only — call .to("cuda") (there is no CPU fallback)')
        if dev.index is not None and dev.index != torch.cuda.current_device():
            # librtg launches on the current stream of the CURRENT device (rtg/lib.py:current_stream_ptr); a model on
            # another device would be computed on by kernels queued to the wrong device's stream
            raise RtgError(f'{type(self).__name__} lives on {dev} but the current device is cuda:'
                           f'{torch.cuda.current_device()}: call torch.cuda.set_device({dev.index}) first (one process '
                           'per GPU; train.Trainer does it for its `dev`)')
        if self._bank is None or self._bank.device != dev or not self._bank.check_views():
            layers = [(n, m) for n, m in self.named_modules() if isinstance(m, WNConv)]
            self._bank = WeightBank(layers, self._extra_bank_params(), dev)
            self._bank.wgrad_side = bool(getattr(self, 'wgrad_side', False))
            for ly in self._bank.layers:
                ly.module._layer = ly
        return self._bank

    def token(self):
        return self.bank().prepare()

    def zero_grad(self, set_to_none=False):
        if self._bank is not None:
            self._bank.zero_grad()
        else:
            super().zero_grad(set_to_none=set_to_none)


import os

_FORK_STREAMS = {}
_FORK_MAX = 0                 # (> 0: that many forked streams at most, branches share them round robin; round 1: uncapped is best)
_FORK_PATH = [()]
if hasattr(torch.autograd.graph, 'set_warn_on_accumulate_grad_stream_mismatch'):
    # leaves are accumulated on the stream of their first use while forked branches run elsewhere: intended
    torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)


def _record(obj, stream):
    if torch.is_tensor(obj):
        obj.record_stream(stream)
        base = getattr(obj, '_rtg_base', None)
        if base is not None:
            base.record_stream(stream)
    elif isinstance(obj, (list, tuple)):
        for o in obj:
            _record(o, stream)
    elif isinstance(obj, dict):
        for o in obj.values():
            _record(o, stream)


def fork_join(fns):
    """Run independent sub-networks (the 3 MSD scales, the 4 MPD periods, the 3 MTD resolutions, the 3 ResBlock3
    branches) on separate HIP streams so that their kernels overlap: each of these launches only fills the chip for part
    of its duration (ramp-up, last partial wave of workgroups), a second queue fills the idle CUs.  Autograd replays the
    backward of every op on the stream of its forward, so the backward overlaps the same way.  Nested forks (the
    discriminator stacks forked by the trainer, their sub-discriminators forked inside) get streams of their own per
    branch: a stream never carries work of two different parents, which keeps the fork tree a tree (no false ordering
    between e.g. MSD scale 0 and MPD period 0, and a shape HIP graph capture accepts).  RTG_STREAMS=0 disables."""
    if len(fns) < 2 or config.get('RTG_STREAMS') == '0' or ops.PROFILE is not None or tune.ACTIVE:
        return [f() for f in fns]
    main = torch.cuda.current_stream()
    path = _FORK_PATH[0]
    pool = _FORK_STREAMS.setdefault((main.device, path), [])
    n_str = len(fns) if _FORK_MAX <= 0 else min(len(fns), _FORK_MAX)
    while len(pool) < n_str:
        pool.append(new_stream(device=main.device))      # (never a pooled torch stream: rtg/lib.py:new_stream)
    outs = []
    try:
        for i, f in enumerate(fns):
            s = pool[i % n_str]
            if i < n_str:
                s.wait_stream(main)
            _FORK_PATH[0] = path + (i,)
            with torch.cuda.stream(s):
                outs.append(f())
    finally:
        _FORK_PATH[0] = path
    for s in pool[:n_str]:
        main.wait_stream(s)
    _record(outs, main)          # produced on a side stream, consumed (and later freed) on the main one
    return outs


def conv(tok, m, x1, x2=None, res=None, pre_slope=1.0, act=ACT_NONE, act_slope=1.0, out_scale=1.0):
    return ops.conv(tok, m._layer, x1, x2, res, pre_slope, act, act_slope, out_scale)
