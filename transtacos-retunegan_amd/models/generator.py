"""UNet-G (`Generator_RefineGAN_small`, retunegan/models/generator.py:670-796) on the MI355X kernels.

Same constructor signature, module tree and state-dict keys as the reference; the forward is re-expressed as a chain
of fused conv launches (activation prologues/epilogues, residual adds and torch.cat fused into rtg_conv1d)."""
import torch
import torch.nn as nn
import torch.nn.functional as F  # noqa: F401  (re-exported like the reference's star import does)

import hparam as hp
from utils import *  # noqa: F401,F403
from utils import LRELU_SLOPE
from rtg import ops
from .layers import WNConv, BankedModel, conv, fork_join, ACT_LRELU, ACT_TANH

device = 'cuda' if torch.cuda.is_available() else 'cpu'


def noise_seed(seed, rank, calls):
    """64-bit counter-RNG key of one GaussianNoise call: the run seed, the data-parallel rank (a
    different stream per rank — the reference draws independent noise per sample) and the per-process call counter."""
    return ((seed * 0x9E3779B1 + calls) ^ (rank * 0x632BE59BD9B4E019)) & (2 ** 63 - 1)


class GaussianNoise(nn.Module):
    """generator.py:19-30: x + U[0,1)*w followed by leaky_relu(0.15); one shared trainable scalar w = 1e-6."""

    def __init__(self):
        super().__init__()
        self.w = nn.Parameter(torch.FloatTensor([1e-6]), requires_grad=True)
        self.seed = hp.randseed
        self.rank = None          # data-parallel rank, mixed into the seed: every rank draws its own noise field
        self.calls = 0
        self.salt = None          # optional device word mixed into the seed (changes every optimizer step)

    def forward(self, x, u=None):
        self.calls += 1
        if self.rank is None:
            import torch.distributed as dist
            self.rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        return ops.NoiseFn.apply(x, self.w, u, LRELU_SLOPE, noise_seed(self.seed, self.rank, self.calls), self.salt)


class _Seq(nn.Module):
    """nn.Sequential-compatible child naming for ResidualStack ('res_1.1', 'res_1.3': the convs sit at the odd
    indices of the reference's Sequential(LeakyReLU, Conv1d, LeakyReLU, Conv1d), generator.py:39-56)."""

    def __init__(self, c1, c3):
        super().__init__()
        self.add_module('1', c1)
        self.add_module('3', c3)


class ResidualStack(nn.Module):
    """generator.py:33-77.  x <- x + conv_d1(lrelu_.01(conv_dil(lrelu_.01(x)))) for dil in (1, 3, 9)."""

    def __init__(self, channels, k=3):
        super().__init__()
        self.channels = channels
        for name, d in (('res_1', 1), ('res_2', 3), ('res_3', 9)):
            self.add_module(name, _Seq(WNConv('conv', channels, channels, k, pad=get_same_padding(3, d), dil=d),
                                       WNConv('conv', channels, channels, k, pad=get_same_padding(3))))

    def run(self, tok, x, final_act_slope=None):
        blocks = (self.res_1, self.res_2, self.res_3)
        lys = [getattr(blk, n)._layer for blk in blocks for n in ('1', '3')]
        if ops.resstack_shape_ok(lys, x):    # one autograd node; clips that fit in LDS: one launch per direction
            return ops.resstack(tok, lys, x, 0.01, final_act_slope)
        for i, blk in enumerate(blocks):
            r = conv(tok, getattr(blk, '1'), x, pre_slope=0.01)
            if i == len(blocks) - 1 and final_act_slope is not None:
                x = conv(tok, getattr(blk, '3'), r, res=x, pre_slope=0.01, act=ACT_LRELU, act_slope=final_act_slope)
            else:
                x = conv(tok, getattr(blk, '3'), r, res=x, pre_slope=0.01)
        return x


class ResBlock3(nn.Module):
    """generator.py:133-155: three times x <- conv_d(lrelu_.15(x)) + x."""

    def __init__(self, channels, kernel_size=3, dilation=(1, 3, 5)):
        super().__init__()
        self.convs = nn.ModuleList([WNConv('conv', channels, channels, kernel_size, dil=d,
                                           pad=get_padding(kernel_size, d)) for d in dilation])
        for c in self.convs:
            c.burn_init_rng()       # self.convs.apply(init_weights), generator.py:143

    def run(self, tok, x):
        for c in self.convs:
            x = conv(tok, c, x, res=x, pre_slope=LRELU_SLOPE)
        return x


class ResBlock(nn.Module):
    """generator.py:109-131 (HiFiGAN block of the full-size RefineGAN): twice x <- conv_d(lrelu_.15(x)) + x."""

    def __init__(self, channels, kernel_size=3, dilation=(1, 3)):
        super().__init__()
        self.convs = nn.ModuleList([WNConv('conv', channels, channels, kernel_size, dil=d,
                                           pad=get_padding(kernel_size, d)) for d in dilation])
        for c in self.convs:
            c.burn_init_rng()       # self.convs.apply(init_weights), generator.py:119

    def run(self, tok, x, final_act_slope=None):
        n = len(self.convs)
        for i, c in enumerate(self.convs):
            if i == n - 1 and final_act_slope is not None:
                x = conv(tok, c, x, res=x, pre_slope=LRELU_SLOPE, act=ACT_LRELU, act_slope=final_act_slope)
            else:
                x = conv(tok, c, x, res=x, pre_slope=LRELU_SLOPE)
        return x


def run_branches(tok, blocks, z):
    """the parallel ResBlock branches of one decoder stage on the same input z (generator.py:776-778, 650-653): conv by
    conv as grouped launches where that pays (ops.mrf_group_ok), else each branch on its own stream"""
    n_conv = len(blocks[0].convs)
    if all(len(b.convs) == n_conv for b in blocks) and \
            all(ops.mrf_group_ok([b.convs[c]._layer for b in blocks], z) for c in range(n_conv)):
        xs = [z] * len(blocks)
        for c in range(n_conv):
            xs = ops.group_conv(tok, [b.convs[c]._layer for b in blocks], xs, pre_slope=LRELU_SLOPE, res_self=True)
        return xs
    return fork_join([(lambda blk=blk, zz=z: blk.run(tok, zz)) for blk in blocks])


class _Mean3(torch.autograd.Function):
    """(a + b + c) / 3 — the average of the three ResBlock3 branches (generator.py:776-778)."""

    @staticmethod
    def forward(ctx, a, b, c):
        import ctypes as C
        from rtg.lib import lib, check, current_stream_ptr as _sp
        out = torch.empty_like(a)
        st = _sp()
        third = 1.0 / 3
        check(lib.rtg_axpby(C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(out.data_ptr()), a.numel(),
                            third, third, 0, st), 'mean3')
        check(lib.rtg_axpby(C.c_void_p(c.data_ptr()), None, C.c_void_p(out.data_ptr()), a.numel(), third, 0.0, 1, st),
              'mean3')
        return out

    @staticmethod
    def backward(ctx, dy):
        import ctypes as C
        from rtg.lib import lib, check, current_stream_ptr as _sp
        dy = dy.contiguous()
        g = torch.empty_like(dy)
        check(lib.rtg_axpby(C.c_void_p(dy.data_ptr()), None, C.c_void_p(g.data_ptr()), dy.numel(), 1.0 / 3, 0.0, 0,
                            _sp()), 'mean3 bwd')
        return g, g, g


class Generator_RefineGAN_small(BankedModel):
    """Zero-argument constructor, `forward(mel[B,80,T/256], wav_tmpl[B,1,T]) -> wav[B,1,T]`
    (retunegan/train.py:48,126)."""

    # the backward is one serial chain of small launches: weight gradients run beside it (rtg.ops.wgrad_side)
    wgrad_side = True

    def __init__(self):
        super().__init__()
        self.num_kernels = len(hp.resblock_kernel_sizes)
        self.num_upsamples = len(hp.upsample_rates)
        self.n_layer = self.num_upsamples
        ch = 32
        self.conv_pre = WNConv('conv', 1, ch // 2, 7, pad=3)
        self.downs = nn.ModuleList([
            WNConv('conv', ch * 2 ** i // 2, ch * 2 ** (i + 1) // 2, k, stride=u, pad=k // 2)
            for i, (u, k) in enumerate(zip(hp.upsample_rates[::-1], hp.upsample_kernel_sizes[::-1]))])
        self.resblock = nn.ModuleList([ResidualStack(ch * 2 ** i) for i in range(len(self.downs))])
        uic = hp.upsample_initial_channel
        self.conv_fuse = WNConv('conv', hp.n_mel + uic // 2, uic, 7, pad=3)
        self.ups = nn.ModuleList([
            WNConv('convT', uic // (2 ** i), uic // (2 ** (i + 1)), k, stride=u, pad=k // 2, out_pad=u - 1)
            for i, (u, k) in enumerate(zip(hp.upsample_rates, hp.upsample_kernel_sizes))])
        self.resblocks = nn.ModuleList([ResBlock3(c, k, [9, 3, 1]) for c in (128, 64, 32) for k in (3, 5, 7)])
        self.merge = nn.ModuleList([WNConv('conv', 128 + 64, 128, 7, pad=3), WNConv('conv', 64 + 32, 64, 7, pad=3),
                                    WNConv('conv', 32 + 16, 32, 7, pad=3)])
        self.conv_post = WNConv('conv', ch, 1, 7, pad=3)
        self.noise = GaussianNoise()
        for m in [self.conv_pre, self.conv_fuse, self.conv_post, *self.downs, *self.merge, *self.ups]:
            m.burn_init_rng()       # the .apply(init_weights) calls of generator.py:727-732
        self._wn_removed = False

    def _extra_bank_params(self):
        return [('noise.w', self.noise.w)]

    def forward(self, x, y, noise_list=None):
        """noise_list: optional six pre-drawn U[0,1) tensors (parity tests); by default drawn on the device."""
        tok = self.token()
        nz = (lambda i: None) if noise_list is None else (lambda i: noise_list[i])
        o = []
        y = conv(tok, self.conv_pre, y, act=ACT_LRELU, act_slope=LRELU_SLOPE)     # lrelu(conv_pre(y)) = skip o[0]
        for i in range(self.n_layer):
            o.append(y)
            y = conv(tok, self.downs[i], y)
            y = self.resblock[i].run(tok, y, final_act_slope=LRELU_SLOPE)          # next skip / fuse input
        z = conv(tok, self.conv_fuse, x, y)                                        # cat([mel, y]) fused
        for i in range(self.n_layer):
            z = conv(tok, self.ups[i], z, pre_slope=LRELU_SLOPE)
            z = conv(tok, self.merge[i], z, o[self.n_layer - i - 1])               # cat([z, skip]) fused
            z = self.noise(z, nz(2 * i))
            nk = self.num_kernels
            z = _Mean3.apply(*run_branches(tok, [self.resblocks[i * nk + j] for j in range(nk)], z))
            z = self.noise(z, nz(2 * i + 1))
        return conv(tok, self.conv_post, z, pre_slope=LRELU_SLOPE, act=ACT_TANH)

    def remove_weight_norm(self):
        """generator.py:790-796 (inference, retunegan/server.py:81).  The kernels always consume g*v/||v||, so the
        result of the forward is unchanged; kept for API compatibility."""
        self._wn_removed = True


class Generator_RefineGAN(BankedModel):
    """The full-size RefineGAN (retunegan/models/generator.py:560-667; `hparam.generator_ver = 'RefineGAN'`): same UNet
    at twice the channels with 2-conv ResBlocks, the mel entering through its own conv_pre and the encoder output
    concatenated to it before the first upsampling.  Same constructor / forward signature and state-dict keys."""

    # the backward is one serial chain of small launches: weight gradients run beside it (rtg.ops.wgrad_side)
    wgrad_side = True

    def __init__(self):
        super().__init__()
        self.num_kernels = len(hp.resblock_kernel_sizes)
        self.num_upsamples = len(hp.upsample_rates)
        self.n_layer = self.num_upsamples
        ch = 32
        uic = hp.upsample_initial_channel
        self.conv_pre_y = WNConv('conv', 1, ch, 7, pad=3)
        self.downs = nn.ModuleList([
            WNConv('conv', ch * 2 ** i, ch * 2 ** (i + 1), k, stride=u, pad=k // 2)
            for i, (u, k) in enumerate(zip(hp.upsample_rates[::-1], hp.upsample_kernel_sizes[::-1]))])
        self.resblock = nn.ModuleList([ResBlock(ch * 2 ** (i + 1), 5, [1, 3]) for i in range(len(self.downs))])
        self.conv_pre = WNConv('conv', hp.n_mel, uic, 7, pad=3)
        self.ups = nn.ModuleList([
            WNConv('convT', uic // (2 ** i) * 2, uic // (2 ** (i + 1)) * 2, k, stride=u, pad=k // 2, out_pad=u - 1)
            for i, (u, k) in enumerate(zip(hp.upsample_rates, hp.upsample_kernel_sizes))])
        self.resblocks = nn.ModuleList([ResBlock(uic // (2 ** i), k, d) for i in range(len(self.ups))
                                        for k, d in zip(hp.resblock_kernel_sizes, hp.resblock_dilation_sizes)])
        self.merge = nn.ModuleList([WNConv('conv', 256 + 128, 256, 7, pad=3), WNConv('conv', 128 + 64, 128, 7, pad=3),
                                    WNConv('conv', 64 + 32, 64, 7, pad=3)])
        self.conv_post = WNConv('conv', ch * 2, 1, 7, pad=3)
        self.noise = GaussianNoise()
        self._wn_removed = False

    def _extra_bank_params(self):
        return [('noise.w', self.noise.w)]

    def forward(self, x, y, noise_list=None):
        tok = self.token()
        nz = (lambda i: None) if noise_list is None else (lambda i: noise_list[i])
        o = []
        # every skip is lrelu(.) of the previous stage: fused as that stage's output activation (generator.py:619-626)
        y = conv(tok, self.conv_pre_y, y, act=ACT_LRELU, act_slope=LRELU_SLOPE)
        for i in range(self.n_layer):
            o.append(y)
            y = conv(tok, self.downs[i], y)
            # the last encoder output meets the mel before the decoder's first lrelu (generator.py:633-638)
            y = self.resblock[i].run(tok, y, final_act_slope=LRELU_SLOPE)
        x = conv(tok, self.conv_pre, x, act=ACT_LRELU, act_slope=LRELU_SLOPE)
        z = torch.cat([x, y], dim=1)                   # lrelu(cat) == cat(lrelu, lrelu)
        for i in range(self.n_layer):
            z = conv(tok, self.ups[i], z, pre_slope=LRELU_SLOPE if i > 0 else 1.0)
            z = conv(tok, self.merge[i], z, o[self.n_layer - i - 1])
            z = self.noise(z, nz(2 * i))
            nk = self.num_kernels
            z = _Mean3.apply(*run_branches(tok, [self.resblocks[i * nk + j] for j in range(nk)], z))
            z = self.noise(z, nz(2 * i + 1))
        return conv(tok, self.conv_post, z, pre_slope=LRELU_SLOPE, act=ACT_TANH)

    def remove_weight_norm(self):
        """generator.py:661-667; see Generator_RefineGAN_small.remove_weight_norm."""
        self._wn_removed = True
