"""Autograd wrappers over the C ABI of librtg.so (include/rtg.h).  Every function here launches hand-written HIP
kernels on the current torch stream; PyTorch only owns the memory and the autograd graph.  CPU tensors are refused."""
import ctypes as C
import os as _os

import torch

from . import lib as L
from . import tune
from . import bank as _bank
from .lib import lib, check, current_stream_ptr as _lib_stream_ptr


def _stream():
    return _lib_stream_ptr()


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise L.RtgError('RetuneGAN hot-path ops run on the MI355X HIP kernels only: got a CPU tensor '
                             '(move the model and its inputs to cuda; there is no CPU fallback)')


def _c(t):
    return t if t is None or t.is_contiguous() else t.contiguous()


# bench.py's live per-kernel timing: when PROFILE is a list, every conv launch is bracketed by events on the stream it
# is launched on and recorded as (kernel, variant, algorithmic flop, start event, end event)
PROFILE = None


def _timed(kernel, variant, flop, launch, label='', nbytes=0):
    if PROFILE is None:
        return launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = launch()
    e1.record()
    PROFILE.append((kernel, variant, flop, e0, e1, label, nbytes))
    return r


def timed_bw(name, nbytes, launch, label=''):
    """bench.py's live timing of a bandwidth kernel (PROFILE on): recorded as ('bw:<name>', 0, 0, e0, e1, label, bytes),
    bytes = the ALGORITHMIC HBM bytes of the launch (every operand once)."""
    return _timed('bw:' + name, 0, 0.0, launch, label, nbytes)


def _conv_bytes(d, args):
    """algorithmic HBM bytes of one rtg_conv1d launch: every operand tensor once (fp32), weights once"""
    two_d = d.h_k > 1 or d.h_n > 1
    x = (d.B // max(d.h_n, 1)) * (d.C1 // max(d.h_k, 1)) * d.h_in * d.L_in if two_d else d.B * (d.C1 + d.C2) * d.L_in
    out = d.B * d.out_C * d.out_L
    io = d.io_bf16
    aux, mask, res = args[2], args[5], args[6]
    n = x * (2 if io & 1 else 4) + out * (2 if io & 2 else 4) + 4 * (d.groups * d.Mg * d.Cg * d.K // (2 if d.bf16 else 1))
    n += 4 * x if aux else 0
    n += out * ((2 if io & 4 else 4) if mask else 0) + out * ((2 if io & 8 else 4) if res else 0)
    return n


# ---------------------------------------------------------------------------------------------------------------
# bf16 feature maps in HBM (hparam.compute_dtype = 'bf16' with hparam.bf16_maps; BASELINE configs[2])
# ---------------------------------------------------------------------------------------------------------------
# A bf16 ACTIVATION tensor of this package holds bf16(leaky_relu(x, ENC_SLOPE)) of the feature map x it stands for: the
# dense layers of the discriminators store their outputs that way (the consumer's activation applied once by the producer,
# RtgConv1dDesc.io_bf16), every consumer inside the stacks applies leaky_relu(., LRELU_SLOPE) to it anyway
# (retunegan/models/discrminator.py:90-98,212-218,300-304) and the sign — all a leaky-relu backward needs — survives; the
# feature-matching loss decodes (RTG_LOSS_L1_ENC).  A bf16 GRADIENT tensor is a plain rounding.  Kernels without a native
# bf16 path are served through rtg_bf16_decode / rtg_bf16_encode around their fp32 launch (correct for every shape; the
# hot shapes take the native path).
ENC_SLOPE = 0.15
_BF = torch.bfloat16


def _is_bf(t):
    return t is not None and t.dtype == _BF


_SLACK = 16        # readable bytes a bf16 tensor must have behind its last element (include/rtg.h, RtgConv1dDesc.io_bf16)


def empty_bf(shape, device):
    """an uninitialised bf16 tensor with _SLACK readable bytes behind it: the kernels read bf16 rows with 16-byte loads at
    2-byte granularity, and a load that starts at the last elements of the tensor runs past its end (the hardware's range
    check works on whole dwords: clipping the load at the tensor's end would drop the last element with it)"""
    n = 1
    for s in shape:
        n *= int(s)
    return torch.empty(n + _SLACK // 2, device=device, dtype=_BF)[:n].view(tuple(shape))


def _slacked(t):
    """`t` (bf16, contiguous) with the slack its kernels need: itself when its storage extends far enough behind it (every
    tensor this package allocates, and every clip-range slice of one), else a copy (tensors made by ATen, e.g. the sum
    autograd forms of two gradients)"""
    if t is None or t.dtype != _BF:
        return t
    t = _c(t)
    end = (t.storage_offset() + t.numel()) * 2
    if t.untyped_storage().nbytes() - end >= _SLACK:
        return t
    out = empty_bf(t.shape, t.device)
    out.copy_(t)
    return out


def bf16_encode(x, slope):
    """fp32 -> bf16(leaky_relu(x, slope)); slope 1: a plain rounding (gradients)"""
    x = _c(x)
    out = empty_bf(x.shape, x.device)
    check(timed_bw('bf16_cvt', 6 * x.numel(), lambda: lib.rtg_bf16_encode(_p(x), _p(out), x.numel(), float(slope), _stream())),
          'bf16_encode')
    return out


def bf16_decode(x, slope):
    """bf16(leaky_relu(x, slope)) -> fp32 x (slope 1: a plain widening)"""
    x = _c(x)
    out = torch.empty(x.shape, device=x.device, dtype=torch.float32)
    check(timed_bw('bf16_cvt', 6 * x.numel(), lambda: lib.rtg_bf16_decode(_p(x), _p(out), x.numel(), float(slope), _stream())),
          'bf16_decode')
    return out


class DecodeFn(torch.autograd.Function):
    """bf16 activation tensor -> the fp32 feature map it stands for (for layers / kernels that read fp32); the gradient goes
    back as a bf16 tensor"""

    @staticmethod
    def forward(ctx, x):
        _need_cuda(x)
        return bf16_decode(x, ENC_SLOPE)

    @staticmethod
    def backward(ctx, dy):
        return bf16_encode(dy, 1.0) if dy is not None else None


class PairDecodeFn(torch.autograd.Function):
    """DecodeFn for the two halves (constant, differentiable) of one 2B-clip buffer: -> the halves of ONE fp32 buffer"""

    @staticmethod
    def forward(ctx, x_c, x_g):
        _need_cuda(x_c, x_g)
        assert _adjacent(x_c, x_g)
        B = x_c.shape[0]
        whole = torch.as_strided(x_c, (2 * B,) + tuple(x_c.shape[1:]), x_c.stride())
        buf = bf16_decode(whole, ENC_SLOPE)
        o_c, o_g = buf[:B], buf[B:]
        ctx.mark_non_differentiable(o_c)
        ctx.set_materialize_grads(False)
        return o_c, o_g

    @staticmethod
    def backward(ctx, d_c, d_g):
        return None, (bf16_encode(d_g, 1.0) if d_g is not None else None)


def decode(x):
    return DecodeFn.apply(x) if _is_bf(x) else x


_NATIVE = {}


def _conv_native(d):
    """does a kernel read / write the bf16 tensors of descriptor `d` (io_bf16 set) natively at this shape?"""
    d.tile_cfg = 0
    key = bytes(d)
    ok = _NATIVE.get(key)
    if ok is None:
        cands = (C.c_int * 8)()
        ok = _NATIVE[key] = lib.rtg_conv1d_tile_candidates(C.byref(d), cands, 8) > 0
    return ok


def _wgrad_native(wd):
    wd.shape_cfg, wd.splits, wd.part_stride = 0, 1, 0
    key = b'w' + bytes(wd)
    ok = _NATIVE.get(key)
    if ok is None:
        cands = (C.c_int * 12)()
        ok = _NATIVE[key] = lib.rtg_wgrad_shape_candidates(C.byref(wd), cands, 12) > 0
    return ok


def _run_conv_t(d, x1, wp, bias, mask, res, out, flop, label, what, x_slope=1.0, out_slope=1.0):
    """rtg_conv1d on tensors of either type (single-input layers: no x2 / aux / out2).  x1 bf16: activations encoded with
    x_slope (= the layer's pre_slope) or, x_slope 1, a gradient; out bf16: stored as bf16(leaky_relu(., out_slope)).  Native
    where the dense-layer kernel serves the shape, else through fp32 copies."""
    io = (L.IO_X_BF16 if _is_bf(x1) else 0) | (L.IO_OUT_BF16 if _is_bf(out) else 0) | \
         (L.IO_MASK_BF16 if _is_bf(mask) else 0) | (L.IO_RES_BF16 if _is_bf(res) else 0)
    if io:
        x1 = _slacked(x1)
        if _is_bf(x1) and abs(x_slope - 1.0) > 1e-6 and (d.pre_mode != L.PRE_LRELU or abs(d.pre_slope - x_slope) > 1e-6):
            # (an encoded tensor holds leaky_relu(x, x_slope): only a layer that applies exactly that activation may read
            # it as if it were x; a layer without pre-activation would silently convolve the activated values)
            raise L.RtgError(f'{label}: a bf16 activation tensor encoded with slope {x_slope} feeds a layer with pre-activation '
                             f'{"none" if d.pre_mode != L.PRE_LRELU else d.pre_slope}')
        d.io_bf16, d.enc_slope = io, float(out_slope)
        if not _conv_native(d):
            # no native kernel at this shape: fp32 copies around the fp32 launch
            d.io_bf16, d.enc_slope = 0, 1.0
            x32 = bf16_decode(x1, x_slope) if _is_bf(x1) else x1
            m32 = bf16_decode(mask, 1.0) if _is_bf(mask) else mask
            r32 = bf16_decode(res, 1.0) if _is_bf(res) else res
            o32 = torch.empty(out.shape, device=out.device, dtype=torch.float32) if _is_bf(out) else out
            _run_conv(d, (_p(x32), None, None, wp, bias, _p(m32), _p(r32), _p(o32), None, _stream()), flop, label, what)
            if _is_bf(out):
                check(lib.rtg_bf16_encode(_p(o32), _p(out), out.numel(), float(out_slope), _stream()), 'bf16_encode')
            return
    _run_conv(d, (_p(x1), None, None, wp, bias, _p(mask), _p(res), _p(out), None, _stream()), flop, label, what)


def _run_conv(d, args, flop, label, what):
    """rtg_conv1d with the tuned block shape (rtg/tune.py); args = everything after the descriptor"""
    d.tile_cfg = tune.conv_cfg(d, lambda: lib.rtg_conv1d(C.byref(d), *args))
    if d.tile_cfg < 8000:
        # a general block shape reads the STANDARD image: report it, keyed on the image's address (a dictionary miss for
        # layers without a fragment image) rather than on d.wp16 — a caller that forgot wp16 in its descriptor must not
        # make the lean pack drop an image that is still read (round 4: the unfused ResidualStack forward)
        _bank.note_std_use(args[3].value)
    check(_timed('conv1d', lib.rtg_conv1d_variant(C.byref(d)) if PROFILE is not None else 0, flop,
                 lambda: lib.rtg_conv1d(C.byref(d), *args), label, _conv_bytes(d, args) if PROFILE is not None else 0), what)


def _wgrad_io(wd, ly, a1, gyt, x_encoded=True):
    """bf16 operands of a weight gradient (x = a bf16 activation tensor, ENCODED with ENC_SLOPE unless `x_encoded` is False,
    and / or dy = a plain bf16 gradient): native where the dense-layer kernel serves the shape (RtgWgradDesc.io_bf16), else
    fp32 copies.  -> (x, dy) to launch on"""
    io = (L.IO_X_BF16 if _is_bf(a1) else 0) | (L.IO_OUT_BF16 if _is_bf(gyt) else 0)
    wd.io_bf16 = 0
    if io:
        if _is_bf(a1) and x_encoded and (wd.pre_mode != L.PRE_LRELU or abs(wd.pre_slope - ENC_SLOPE) > 1e-6):
            raise L.RtgError(f'wgrad {ly.name}: a bf16 activation tensor encoded with slope {ENC_SLOPE} feeds a layer with '
                             f'pre-activation {"none" if wd.pre_mode != L.PRE_LRELU else wd.pre_slope}')
        a1, gyt = _slacked(a1), _slacked(gyt)
        wd.bf16, wd.io_bf16 = int(getattr(ly, 'wgrad_bf', 0)), io
        if not (wd.bf16 and _wgrad_native(wd)):
            wd.io_bf16 = 0
            if _is_bf(a1):
                a1 = bf16_decode(a1, ENC_SLOPE if x_encoded else 1.0)
            if _is_bf(gyt):
                gyt = bf16_decode(gyt, 1.0)
    return a1, gyt


def _run_wgrad(wd, ptrs, st, bank, ly, tok_id, flop, label, what):
    """rtg_conv1d_wgrad into the bank's partial slot with the tuned block shape; ptrs = (x1, x2, dy, gy_aux)"""
    wd.bf16 = int(getattr(ly, 'wgrad_bf', 0))
    run = lambda part: lib.rtg_conv1d_wgrad(C.byref(wd), *ptrs, _p(part), st)  # noqa: E731
    if wd.bf16 and BF16_FP32_THIN and ly.kind == 'conv' and ly.groups > 1 and ly.k == 41:
        # (a thin-group layer under bf16 operands: its fp32 shapes — exact-fit matrix-core tiles, vector ALUs — compete)
        wd.bf16, wd.shape_cfg = tune.wgrad_cfg_any(wd, run)
    else:
        wd.shape_cfg = tune.wgrad_cfg(wd, run)
    splits = lib.rtg_wgrad_splits(C.byref(wd))
    if splits < 1:
        raise L.RtgError(f'wgrad geometry refused for {ly.name}: {splits}')
    part, stride, immediate = bank.partial_slot(ly, splits, tok_id)
    bank.note_backward_stream()
    wd.splits, wd.part_stride = splits, stride
    check(_timed('wgrad', 32 if wd.Mg >= 32 else 16, flop,
                 lambda: lib.rtg_conv1d_wgrad(C.byref(wd), *ptrs, _p(part), st), f'{label} splits{splits} s{wd.shape_cfg}'), what)
    return part, splits, immediate


# Weight gradients off the critical path.  The backward of a chain of layers is the chain of its backward-data launches;
# a layer's weight gradient depends on that chain but nothing of the chain depends on it.  For a bank that asks for it
# (`bank.wgrad_side`: the generator, whose layers are small launches that leave most of the chip idle, and whose backward
# is one serial chain) the weight-gradient launches go to a side stream: they overlap the following backward-data
# launches instead of delaying them.  The bank's flush waits for every stream that ran weight gradients
# (note_backward_stream).  Not under the tuner or the per-launch profile (both time launches in isolation).
# (Rounds 2-4 A/B, DESIGN.md 3: neutral eager, +1.2 % under graph replay: settled on.)
WGRAD_SIDE = True
_SIDE_STREAMS = {}


class wgrad_side:
    """context: the current stream becomes the side stream paired with it (ordered after everything queued so far);
    `tensors` are the operands the launches inside read: the caching allocator must not recycle them before the side
    stream is done with them"""

    def __init__(self, bank, tensors):
        self.on = WGRAD_SIDE and bank.wgrad_side and not tune.ACTIVE and PROFILE is None
        self.tensors = tensors

    def __enter__(self):
        if not self.on:
            return self
        main = torch.cuda.current_stream()
        side = _SIDE_STREAMS.get(main)
        if side is None:
            side = _SIDE_STREAMS[main] = L.new_stream(device=main.device)
        side.wait_stream(main)
        for t in self.tensors:
            if t is not None:
                t.record_stream(side)
        self.ctx = torch.cuda.stream(side)
        self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.on:
            self.ctx.__exit__(*exc)
        return False


# (False: every weight gradient as its own launch; settled on in round 2: 0.5 ms per step)
WGRAD_GROUP = True


def _run_wgrad_group(items, st, bank, tok_id, label):
    """The weight gradients of several layers of one bank, as ONE launch where the tuner found that faster
    (rtg_conv1d_wgrad_group: small problems that each leave most of the chip idle), else one by one.
    items: [(wd, (x1, x2, dy, gy_aux), ly, flop, label, what)]."""
    n = len(items)
    wds = [it[0] for it in items]
    code = 0
    if WGRAD_GROUP and 2 <= n <= L.WGRAD_MAX_GROUP and all(w.Mg >= 32 and not getattr(it[2], 'wgrad_bf', 0)
                                                            for w, it in zip(wds, items)):
        for w in wds:
            w.bf16, w.shape_cfg, w.splits, w.part_stride = 0, 0, 1, 0
        need = [w.groups * w.Mg * (w.Cg * w.K + 1) for w in wds]

        def set_cfg(c, parts):
            """descriptor / pointer arrays of the group at code c into the partial buffers parts(i, splits) -> (ptr, stride)"""
            darr, parr, spl = (L.WgradDesc * n)(), (L.WgradPtrs * n)(), []
            for i, (w, it) in enumerate(zip(wds, items)):
                w.shape_cfg, w.splits, w.part_stride = c & 15, 1, 0
                s = lib.rtg_wgrad_splits(C.byref(w))
                if s < 1:
                    return None
                if c & 16:
                    s = max(1, -(-s // n))
                ptr, stride = parts(i, s)
                w.splits, w.part_stride = s, stride
                darr[i] = w
                x1, x2, dy, aux = it[1]
                parr[i] = L.WgradPtrs(x1.value if x1 else None, x2.value if x2 else None, dy.value if dy else None,
                                      aux.value if aux else None, ptr)
                spl.append(s)
            return darr, parr, spl

        def reset():
            for w in wds:
                w.shape_cfg, w.splits, w.part_stride = 0, 1, 0

        def run_group(c):
            scratch = []

            def parts(i, s):
                scratch.append(torch.empty(s * need[i], device='cuda'))
                return scratch[-1].data_ptr(), need[i]
            r = set_cfg(c, parts)
            if r is None:
                return -1
            return lib.rtg_conv1d_wgrad_group(n, r[0], r[1], st)

        def run_singles():
            for i, (w, it) in enumerate(zip(wds, items)):
                cfg = tune.wgrad_cfg(w, lambda part, w=w, it=it: lib.rtg_conv1d_wgrad(C.byref(w), *it[1], _p(part), st))
                w.shape_cfg, w.splits, w.part_stride = cfg, 1, 0
                s = lib.rtg_wgrad_splits(C.byref(w))
                if s < 1:
                    return -1
                part = torch.empty(s * need[i], device='cuda')
                w.splits, w.part_stride = s, need[i]
                rc = lib.rtg_conv1d_wgrad(C.byref(w), *it[1], _p(part), st)
                if rc:
                    return rc
            return 0
        reset()
        code = tune.wgrad_group_cfg(wds, run_group, lambda: (run_singles(), reset())[0])
        reset()
    if code == 0:
        out = []
        for wd, ptrs, ly, flop, lab, what in items:
            wd.shape_cfg, wd.splits, wd.part_stride = 0, 1, 0
            out.append(_run_wgrad(wd, ptrs, st, bank, ly, tok_id, flop, lab, what))
        return out
    slots = []

    def parts(i, s):
        part, stride, immediate = bank.partial_slot(items[i][2], s, tok_id)
        slots.append((part, s, immediate))
        return part.data_ptr(), stride
    darr, parr, spl = set_cfg(code, parts)
    bank.note_backward_stream()
    check(_timed('wgrad', 32, sum(it[3] for it in items),
                 lambda: lib.rtg_conv1d_wgrad_group(n, darr, parr, st), f'wgrad group {label} x{n} s{code}'),
          f'conv1d wgrad group {label}')
    return slots


def _conv_flop(ly, B, L_conv_out):
    """algorithmic flop of one pass (forward, backward-data or backward-weight) of the layer's convolution"""
    return 2.0 * B * L_conv_out * ly.cout * (ly.cin // ly.groups) * ly.k


# ---------------------------------------------------------------------------------------------------------------
# convolution (Conv1d / ConvTranspose1d, weight-normed, banked)
# ---------------------------------------------------------------------------------------------------------------
def _conv_out_len(ly, L_in):
    if ly.kind == 'conv':
        return (L_in + 2 * ly.pad - ly.dil * (ly.k - 1) - 1) // ly.stride + 1
    return (L_in - 1) * ly.stride - 2 * ly.pad + ly.k + ly.out_pad


def _desc(**kw):
    base = dict(B=1, C1=1, C2=0, L_in=1, groups=1, Cg=1, Mg=1, K=1, stride=1, dil=1, pad=0, Q=1, out_C=1, out_L=1,
                shuf_S=1, shuf_P=0, pre_mode=0, pre_slope=1.0, mask_slope=1.0, out_scale=1.0, act=0, act_slope=1.0,
                accumulate=0, tile_m=32, out_split=0)
    base.update(kw)
    return L.Conv1dDesc(**base)


# the thin-group MSD layers: vector-ALU kernels (rtg_gconv.hip) and the exact-fit matrix-core forward (rtg_gmfma.hip) as
# alternatives the tuner times against the general kernel (False: the general kernel only; tests flip GCONV)
GCONV = True
GMFMA = True
# bf16 operands (hparam.compute_dtype, BASELINE configs[2]): the thin-group layers' fp32 kernels (vector ALUs, exact-fit
# matrix-core tiles) stay candidates next to the bf16 general kernel, whose tiles are mostly padding for 8 channels per
# group — fp32 arithmetic where it is FASTER than bf16 is no loss of precision (round 4: config 3 51.3 -> 49.3 ms/step)
BF16_FP32_THIN = True


def _gconv_forward(ly, bank, tok_id, d, args, x_ptr, B, L_in, out, pre_slope, flop, label):
    """The thin-group k41 layers of MSD (4-8 input, 8-16 output channels per group): matrix cores (tiles that are mostly
    padding) or the vector ALUs (rtg_gconv.hip), whichever the tuner measured faster for this problem.  d / args: the
    rtg_conv1d launch of the same plain forward (out = conv(lrelu(x)) + bias) on B clips at x_ptr.  True: ran on the
    vector ALUs."""
    if not (GCONV and ly.kind == 'conv' and ly.groups > 1 and ly.k == 41 and (not ly.fwd_bf or BF16_FP32_THIN)):
        return False
    gd = L.GconvDesc(B, ly.groups, ly.cin // ly.groups, ly.cout // ly.groups, ly.k, ly.stride, ly.pad, L_in, out.shape[-1],
                     pre_slope)
    if lib.rtg_gconv_ok(C.byref(gd)) != 1:
        return False
    d.tile_cfg = tune.conv_cfg(d, lambda: lib.rtg_conv1d(C.byref(d), *args))
    gargs = (x_ptr, _p(bank.gconv_weights(ly, gd, tok_id)), bank.bias_ptr(ly), _p(out), _stream())
    ways = [lambda: lib.rtg_conv1d(C.byref(d), *args), lambda: lib.rtg_gconv_forward(C.byref(gd), *gargs)]
    # round 4: the exact-fit matrix-core kernel (rtg_gmfma.hip) as a third way, where it serves the shape
    wm = bank.gmfma_weights(ly, gd) if GMFMA and lib.rtg_gmfma_ok(C.byref(gd)) == 1 else None
    if wm is not None:
        margs = (x_ptr, _p(wm), bank.bias_ptr(ly), _p(out), _stream())
        ways.append(lambda: lib.rtg_gmfma_forward(C.byref(gd), *margs))
    which = tune.alt_choice((b'gconv3' if wm is not None else b'gconv') + bytes(gd), ways)
    if which == 0:
        return False
    check(_timed('conv1d', 7200 if which == 1 else 7202, flop, ways[which], label,
                 _conv_bytes(d, args) if PROFILE is not None else 0), f'gconv {label}')
    return True


def _gconv_dgrad(ly, bank, tok_id, d, args, dy_ptr, mask_ptr, res_ptr, B, L_in, L_out, dx, pre_slope, flop, label):
    """backward-data of the same layers: dx = res + lrelu'(x) * conv_transpose(dy); d / args: the rtg_conv1d launch of it"""
    if not (GCONV and ly.kind == 'conv' and ly.groups > 1 and ly.k == 41 and (not ly.bwd_bf or BF16_FP32_THIN)):
        return False
    gd = L.GconvDesc(B, ly.groups, ly.cin // ly.groups, ly.cout // ly.groups, ly.k, ly.stride, ly.pad, L_in, L_out, pre_slope)
    if lib.rtg_gconv_ok(C.byref(gd)) != 1:
        return False
    d.tile_cfg = tune.conv_cfg(d, lambda: lib.rtg_conv1d(C.byref(d), *args))
    gargs = (dy_ptr, _p(bank.gconv_weights(ly, gd, tok_id, bwd=True)), mask_ptr if pre_slope != 1.0 else None, res_ptr, _p(dx),
             _stream())
    if tune.alt_choice(b'gconv_bwd' + bytes(gd), [lambda: lib.rtg_conv1d(C.byref(d), *args),
                                                 lambda: lib.rtg_gconv_backward_data(C.byref(gd), *gargs)]) != 1:
        return False
    check(_timed('conv1d', 7201, flop, lambda: lib.rtg_gconv_backward_data(C.byref(gd), *gargs), label,
                 _conv_bytes(d, args) if PROFILE is not None else 0), f'gconv {label}')
    return True


class ConvFn(torch.autograd.Function):
    """out = act(out_scale * (conv(pre(xcat)) + bias + res)), conv being the layer's Conv1d or ConvTranspose1d."""

    @staticmethod
    def forward(ctx, token, x1, x2, res, ly, pre_slope, act, act_slope, out_scale, res_is_input):
        _need_cuda(x1, x2, res)
        bank = token._rtg_bank
        x1, x2, res = _c(x1), _c(x2), _c(res)
        B, C1, L_in = x1.shape
        C2 = x2.shape[1] if x2 is not None else 0
        assert C1 + C2 == ly.cin, (ly.name, C1, C2, ly.cin)
        L_out = _conv_out_len(ly, L_in)
        plain = x2 is None and res is None and act == L.ACT_NONE and out_scale == 1.0
        # bf16 feature maps: a dense discriminator layer stores bf16(leaky_relu(out, ENC_SLOPE)) (what its consumer stages)
        out_bf = bool(ly.maps_bf and plain and ly.kind == 'conv')
        if _is_bf(x1) and not (plain and ly.kind == 'conv'):
            raise L.RtgError(f'{ly.name}: a bf16 feature map feeds a layer form without a bf16 path (ops.conv decodes first)')
        out = empty_bf((B, ly.cout, L_out), x1.device) if out_bf else torch.empty(B, ly.cout, L_out, device=x1.device)
        mode, g, mg, cg, k, s = ly.fwd_op
        pre_mode = L.PRE_LRELU if pre_slope != 1.0 else L.PRE_NONE
        if ly.kind == 'conv':
            d = _desc(B=B, C1=C1, C2=C2, L_in=L_in, groups=g, Cg=cg, Mg=mg, K=k, stride=ly.stride, dil=ly.dil,
                      pad=ly.pad, Q=L_out, out_C=ly.cout, out_L=L_out, pre_mode=pre_mode, pre_slope=pre_slope,
                      out_scale=out_scale, act=act, act_slope=act_slope, tile_m=ly.fwd_tm, tap_major=ly.fwd_tap,
                      bf16=ly.fwd_bf, wp16=ly.fwd16)
        else:
            nq = (L_out - 1 + ly.pad) // ly.stride + 1
            d = _desc(B=B, C1=C1, C2=C2, L_in=L_in, groups=1, Cg=cg, Mg=mg, K=k, stride=1, dil=1, pad=k - 1, Q=nq,
                      out_C=ly.cout, out_L=L_out, shuf_S=ly.stride, shuf_P=ly.pad, pre_mode=pre_mode,
                      pre_slope=pre_slope, out_scale=out_scale, act=act, act_slope=act_slope, tile_m=ly.fwd_tm, tap_major=ly.fwd_tap,
                      bf16=ly.fwd_bf, wp16=ly.fwd16)
        lc = L_out if ly.kind == 'conv' else L_in
        if out_bf or _is_bf(x1):
            _run_conv_t(d, x1, bank.fwd_ptr(ly), bank.bias_ptr(ly), None, None, out, _conv_flop(ly, B, lc),
                        f'fwd {ly.name} B{B} L{L_in}', f'conv1d fwd {ly.name}', x_slope=ENC_SLOPE, out_slope=ENC_SLOPE)
        else:
            args = (_p(x1), _p(x2), None, bank.fwd_ptr(ly), bank.bias_ptr(ly), None, _p(res), _p(out), None, _stream())
            if not (plain and _gconv_forward(ly, bank, token._rtg_id, d, args, _p(x1), B, L_in, out, pre_slope, _conv_flop(ly, B, lc),
                                             f'fwd {ly.name} B{B} L{L_in}')):
                _run_conv(d, args, _conv_flop(ly, B, lc), f'fwd {ly.name} B{B} L{L_in}', f'conv1d fwd {ly.name}')
        ctx.ly, ctx.bank, ctx.tok_id = ly, bank, token._rtg_id
        ctx.cfg = (pre_slope, act, act_slope, out_scale, res_is_input, res is not None)
        ctx.save_for_backward(x1, x2, out if act != L.ACT_NONE else None)
        ctx.set_materialize_grads(False)
        return out

    @staticmethod
    def backward(ctx, dy):
        ly, bank = ctx.ly, ctx.bank
        x1, x2, out = ctx.saved_tensors
        pre_slope, act, act_slope, out_scale, res_is_input, has_res = ctx.cfg
        need_w, need_x1, need_x2, need_res = (ctx.needs_input_grad[0], ctx.needs_input_grad[1],
                                              ctx.needs_input_grad[2], ctx.needs_input_grad[3])
        if dy is None:
            return (None,) * 10
        dy = _c(dy)
        B, C1, L_in = x1.shape
        C2 = x2.shape[1] if x2 is not None else 0
        L_out = dy.shape[-1]
        st = _stream()
        if act == L.ACT_LRELU:
            gy_mode, gy_slope = L.PRE_MUL_DLRELU, act_slope
        elif act == L.ACT_TANH:
            gy_mode, gy_slope = L.PRE_MUL_DTANH, 1.0
        else:
            gy_mode, gy_slope = L.PRE_NONE, 1.0

        # ---------------- gradient of the residual input (= effective output gradient)
        dres = None
        fuse_res = has_res and res_is_input and act == L.ACT_NONE and need_x1
        if has_res and need_res and not fuse_res:
            if act == L.ACT_NONE and out_scale == 1.0:
                dres = dy
            else:
                dres = torch.empty_like(dy)
                if act == L.ACT_LRELU:
                    check(lib.rtg_lrelu_bwd(_p(dy), _p(out), _p(dres), dy.numel(), act_slope, st), 'lrelu_bwd')
                    if out_scale != 1.0:
                        check(lib.rtg_axpby(_p(dres), None, _p(dres), dres.numel(), out_scale, 0.0, 0, st), 'axpby')
                elif act == L.ACT_NONE:
                    check(lib.rtg_axpby(_p(dy), None, _p(dres), dy.numel(), out_scale, 0.0, 0, st), 'axpby')
                else:
                    raise L.RtgError('residual + tanh epilogue has no backward kernel')

        # ---------------- backward-data
        dx1 = dx2 = None
        if need_x1 or need_x2:
            mode, g, mg, cg, k, s = ly.bwd_op
            split = C1 if C2 > 0 else 0
            if C2 > 0:
                assert pre_slope == 1.0, 'input activation on a concatenated pair is not used by the path'
            if need_x1 or C2 == 0:
                dx1 = empty_bf(x1.shape, x1.device) if _is_bf(x1) else torch.empty_like(x1)
            if C2 > 0 and need_x2:
                dx2 = torch.empty_like(x2)
            mask = x1 if pre_slope != 1.0 else None
            resg = dy if fuse_res else None
            # (acc * mask + res) * out_scale: the residual branch and the conv branch share the factor out_scale
            if ly.kind == 'conv' and ly.stride == 1:
                d = _desc(B=B, C1=ly.cout, L_in=L_out, groups=g, Cg=cg, Mg=mg, K=k, stride=1, dil=ly.dil,
                          pad=(ly.k - 1) * ly.dil - ly.pad, Q=L_in, out_C=ly.cin, out_L=L_in, pre_mode=gy_mode,
                          pre_slope=gy_slope, mask_slope=pre_slope, out_scale=out_scale, tile_m=ly.bwd_tm,
                          out_split=split, tap_major=ly.bwd_tap, bf16=ly.bwd_bf, wp16=ly.bwd16)
            elif ly.kind == 'conv':
                nq = (L_in - 1 + ly.pad) // ly.stride + 1
                d = _desc(B=B, C1=ly.cout, L_in=L_out, groups=g, Cg=cg, Mg=mg, K=k, stride=1, dil=1, pad=k - 1, Q=nq,
                          out_C=ly.cin, out_L=L_in, shuf_S=ly.stride, shuf_P=ly.pad, pre_mode=gy_mode,
                          pre_slope=gy_slope, mask_slope=pre_slope, out_scale=out_scale, tile_m=ly.bwd_tm,
                          out_split=split, tap_major=ly.bwd_tap, bf16=ly.bwd_bf, wp16=ly.bwd16)
            else:   # transposed conv: backward-data is the strided conv of dy
                d = _desc(B=B, C1=ly.cout, L_in=L_out, groups=1, Cg=cg, Mg=mg, K=k, stride=ly.stride, dil=1,
                          pad=ly.pad, Q=L_in, out_C=ly.cin, out_L=L_in, pre_mode=gy_mode, pre_slope=gy_slope,
                          mask_slope=pre_slope, out_scale=out_scale, tile_m=ly.bwd_tm, out_split=split,
                          tap_major=ly.bwd_tap, bf16=ly.bwd_bf, wp16=ly.bwd16)
            lc = L_out if ly.kind == 'conv' else L_in
            plain = C2 == 0 and gy_mode == L.PRE_NONE and out_scale == 1.0 and dx2 is None
            if _is_bf(dy) or _is_bf(x1):
                # bf16 feature maps: dy is a bf16 gradient, the mask a bf16 (encoded: same sign) or fp32 feature map, dx takes
                # the type of the tensor it is the gradient of
                assert plain and ly.kind == 'conv' and out is None
                _run_conv_t(d, dy, bank.bwd_ptr(ly), None, mask, resg, dx1, _conv_flop(ly, B, lc),
                            f'dgrad {ly.name} B{B} L{L_in}', f'conv1d bwd-data {ly.name}')
            else:
                dargs = (_p(dy), None, _p(out), bank.bwd_ptr(ly), None, _p(mask), _p(resg), _p(dx1), _p(dx2), st)
                if not (plain and _gconv_dgrad(ly, bank, ctx.tok_id, d, dargs, _p(dy), _p(x1), _p(resg), B, L_in, L_out, dx1, pre_slope,
                                               _conv_flop(ly, B, lc), f'dgrad {ly.name} B{B} L{L_in}')):
                    _run_conv(d, dargs, _conv_flop(ly, B, lc), f'dgrad {ly.name} B{B} L{L_in}', f'conv1d bwd-data {ly.name}')
            if not need_x1:
                dx1 = None

        # ---------------- backward-weight into the bank's partial slot (flushed by the token's backward)
        if need_w:
            pre_mode = L.PRE_LRELU if pre_slope != 1.0 else L.PRE_NONE
            if ly.kind == 'conv':
                wd = L.WgradDesc(B=B, C1=C1, C2=C2, L_in=L_in, groups=ly.groups, Cg=ly.cin // ly.groups,
                                 Mg=ly.cout // ly.groups, K=ly.k, stride=ly.stride, dil=ly.dil, pad=ly.pad, Q=L_out,
                                 dy_L=L_out, pre_mode=pre_mode, pre_slope=pre_slope, gy_mode=gy_mode,
                                 gy_slope=gy_slope, gy_scale=out_scale, splits=1, part_stride=0)
                a1, a2, gyt, aux = x1, x2, dy, out
            else:
                # roles swapped: "input" = effective dy [B,C_out,L_out], "output gradient" = lrelu(x) [B,C_in,L_in]
                assert act == L.ACT_NONE and C2 == 0
                wd = L.WgradDesc(B=B, C1=ly.cout, C2=0, L_in=L_out, groups=1, Cg=ly.cout, Mg=ly.cin, K=ly.k,
                                 stride=ly.stride, dil=1, pad=ly.pad, Q=L_in, dy_L=L_in, pre_mode=L.PRE_NONE,
                                 pre_slope=1.0, gy_mode=pre_mode, gy_slope=pre_slope, gy_scale=out_scale, splits=1,
                                 part_stride=0)
                a1, a2, gyt, aux = dy, None, x1, None
            lc = L_out if ly.kind == 'conv' else L_in
            with wgrad_side(bank, (a1, a2, gyt, aux, dy)):
                st = _stream()
                a1, gyt = _wgrad_io(wd, ly, a1, gyt, x_encoded=(ly.kind == 'conv'))
                part, splits, immediate = _run_wgrad(wd, (_p(a1), _p(a2), _p(gyt), _p(aux)), st, bank, ly, ctx.tok_id,
                                                     _conv_flop(ly, B, lc), f'wgrad {ly.name} B{B} L{L_in}',
                                                     f'conv1d wgrad {ly.name}')
                if ly.kind == 'convT':   # bias gradient of a transposed conv: plain channel sum of dy
                    ws = torch.empty(32 * ly.cout, device=dy.device)
                    check(lib.rtg_channel_sum(_p(dy), C.c_void_p(bank.gflat.data_ptr() + 4 * ly.b_off), B, ly.cout, L_out,
                                              _p(ws), st), 'channel_sum')
                if immediate:
                    bank.flush_one(ly, part, splits)
        return None, dx1, dx2, dres, None, None, None, None, None, None


# ---------------------------------------------------------------------------------------------------------------
# ResidualStack in one launch per direction (rtg_resstack.hip)
# ---------------------------------------------------------------------------------------------------------------
RESSTACK = True                                                  # (False: six conv launches per direction)
# bit mask of the fused-stack instances used (rtg_resstack_ok's instance numbers): only (128 channels, 32 samples) beat six
# launches of the general kernel inside the train step; the 64- and 32-channel instances are built and tested, not served
RESSTACK_KINDS = 1


def _stack_desc(lys, B, Lx, pre_slope, final_act_slope, reverse=False):
    dils = [ly.dil for ly in (reversed(lys) if reverse else lys)]
    return L.ResStackDesc(B, lys[0].cin, Lx, (C.c_int * 6)(*dils), pre_slope, int(final_act_slope is not None),
                          float(final_act_slope or 1.0))


def resstack_shape_ok(lys, x):
    """the six convs of a ResidualStack (execution order) on input x: what ResStackFn runs as one autograd node"""
    if not (RESSTACK and len(lys) == 6 and x.is_cuda and x.dim() == 3):
        return False
    c = lys[0].cin
    return x.shape[1] == c and all(ly.kind == 'conv' and ly.cin == c and ly.cout == c and ly.k == 3 and ly.stride == 1 and
                                   ly.groups == 1 and ly.pad == ly.dil for ly in lys)


def resstack_ok(lys, x):
    """... and can run as ONE fused launch per direction (rtg_resstack.hip)"""
    if not resstack_shape_ok(lys, x):
        return False
    if any(ly.fwd_bf or ly.bwd_bf or ly.fwd_tap or ly.bwd_tap or ly.fwd_tm != 32 or ly.bwd_tm != 32 for ly in lys):
        return False
    B, c, Lx = x.shape
    d = L.ResStackDesc(B, c, Lx, (C.c_int * 6)(*[ly.dil for ly in lys]), 0.01, 0, 1.0)
    kind = lib.rtg_resstack_ok(C.byref(d))
    return kind >= 1 and bool(RESSTACK_KINDS >> (kind - 1) & 1)


class ResStackFn(torch.autograd.Function):
    """y = ResidualStack(x) (retunegan/models/generator.py:33-77), optionally followed by leaky_relu(final_act_slope), as
    one autograd node.  fused: forward and backward-data of the six convs in ONE launch each (rtg_resstack.hip, clips
    that fit in LDS); else six conv launches per direction, the residual gradient riding the backward-data epilogue
    (no accumulation kernels).  Either way the six weight gradients run on the tensors left in HBM, as one grouped
    launch where that is faster."""

    @staticmethod
    def forward(ctx, token, x, lys, pre_slope, final_act_slope, fused):
        _need_cuda(x)
        bank = token._rtg_bank
        x = _c(x)
        B, Cc, Lx = x.shape
        outs = [torch.empty_like(x) for _ in range(6)]            # r1, x1, r2, x2, r3, y
        st = _stream()
        if fused:
            d = _stack_desc(lys, B, Lx, pre_slope, final_act_slope)
            wp = L.PtrArray6(*[bank.fwd_ptr(ly).value for ly in lys])
            for ly in lys:                                        # (the fused kernel reads the STANDARD images: a layer that
                if ly.fwd16:                                      # also carries the fragment image must keep it in the pack)
                    _bank.note_std_use(bank.fwd_ptr(ly).value)
            bias = L.PtrArray6(*[bank.bias_ptr(ly).value for ly in lys])
            op = L.PtrArray6(*[o.data_ptr() for o in outs])
            flop = sum(_conv_flop(ly, B, Lx) for ly in lys)
            check(_timed('conv1d', 7100, flop, lambda: lib.rtg_resstack_forward(C.byref(d), _p(x), C.byref(wp),
                                                                               C.byref(bias), C.byref(op), st),
                         f'fwd {lys[0].name}..stack B{B} L{Lx}', 4 * 7 * x.numel()), 'resstack fwd')
        else:
            ins = (x, *outs[:5])
            for i, ly in enumerate(lys):
                last = i == 5 and final_act_slope is not None
                mode, g, mg, cg, k, s = ly.fwd_op
                d = _desc(B=B, C1=Cc, C2=0, L_in=Lx, groups=g, Cg=cg, Mg=mg, K=k, stride=1, dil=ly.dil, pad=ly.pad, Q=Lx,
                          out_C=Cc, out_L=Lx, pre_mode=L.PRE_LRELU, pre_slope=pre_slope,
                          act=L.ACT_LRELU if last else L.ACT_NONE, act_slope=final_act_slope if last else 1.0,
                          tile_m=ly.fwd_tm, tap_major=ly.fwd_tap, bf16=ly.fwd_bf, wp16=ly.fwd16)
                res = ins[i - 1] if i % 2 else None           # x_{j+1} = x_j + conv3(lrelu(r_{j+1}))
                _run_conv(d, (_p(ins[i]), None, None, bank.fwd_ptr(ly), bank.bias_ptr(ly), None, _p(res), _p(outs[i]), None,
                              st), _conv_flop(ly, B, Lx), f'fwd {ly.name} B{B} L{Lx}', f'conv1d fwd {ly.name}')
        ctx.lys, ctx.bank, ctx.tok_id = lys, bank, token._rtg_id
        ctx.cfg = (pre_slope, final_act_slope, fused)
        ctx.save_for_backward(x, *outs)
        ctx.set_materialize_grads(False)
        return outs[5]

    @staticmethod
    def backward(ctx, dy):
        if dy is None:
            return None, None, None, None, None, None
        lys, bank = ctx.lys, ctx.bank
        pre_slope, final_act_slope, fused = ctx.cfg
        x0, r1, x1, r2, x2, r3, y = ctx.saved_tensors
        dy = _c(dy)
        B, Cc, Lx = x0.shape
        gouts = [torch.empty_like(x0) for _ in range(6)]          # g_r3, g_x2, g_r2, g_x1, g_r1, dx0
        masks_t = (r3, x2, r2, x1, r1, x0)
        st = _stream()
        if fused:
            d = _stack_desc(lys, B, Lx, pre_slope, final_act_slope, reverse=True)
            wpb = L.PtrArray6(*[bank.bwd_ptr(ly).value for ly in reversed(lys)])
            for ly in lys:
                if ly.bwd16:
                    _bank.note_std_use(bank.bwd_ptr(ly).value)
            masks = L.PtrArray6(*[t.data_ptr() for t in masks_t])
            gp = L.PtrArray6(*[g.data_ptr() for g in gouts])
            flop = sum(_conv_flop(ly, B, Lx) for ly in lys)
            check(_timed('conv1d', 7100, flop, lambda: lib.rtg_resstack_backward(C.byref(d), _p(dy), _p(y), C.byref(wpb),
                                                                                C.byref(masks), C.byref(gp), st),
                         f'dgrad {lys[0].name}..stack B{B} L{Lx}', 4 * 14 * x0.numel()), 'resstack bwd')
        else:
            g_x = dy                                              # gradient of the block's output x_j
            if final_act_slope is not None:
                g_x = torch.empty_like(dy)
                check(lib.rtg_lrelu_bwd(_p(dy), _p(y), _p(g_x), dy.numel(), final_act_slope, st), 'lrelu_bwd')
            for j, ly in enumerate(reversed(lys)):
                # odd j (a block's first conv): g_x_{j-1} = lrelu'(x) * convT(g_r) + g_x, the residual in the epilogue
                d = _dgrad_desc(ly, B, Lx, Lx, pre_slope)
                src = g_x if j % 2 == 0 else gouts[j - 1]
                res = g_x if j % 2 else None
                _run_conv(d, (_p(src), None, None, bank.bwd_ptr(ly), None, _p(masks_t[j]), _p(res), _p(gouts[j]), None, st),
                          _conv_flop(ly, B, Lx), f'dgrad {ly.name} B{B} L{Lx}', f'conv1d bwd-data {ly.name}')
                if j % 2:
                    g_x = gouts[j]
        if ctx.needs_input_grad[0]:
            ins = (x0, r1, x1, r2, x2, r3)
            dys = (gouts[4], gouts[3], gouts[2], gouts[1], gouts[0], dy)
            items = []
            for i, ly in enumerate(lys):
                last = i == 5 and final_act_slope is not None
                wd = L.WgradDesc(B=B, C1=Cc, C2=0, L_in=Lx, groups=1, Cg=Cc, Mg=Cc, K=3, stride=1, dil=ly.dil, pad=ly.pad,
                                 Q=Lx, dy_L=Lx, pre_mode=L.PRE_LRELU, pre_slope=pre_slope,
                                 gy_mode=L.PRE_MUL_DLRELU if last else L.PRE_NONE,
                                 gy_slope=final_act_slope if last else 1.0, gy_scale=1.0, splits=1, part_stride=0)
                items.append((wd, (_p(ins[i]), None, _p(dys[i]), _p(y) if last else None), ly, _conv_flop(ly, B, Lx),
                              f'wgrad {ly.name} B{B} L{Lx}', f'conv1d wgrad {ly.name}'))
            with wgrad_side(bank, (*ins, *gouts, dy, y)):
                for ly, (part, splits, immediate) in zip(lys, _run_wgrad_group(items, _stream(), bank, ctx.tok_id,
                                                                                 f'{lys[0].name}..stack')):
                    if immediate:
                        bank.flush_one(ly, part, splits)
        return None, (gouts[5] if ctx.needs_input_grad[1] else None), None, None, None, None


def resstack(token, lys, x, pre_slope, final_act_slope=None):
    return ResStackFn.apply(token, x, tuple(lys), float(pre_slope), None if final_act_slope is None else float(final_act_slope),
                            resstack_ok(lys, x))


def conv(token, ly, x1, x2=None, res=None, pre_slope=1.0, act=L.ACT_NONE, act_slope=1.0, out_scale=1.0):
    if _is_bf(x1) and not ly.maps_bf:
        x1 = decode(x1)                  # (a layer without a bf16 input path, e.g. conv_post: the fp32 feature map)
    if ly.kind == 'conv2d':
        assert x2 is None and res is None and act == L.ACT_NONE and out_scale == 1.0
        return Conv2dFn.apply(token, x1, ly, float(pre_slope))
    res_is_input = res is not None and res is x1
    return ConvFn.apply(token, x1, x2, res, ly, float(pre_slope), int(act), float(act_slope), float(out_scale),
                        res_is_input)


def _fwd2d_desc(ly, B, H, W, pre_slope):
    """descriptor of a Conv2d layer's forward on [B, C_in, H, W] -> (desc, Ho, Wo)"""
    Cin = ly.cin
    Ho = (H + 2 * ly.ph - ly.kh) // ly.sh + 1
    Wo = (W + 2 * ly.pad - ly.k) // ly.stride + 1
    pre_mode = L.PRE_LRELU if pre_slope != 1.0 else L.PRE_NONE
    d = _desc(B=B * Ho, C1=Cin * ly.kh, L_in=W, groups=1, Cg=Cin * ly.kh, Mg=ly.cout, K=ly.k, stride=ly.stride,
              pad=ly.pad, Q=Wo, out_C=ly.cout, out_L=Wo, pre_mode=pre_mode, pre_slope=pre_slope, tile_m=ly.fwd_tm,
              h_in=H, h_k=ly.kh, h_stride=ly.sh, h_pad=ly.ph, h_n=Ho, h_mode=0, bf16=ly.fwd_bf, wp16=ly.fwd16)
    if ly.fwd_khc:
        # the fragment image is kernel-row major (RtgPackJob.kh_major): the dense kernel's h_mode-2 instances read it; a
        # shape they do not serve runs on the standard image (the fragment image is not offered to h_mode 0 then)
        d.h_mode = 2
        if not _conv_native(d):
            d.h_mode, d.wp16 = 0, 0
    return d, Ho, Wo


def _dgrad2d_desc(ly, B, H, W, Ho, Wo, pre_slope):
    """... and of its backward-data: dy [B, C_out, Ho, Wo] -> dx [B, C_in, H, W]"""
    mode, g, mg, cg, k, s = ly.bwd_op
    common = dict(B=B * H, C1=ly.cout * ly.kh, L_in=Wo, groups=1, Cg=cg, Mg=mg, K=k, out_C=ly.cin, out_L=W,
                  mask_slope=pre_slope, tile_m=ly.bwd_tm, h_in=Ho, h_k=ly.kh, h_stride=ly.sh, h_pad=ly.ph,
                  h_n=H, h_mode=1, bf16=ly.bwd_bf, wp16=ly.bwd16)
    if ly.stride == 1:
        return _desc(stride=1, pad=(ly.k - 1) - ly.pad, Q=W, **common)
    return _desc(stride=1, pad=k - 1, Q=(W - 1 + ly.pad) // ly.stride + 1, shuf_S=ly.stride, shuf_P=ly.pad, **common)


class Conv2dFn(torch.autograd.Function):
    """out = conv2d(leaky_relu(x, pre_slope)) + bias for the StftDiscriminator layers (discrminator.py:255-262),
    executed by the 1-D MFMA kernels along the last axis: clips = (item, output row), channels = (c, kernel row)."""

    @staticmethod
    def forward(ctx, token, x, ly, pre_slope):
        _need_cuda(x)
        bank = token._rtg_bank
        x = _c(x)
        B, Cin, H, W = x.shape
        assert Cin == ly.cin
        d, Ho, Wo = _fwd2d_desc(ly, B, H, W, pre_slope)
        out = empty_bf((B, ly.cout, Ho, Wo), x.device) if ly.maps_bf else torch.empty(B, ly.cout, Ho, Wo, device=x.device)
        flop = 2.0 * B * Ho * Wo * ly.cout * Cin * ly.kh * ly.k
        _run_conv_t(d, x, bank.fwd_ptr(ly), bank.bias_ptr(ly), None, None, out, flop, f'fwd2d {ly.name} B{B} {H}x{W}',
                    f'conv2d fwd {ly.name}', x_slope=ENC_SLOPE, out_slope=ENC_SLOPE)
        ctx.ly, ctx.bank, ctx.tok_id, ctx.pre_slope = ly, bank, token._rtg_id, pre_slope
        ctx.save_for_backward(x)
        ctx.set_materialize_grads(False)
        return out

    @staticmethod
    def backward(ctx, dy):
        ly, bank, pre_slope = ctx.ly, ctx.bank, ctx.pre_slope
        (x,) = ctx.saved_tensors
        if dy is None:
            return None, None, None, None
        dy = _c(dy)
        B, Cin, H, W = x.shape
        _, _, Ho, Wo = dy.shape
        st = _stream()
        flop = 2.0 * B * Ho * Wo * ly.cout * Cin * ly.kh * ly.k
        dx = None
        if ctx.needs_input_grad[1]:
            dx = empty_bf(x.shape, x.device) if _is_bf(x) else torch.empty_like(x)
            mask = x if pre_slope != 1.0 else None
            d = _dgrad2d_desc(ly, B, H, W, Ho, Wo, pre_slope)
            _run_conv_t(d, dy, bank.bwd_ptr(ly), None, mask, None, dx, flop, f'dgrad2d {ly.name} B{B} {H}x{W}',
                        f'conv2d bwd-data {ly.name}')
        if ctx.needs_input_grad[0]:
            pre_mode = L.PRE_LRELU if pre_slope != 1.0 else L.PRE_NONE
            wd = L.WgradDesc(B=B * Ho, C1=Cin * ly.kh, C2=0, L_in=W, groups=1, Cg=Cin * ly.kh, Mg=ly.cout, K=ly.k,
                             stride=ly.stride, dil=1, pad=ly.pad, Q=Wo, dy_L=Wo, pre_mode=pre_mode, pre_slope=pre_slope,
                             gy_mode=L.PRE_NONE, gy_slope=1.0, gy_scale=1.0, splits=1, part_stride=0, h_in=H,
                             h_k=ly.kh, h_stride=ly.sh, h_pad=ly.ph, h_n=Ho)
            xw, dyw = _wgrad_io(wd, ly, x, dy)
            part, splits, immediate = _run_wgrad(wd, (_p(xw), None, _p(dyw), None), st, bank, ly, ctx.tok_id, flop,
                                                 f'wgrad2d {ly.name} B{B} {H}x{W}', f'conv2d wgrad {ly.name}')
            if immediate:
                bank.flush_one(ly, part, splits)
        return None, dx, None, None


# ---------------------------------------------------------------------------------------------------------------
# grouped convolution: the same layer of up to MAX_GROUP sub-discriminators in one launch (rtg_conv1d_group)
# ---------------------------------------------------------------------------------------------------------------
def _fwd_desc(ly, B, C1, L_in, pre_slope):
    """descriptor of a plain Conv1d forward (no concat, residual or output activation: what the D stacks use)"""
    L_out = _conv_out_len(ly, L_in)
    mode, g, mg, cg, k, s = ly.fwd_op
    pre_mode = L.PRE_LRELU if pre_slope != 1.0 else L.PRE_NONE
    return _desc(B=B, C1=C1, C2=0, L_in=L_in, groups=g, Cg=cg, Mg=mg, K=k, stride=ly.stride, dil=ly.dil, pad=ly.pad,
                 Q=L_out, out_C=ly.cout, out_L=L_out, pre_mode=pre_mode, pre_slope=pre_slope, tile_m=ly.fwd_tm,
                 tap_major=ly.fwd_tap, bf16=ly.fwd_bf, wp16=ly.fwd16), L_out


def _dgrad_desc(ly, B, L_in, L_out, pre_slope):
    """backward-data of the same (mask = leaky-relu derivative of the layer's input when it was pre-activated)"""
    mode, g, mg, cg, k, s = ly.bwd_op
    if ly.stride == 1:
        return _desc(B=B, C1=ly.cout, L_in=L_out, groups=g, Cg=cg, Mg=mg, K=k, stride=1, dil=ly.dil,
                     pad=(ly.k - 1) * ly.dil - ly.pad, Q=L_in, out_C=ly.cin, out_L=L_in, mask_slope=pre_slope,
                     tile_m=ly.bwd_tm, tap_major=ly.bwd_tap, bf16=ly.bwd_bf, wp16=ly.bwd16)
    nq = (L_in - 1 + ly.pad) // ly.stride + 1
    return _desc(B=B, C1=ly.cout, L_in=L_out, groups=g, Cg=cg, Mg=mg, K=k, stride=1, dil=1, pad=k - 1, Q=nq,
                 out_C=ly.cin, out_L=L_in, shuf_S=ly.stride, shuf_P=ly.pad, mask_slope=pre_slope, tile_m=ly.bwd_tm,
                 tap_major=ly.bwd_tap, bf16=ly.bwd_bf, wp16=ly.bwd16)


def _launch_group(descs, ptr_rows, flops, label, what):
    n = len(descs)
    darr = (L.Conv1dDesc * n)(*descs)
    parr = (L.ConvPtrs * n)(*[L.ConvPtrs(*[(t.data_ptr() if torch.is_tensor(t) else t) for t in row]) for row in ptr_rows])
    st = _stream()
    cfg = tune.group_cfg(darr, n, lambda: lib.rtg_conv1d_group(n, darr, parr, st))
    if cfg == 0:
        return False
    for i in range(n):
        darr[i].tile_cfg = cfg
        _bank.note_std_use(parr[i].wp)       # (group members run general block shapes: the standard images)
    check(_timed('conv1d', lib.rtg_conv1d_variant(C.byref(darr[0])) if PROFILE is not None else 0, sum(flops),
                 lambda: lib.rtg_conv1d_group(n, darr, parr, st), label), what)
    return True


class PairEntryFn(torch.autograd.Function):
    """(x_const, x_grad) -> the two halves of ONE [2B, ...] buffer (x_const first), so that the convs of a frozen stack
    can run both as a single 2B-clip launch (PairConvFn).  Gradient flows to x_grad only."""

    @staticmethod
    def forward(ctx, x_const, x_grad):
        _need_cuda(x_const, x_grad)
        assert x_const.shape == x_grad.shape
        B = x_const.shape[0]
        buf = torch.empty((2 * B,) + tuple(x_const.shape[1:]), device=x_const.device, dtype=torch.float32)
        buf[:B].copy_(x_const)
        buf[B:].copy_(x_grad)
        o_c, o_g = buf[:B], buf[B:]
        ctx.mark_non_differentiable(o_c)
        ctx.set_materialize_grads(False)
        return o_c, o_g

    @staticmethod
    def backward(ctx, d_c, d_g):
        return None, d_g


def _adjacent(a, b):
    return (a.is_contiguous() and b.is_contiguous() and a.shape == b.shape and
            b.data_ptr() == a.data_ptr() + a.numel() * a.element_size())


class PairConvFn(torch.autograd.Function):
    """The generator step runs every (frozen) discriminator on the real and on the generated clips; only the generated
    half carries a gradient.  Forward: ONE launch over both halves (they are the two halves of one buffer: same work per
    launch as the discriminator step's 2B batch instead of two half-filled grids).  Backward: backward-data of the
    generated half alone; the real half's output is non-differentiable, D's weights get no gradient here (the reference
    discards them: train.py:133)."""

    @staticmethod
    def forward(ctx, token, ly, pre_slope, tap, x_c, x_g):
        _need_cuda(x_c, x_g)
        assert _adjacent(x_c, x_g), 'PairConvFn needs the halves of one buffer (PairEntryFn)'
        assert ly.kind == 'conv'
        bank = token._rtg_bank
        B, C1, L_in = x_c.shape
        assert C1 == ly.cin
        d, L_out = _fwd_desc(ly, 2 * B, C1, L_in, pre_slope)
        out = empty_bf((2 * B, ly.cout, L_out), x_c.device) if ly.maps_bf else torch.empty(2 * B, ly.cout, L_out, device=x_c.device)
        if ly.maps_bf or _is_bf(x_c):
            whole = torch.as_strided(x_c, (2 * B, C1, L_in), x_c.stride())
            _run_conv_t(d, whole, bank.fwd_ptr(ly), bank.bias_ptr(ly), None, None, out, _conv_flop(ly, 2 * B, L_out),
                        f'fwd {ly.name} B{2 * B} L{L_in}', f'conv1d fwd {ly.name}', x_slope=ENC_SLOPE, out_slope=ENC_SLOPE)
        else:
            args = (_p(x_c), None, None, bank.fwd_ptr(ly), bank.bias_ptr(ly), None, None, _p(out), None, _stream())
            if not _gconv_forward(ly, bank, token._rtg_id, d, args, _p(x_c), 2 * B, L_in, out, pre_slope, _conv_flop(ly, 2 * B, L_out),
                                  f'fwd {ly.name} B{2 * B} L{L_in}'):
                _run_conv(d, args, _conv_flop(ly, 2 * B, L_out), f'fwd {ly.name} B{2 * B} L{L_in}', f'conv1d fwd {ly.name}')
        o_c, o_g = out[:B], out[B:]
        ctx.ly, ctx.bank, ctx.pre_slope, ctx.tok_id = ly, bank, pre_slope, token._rtg_id
        ctx.save_for_backward(x_g)
        ctx.mark_non_differentiable(o_c)
        ctx.set_materialize_grads(False)
        if tap:
            # x_g again, as an output: whoever else reads this feature map (the feature-matching loss) reads it HERE, so
            # that its gradient arrives in this node's backward and is added in the backward-data epilogue instead of by
            # an autograd accumulation kernel per feature map
            return o_c, o_g, x_g.view_as(x_g)
        return o_c, o_g

    @staticmethod
    def backward(ctx, d_c, d_g, d_tap=None):
        if ctx.needs_input_grad[0]:
            raise L.RtgError('PairConvFn is for frozen stacks: no weight gradient path')
        if not ctx.needs_input_grad[5] or (d_g is None and d_tap is None):
            return None, None, None, None, None, None
        if d_g is None:
            return None, None, None, None, None, d_tap
        ly, bank, pre_slope = ctx.ly, ctx.bank, ctx.pre_slope
        x_g, = ctx.saved_tensors
        d_g = _c(d_g)
        B, C1, L_in = x_g.shape
        L_out = d_g.shape[-1]
        dx = empty_bf(x_g.shape, x_g.device) if _is_bf(x_g) else torch.empty_like(x_g)
        d = _dgrad_desc(ly, B, L_in, L_out, pre_slope)
        res = _c(d_tap) if d_tap is not None else None            # dx = lrelu'(x) * convT(d_g) + d_tap
        if _is_bf(d_g) or _is_bf(x_g):
            _run_conv_t(d, d_g, bank.bwd_ptr(ly), None, x_g if pre_slope != 1.0 else None, res, dx, _conv_flop(ly, B, L_out),
                        f'dgrad {ly.name} B{B} L{L_in}', f'conv1d bwd-data {ly.name}')
            return None, None, None, None, None, dx
        dargs = (_p(d_g), None, None, bank.bwd_ptr(ly), None, _p(x_g) if pre_slope != 1.0 else None, _p(res), _p(dx), None,
                 _stream())
        if not _gconv_dgrad(ly, bank, ctx.tok_id, d, dargs, _p(d_g), _p(x_g), _p(res), B, L_in, L_out, dx, pre_slope,
                            _conv_flop(ly, B, L_out), f'dgrad {ly.name} B{B} L{L_in}'):
            _run_conv(d, dargs, _conv_flop(ly, B, L_out), f'dgrad {ly.name} B{B} L{L_in}', f'conv1d bwd-data {ly.name}')
        return None, None, None, None, None, dx


class PairConv2dFn(torch.autograd.Function):
    """PairConvFn for the Conv2d layers of StftDiscriminator (round 5: the generator step ran the spectrogram discriminators
    on the real and on the generated maps in two half-batch launches per layer)."""

    @staticmethod
    def forward(ctx, token, ly, pre_slope, tap, x_c, x_g):
        _need_cuda(x_c, x_g)
        assert _adjacent(x_c, x_g), 'PairConv2dFn needs the halves of one buffer (PairEntryFn)'
        assert ly.kind == 'conv2d'
        bank = token._rtg_bank
        B, Cin, H, W = x_c.shape
        assert Cin == ly.cin
        d, Ho, Wo = _fwd2d_desc(ly, 2 * B, H, W, pre_slope)
        out = empty_bf((2 * B, ly.cout, Ho, Wo), x_c.device) if ly.maps_bf else torch.empty(2 * B, ly.cout, Ho, Wo, device=x_c.device)
        whole = torch.as_strided(x_c, (2 * B, Cin, H, W), x_c.stride())
        flop = 2.0 * 2 * B * Ho * Wo * ly.cout * Cin * ly.kh * ly.k
        _run_conv_t(d, whole, bank.fwd_ptr(ly), bank.bias_ptr(ly), None, None, out, flop, f'fwd2d {ly.name} B{2 * B} {H}x{W}',
                    f'conv2d fwd {ly.name}', x_slope=ENC_SLOPE, out_slope=ENC_SLOPE)
        o_c, o_g = out[:B], out[B:]
        ctx.ly, ctx.bank, ctx.pre_slope = ly, bank, pre_slope
        ctx.save_for_backward(x_g)
        ctx.mark_non_differentiable(o_c)
        ctx.set_materialize_grads(False)
        if tap:
            return o_c, o_g, x_g.view_as(x_g)          # (see PairConvFn.forward)
        return o_c, o_g

    @staticmethod
    def backward(ctx, d_c, d_g, d_tap=None):
        if ctx.needs_input_grad[0]:
            raise L.RtgError('PairConv2dFn is for frozen stacks: no weight gradient path')
        if not ctx.needs_input_grad[5] or (d_g is None and d_tap is None):
            return None, None, None, None, None, None
        if d_g is None:
            return None, None, None, None, None, d_tap
        ly, bank, pre_slope = ctx.ly, ctx.bank, ctx.pre_slope
        x_g, = ctx.saved_tensors
        d_g = _c(d_g)
        B, Cin, H, W = x_g.shape
        _, _, Ho, Wo = d_g.shape
        dx = empty_bf(x_g.shape, x_g.device) if _is_bf(x_g) else torch.empty_like(x_g)
        d = _dgrad2d_desc(ly, B, H, W, Ho, Wo, pre_slope)
        res = _c(d_tap) if d_tap is not None else None            # dx = lrelu'(x) * convT(d_g) + d_tap
        flop = 2.0 * B * Ho * Wo * ly.cout * Cin * ly.kh * ly.k
        _run_conv_t(d, d_g, bank.bwd_ptr(ly), None, x_g if pre_slope != 1.0 else None, res, dx, flop,
                    f'dgrad2d {ly.name} B{B} {H}x{W}', f'conv2d bwd-data {ly.name}')
        return None, None, None, None, None, dx


def pair_entry(x_const, x_grad):
    return PairEntryFn.apply(x_const, x_grad)


def pair_conv(token, ly, x_c, x_g, pre_slope=1.0, tap=False):
    """-> (out_const, out_grad) and, with tap, the input x_g passed through as a third output (see PairConvFn.forward)"""
    if _is_bf(x_g) and not ly.maps_bf:
        x_c, x_g = PairDecodeFn.apply(x_c, x_g)      # (a layer without a bf16 input path, e.g. conv_post)
    fn = PairConv2dFn if ly.kind == 'conv2d' else PairConvFn
    return fn.apply(token, ly, float(pre_slope), bool(tap), x_c, x_g)


class GroupConvFn(torch.autograd.Function):
    """outs[i] = conv_i(leaky_relu(xs[i], pre_slope)) + bias_i (+ xs[i] when res_self: a ResBlock conv) for the same layer
    position of n sub-discriminators / of the parallel ResBlock branches of a UNet-G stage (same bank): forward and
    backward-data are ONE launch each for the whole group, the weight gradients go through the per-layer wgrad kernels
    into the bank as usual."""

    @staticmethod
    def forward(ctx, token, lys, pre_slope, res_self, *xs):
        _need_cuda(*xs)
        bank = token._rtg_bank
        xs = [_c(x) for x in xs]
        descs, rows, outs, flops = [], [], [], []
        for ly, x in zip(lys, xs):
            B, C1, L_in = x.shape
            assert C1 == ly.cin and ly.kind == 'conv'
            d, L_out = _fwd_desc(ly, B, C1, L_in, pre_slope)
            out = torch.empty(B, ly.cout, L_out, device=x.device, dtype=torch.float32)
            descs.append(d); outs.append(out); flops.append(_conv_flop(ly, B, L_out))
            rows.append((x, None, None, bank.fwd_ptr(ly).value, bank.bias_ptr(ly).value, None, x if res_self else None, out,
                         None))
        if not _launch_group(descs, rows, flops, f'fwd group {lys[0].name} x{len(lys)}', f'conv1d group fwd {lys[0].name}'):
            raise L.RtgError(f'no common block shape for the group of {lys[0].name}')
        ctx.lys, ctx.bank, ctx.tok_id, ctx.pre_slope, ctx.res_self = lys, bank, token._rtg_id, pre_slope, res_self
        ctx.save_for_backward(*xs)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *dys):
        lys, bank, pre_slope = ctx.lys, ctx.bank, ctx.pre_slope
        xs = ctx.saved_tensors
        dys = [_c(dy) for dy in dys]
        st = _stream()
        need_x = any(ctx.needs_input_grad[4:])
        dxs = [None] * len(xs)
        if need_x:
            descs, rows, flops = [], [], []
            for i, (ly, x, dy) in enumerate(zip(lys, xs, dys)):
                B, C1, L_in = x.shape
                L_out = dy.shape[-1]
                dxs[i] = torch.empty_like(x)
                descs.append(_dgrad_desc(ly, B, L_in, L_out, pre_slope))
                flops.append(_conv_flop(ly, B, L_out))
                # res_self: dx = dy + lrelu'(x) * convT(dy), the residual branch rides the same epilogue
                rows.append((dy, None, None, bank.bwd_ptr(ly).value, None, x if pre_slope != 1.0 else None,
                             dy if ctx.res_self else None, dxs[i], None))
            if not _launch_group(descs, rows, flops, f'dgrad group {lys[0].name} x{len(lys)}',
                                 f'conv1d group bwd-data {lys[0].name}'):
                raise L.RtgError(f'no common block shape for the dgrad group of {lys[0].name}')
        if ctx.needs_input_grad[0]:
            pre_mode = L.PRE_LRELU if pre_slope != 1.0 else L.PRE_NONE

            items = []
            for ly, x, dy in zip(lys, xs, dys):
                B, C1, L_in = x.shape
                L_out = dy.shape[-1]
                wd = L.WgradDesc(B=B, C1=C1, C2=0, L_in=L_in, groups=ly.groups, Cg=ly.cin // ly.groups,
                                 Mg=ly.cout // ly.groups, K=ly.k, stride=ly.stride, dil=ly.dil, pad=ly.pad, Q=L_out,
                                 dy_L=L_out, pre_mode=pre_mode, pre_slope=pre_slope, gy_mode=L.PRE_NONE, gy_slope=1.0,
                                 gy_scale=1.0, splits=1, part_stride=0)
                items.append((wd, (_p(x), None, _p(dy), None), ly, _conv_flop(ly, B, L_out),
                              f'wgrad {ly.name} B{B} L{L_in}', f'conv1d wgrad {ly.name}'))
            # one launch for the group where the tuner found that faster, else layer by layer
            # (forking the layers over streams measured slower: 43.9 vs 42.7 ms/step)
            with wgrad_side(bank, (*xs, *dys)):
                for ly, (part, splits, immediate) in zip(lys, _run_wgrad_group(items, _stream(), bank, ctx.tok_id,
                                                                                 lys[0].name)):
                    if immediate:
                        bank.flush_one(ly, part, splits)
        return (None, None, None, None, *dxs)


def group_conv(token, lys, xs, pre_slope=1.0, res_self=False):
    """lys: rtg.bank.ConvLayer of each group member (same position in sibling sub-discriminators); xs: their inputs"""
    return list(GroupConvFn.apply(token, tuple(lys), float(pre_slope), bool(res_self), *xs))


# MRF_GROUP: the parallel ResBlock branches of a UNet-G decoder stage (kernel sizes 3 / 5 / 7 on the same input) as
# grouped launches, conv by conv.  1: stages of >= 64 channels, where a branch alone is 256-1024 workgroups of
# 25-50 us and three of them share one launch's fixed cost; the 32-channel stage is HBM-bound and keeps its
# weights-in-registers kernel (rtg_resconv) per branch.  0: never (forked streams), 2: every stage.
MRF_GROUP = 1


_GROUP_OK = {}


def _common_group_code(descs):
    """True when the members of a grouped launch share at least one general block shape (what tune.group_cfg picks from)"""
    lists = []
    for d in descs:
        d.tile_cfg = 0
        cands = (C.c_int * tune.MAX_CANDS)()               # the same list tune.group_cfg intersects
        k = lib.rtg_conv1d_tile_candidates(C.byref(d), cands, tune.MAX_CANDS)
        lists.append({c for c in cands[:max(k, 0)] if 0 < c < 7000})
    return bool(set.intersection(*lists)) if lists else False


def mrf_group_ok(lys, x):
    if MRF_GROUP == 0 or len(lys) < 2 or len(lys) > L.MAX_GROUP:
        return False
    c = lys[0].cin
    if any(ly.kind != 'conv' or ly.stride != 1 or ly.groups != 1 or ly.cin != c or ly.cout != c or
           2 * ly.pad != ly.dil * (ly.k - 1) for ly in lys):
        return False
    if not (c >= 64 or MRF_GROUP == 2):
        return False
    # the k3 / k5 / k7 members must also share a block shape, forward and backward-data, at THIS batch and length
    # (GroupConvFn raises otherwise): any other segment length or batch falls back to the forked branches
    B, _, L_in = x.shape
    # (keyed on the problem, not on object identity: a freed model's layer id can be reused by the next model's)
    key = (tuple((ly.lid, ly.k, ly.dil, ly.pad, ly.fwd_bf, ly.bwd_bf) for ly in lys), c, B, L_in)
    ok = _GROUP_OK.get(key)
    if ok is None:
        ok = _common_group_code([_fwd_desc(ly, B, c, L_in, 0.15)[0] for ly in lys]) and \
            _common_group_code([_dgrad_desc(ly, B, L_in, L_in, 0.15) for ly in lys])
        _GROUP_OK[key] = ok
    return ok


# ---------------------------------------------------------------------------------------------------------------
# GaussianNoise
# ---------------------------------------------------------------------------------------------------------------
# The shared scalar's gradient is added straight into the bank's gradient slot by a launch of ours — but only inside a
# backward that asked for it (train.Trainer.g_step wraps total.backward() in noise_grad_accumulate()): any other
# differentiation (torch.autograd.grad(...), a user's own backward) gets dw returned like every other gradient and
# w.grad stays untouched.
_NOISE_ACC_DEPTH = 0


class noise_grad_accumulate:
    """context: GaussianNoise backward passes inside accumulate d noise.w into noise.w.grad in place (one launch of ours
    instead of an ATen sum + add_ per noise call) and return no gradient for it"""

    def __enter__(self):
        global _NOISE_ACC_DEPTH
        _NOISE_ACC_DEPTH += 1
        return self

    def __exit__(self, *exc):
        global _NOISE_ACC_DEPTH
        _NOISE_ACC_DEPTH -= 1
        return False


class NoiseFn(torch.autograd.Function):
    N_BLOCKS = 2048         # (256: one block per CU streamed 1.2 TB/s; the kernel is a pure 12-byte-per-element stream)

    @staticmethod
    def forward(ctx, x, w, u_in, slope, seed, salt):
        _need_cuda(x, w, u_in)
        x = _c(x)
        out = torch.empty_like(x)
        check(lib.rtg_noise_lrelu_fwd(_p(x), _p(w), _p(u_in), _p(out), x.numel(), slope, C.c_ulonglong(seed),
                                      _p(salt), _stream()), 'noise fwd')
        ctx.save_for_backward(x, w, u_in, salt)
        ctx.cfg = (slope, seed)
        return out

    @staticmethod
    def backward(ctx, dy):
        x, w, u_in, salt = ctx.saved_tensors
        slope, seed = ctx.cfg
        dy = _c(dy)
        dx = torch.empty_like(x)
        part = torch.empty(NoiseFn.N_BLOCKS, device=x.device)
        # the shared scalar's gradient: where w.grad is a live buffer (the weight bank's flat gradient view) a one-block
        # launch adds the partials into it in fixed order and autograd gets no gradient to accumulate (six ATen sum + five
        # grad add_ launches per generator backward -> six launches of ours)
        g = w.grad if ctx.needs_input_grad[1] and _NOISE_ACC_DEPTH > 0 and not torch.is_grad_enabled() else None
        if g is not None and g.is_cuda and g.dtype == torch.float32 and g.numel() == 1:
            check(lib.rtg_noise_lrelu_bwd_acc(_p(x), _p(w), _p(u_in), _p(dy), _p(dx), _p(part), NoiseFn.N_BLOCKS, x.numel(),
                                              slope, C.c_ulonglong(seed), _p(salt), _p(g), _stream()), 'noise bwd')
            return dx, None, None, None, None, None
        check(lib.rtg_noise_lrelu_bwd(_p(x), _p(w), _p(u_in), _p(dy), _p(dx), _p(part), NoiseFn.N_BLOCKS, x.numel(),
                                      slope, C.c_ulonglong(seed), _p(salt), _stream()), 'noise bwd')
        dw = part.sum().reshape(w.shape) if ctx.needs_input_grad[1] else None
        return dx, dw, None, None, None, None


# ---------------------------------------------------------------------------------------------------------------
# small layers
# ---------------------------------------------------------------------------------------------------------------
class AvgPoolFn(torch.autograd.Function):
    """nn.AvgPool1d(4, 2, 1) (retunegan/models/discrminator.py:113)."""

    @staticmethod
    def forward(ctx, x):
        _need_cuda(x)
        x = _c(x)
        B, Cc, Lx = x.shape
        out = torch.empty(B, Cc, Lx // 2, device=x.device)
        check(lib.rtg_avgpool4s2_fwd(_p(x), _p(out), B * Cc, Lx, _stream()), 'avgpool fwd')
        ctx.shape = (B, Cc, Lx)
        return out

    @staticmethod
    def backward(ctx, dy):
        B, Cc, Lx = ctx.shape
        dy = _c(dy)
        dx = torch.empty(B, Cc, Lx, device=dy.device)
        check(lib.rtg_avgpool4s2_bwd(_p(dy), _p(dx), B * Cc, Lx, _stream()), 'avgpool bwd')
        return dx


class PeriodFoldFn(torch.autograd.Function):
    """y [B,1,T] -> [B*p, 1, H]: reflect-pad the tail to a multiple of p and put the period axis into the batch
    (retunegan/models/discrminator.py:203-210 builds [B,1,H,p]; the (k,1) convs then act along H only)."""

    @staticmethod
    def forward(ctx, y, p):
        _need_cuda(y)
        y = _c(y)
        B, _, T = y.shape
        H = (T + p - 1) // p
        out = torch.empty(B * p, 1, H, device=y.device)
        check(lib.rtg_period_fold_fwd(_p(y), _p(out), B, T, p, H, _stream()), 'fold fwd')
        ctx.cfg = (B, T, p, H)
        return out

    @staticmethod
    def backward(ctx, dout):
        B, T, p, H = ctx.cfg
        dout = _c(dout)
        dy = torch.empty(B, 1, T, device=dout.device)
        check(lib.rtg_period_fold_bwd(_p(dout), _p(dy), B, T, p, H, _stream()), 'fold bwd')
        return dy, None


# ---------------------------------------------------------------------------------------------------------------
# losses (multi-tensor)
# ---------------------------------------------------------------------------------------------------------------
def _jobs(entries):
    arr = (L.LossJob * len(entries))()
    for i, (a, b, da, db, w, target) in enumerate(entries):
        arr[i] = L.LossJob(a.data_ptr(), b.data_ptr() if b is not None else None,
                           da.data_ptr() if da is not None else None, db.data_ptr() if db is not None else None,
                           a.numel(), w, target)
    return arr


class MultiLossFn(torch.autograd.Function):
    """sum_j w_j * mean(term(a_j, b_j)) over a list of tensor pairs, one launch per <= 48 pairs.
    Inputs are passed flat: a_0..a_{n-1}, b_0..b_{n-1}."""

    @staticmethod
    def forward(ctx, kind, weights, target, n, *tensors):
        a_list = [_c(t) for t in tensors[:n]]
        b_list = [_c(t) if t is not None else None for t in tensors[n:]]
        _need_cuda(*a_list)
        want = _BF if kind == L.LOSS_L1_ENC else torch.float32
        if any(t.dtype != want for t in a_list) or any(t is not None and t.dtype != want for t in b_list):
            raise L.RtgError(f'multi_loss kind {kind}: expects {want} tensors (bf16 feature maps go through LOSS_L1_ENC or ops.decode)')
        dev = a_list[0].device
        loss = torch.zeros(1, device=dev)
        ws = torch.empty(64 * L.MAX_LOSS_JOBS, device=dev)
        st = _stream()
        for s in range(0, n, L.MAX_LOSS_JOBS):
            ent = [(a_list[i], b_list[i], None, None, weights[i], target) for i in range(s, min(n, s + L.MAX_LOSS_JOBS))]
            check(lib.rtg_loss_fwd(kind, _jobs(ent), len(ent), _p(ws), _p(loss), st), 'loss fwd')
        ctx.cfg = (kind, weights, target, n)
        ctx.save_for_backward(*a_list, *[b for b in b_list if b is not None])
        ctx.has_b = [b is not None for b in b_list]
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        kind, weights, target, n = ctx.cfg
        saved = list(ctx.saved_tensors)
        a_list = saved[:n]
        rest = saved[n:]
        b_list = []
        for hb in ctx.has_b:
            b_list.append(rest.pop(0) if hb else None)
        g = _c(g).reshape(1)
        st = _stream()
        # (RTG_LOSS_L1_ENC: bf16 feature maps in, bf16 gradients out)
        like = lambda t: empty_bf(t.shape, t.device) if _is_bf(t) else torch.empty_like(t)      # noqa: E731
        da = [like(a) if ctx.needs_input_grad[4 + i] else None for i, a in enumerate(a_list)]
        db = [like(b) if (b is not None and ctx.needs_input_grad[4 + n + i]) else None for i, b in enumerate(b_list)]
        idx = [i for i in range(n) if da[i] is not None or db[i] is not None]
        for s in range(0, len(idx), L.MAX_LOSS_JOBS):
            ent = [(a_list[i], b_list[i], da[i], db[i], weights[i], target) for i in idx[s:s + L.MAX_LOSS_JOBS]]
            check(lib.rtg_loss_bwd(kind, _jobs(ent), len(ent), _p(g), st), 'loss bwd')
        return (None, None, None, None, *da, *db)


class PairLossFn(torch.autograd.Function):
    """discriminator_loss over logits that are still the two halves (real, generated) of one 2B-clip tensor each:
    sum_k mean((1 - r_k)^2) + mean(g_k^2), or with `relative` mean((1 - (r_k - g_k.detach()))^2) for the real term
    (loss.py:102-125).  One node over the whole tensors: the backward writes both halves of the gradient in place, no
    slice / zero-fill / add kernels per sub-discriminator."""

    @staticmethod
    def forward(ctx, relative, *parents):
        ps = [_c(p) for p in parents]
        _need_cuda(*ps)
        assert all(p.shape[0] % 2 == 0 for p in ps) and 2 * len(ps) <= L.MAX_LOSS_JOBS
        loss = torch.zeros(1, device=ps[0].device)
        ws = torch.empty(64 * L.MAX_LOSS_JOBS, device=ps[0].device)
        st = _stream()
        halves = [(p[:p.shape[0] // 2], p[p.shape[0] // 2:]) for p in ps]
        fake = [(g, None, None, None, 1.0, 0.0) for _, g in halves]
        if relative:
            real = [(r, g, None, None, 1.0, 1.0) for r, g in halves]
            check(lib.rtg_loss_fwd(L.LOSS_MSE_REL, _jobs(real), len(real), _p(ws), _p(loss), st), 'loss fwd')
            check(lib.rtg_loss_fwd(L.LOSS_MSE_TARGET, _jobs(fake), len(fake), _p(ws), _p(loss), st), 'loss fwd')
        else:
            ent = [(r, None, None, None, 1.0, 1.0) for r, _ in halves] + fake
            check(lib.rtg_loss_fwd(L.LOSS_MSE_TARGET, _jobs(ent), len(ent), _p(ws), _p(loss), st), 'loss fwd')
        ctx.relative = relative
        ctx.save_for_backward(*ps)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        ps = ctx.saved_tensors
        g = _c(g).reshape(1)
        st = _stream()
        dps = [torch.empty_like(p) for p in ps]
        hv = [(p[:p.shape[0] // 2], p[p.shape[0] // 2:], d[:p.shape[0] // 2], d[p.shape[0] // 2:]) for p, d in zip(ps, dps)]
        fake = [(gg, None, dg, None, 1.0, 0.0) for _, gg, _, dg in hv]
        if ctx.relative:
            real = [(r, gg, dr, None, 1.0, 1.0) for r, gg, dr, _ in hv]
            check(lib.rtg_loss_bwd(L.LOSS_MSE_REL, _jobs(real), len(real), _p(g), st), 'loss bwd')
            check(lib.rtg_loss_bwd(L.LOSS_MSE_TARGET, _jobs(fake), len(fake), _p(g), st), 'loss bwd')
        else:
            ent = [(r, None, dr, None, 1.0, 1.0) for r, _, dr, _ in hv] + fake
            check(lib.rtg_loss_bwd(L.LOSS_MSE_TARGET, _jobs(ent), len(ent), _p(g), st), 'loss bwd')
        return (None, *dps)


def pair_loss(parents, relative=False):
    return PairLossFn.apply(bool(relative), *parents)


class WSumFn(torch.autograd.Function):
    """total = sum_i w_i * term_i over device scalars, in list order: one launch forward, one backward (the fan-out of the
    incoming gradient), instead of an ATen mul / add / fill per term of the loss totals (train.py:137-158, 170-189)."""

    @staticmethod
    def forward(ctx, weights, *terms):
        _need_cuda(*terms)
        ts = [_c(t).reshape(1) for t in terms]
        tab = L.ScalarTerms()
        tab.n = len(ts)
        for i, (t, w) in enumerate(zip(ts, weights)):
            tab.p[i] = t.data_ptr()
            tab.w[i] = w
        out = torch.empty(1, device=ts[0].device)
        check(lib.rtg_scalar_wsum(C.byref(tab), _p(out), _stream()), 'scalar wsum')
        ctx.weights = weights
        ctx.shapes = [t.shape for t in terms]
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        n = len(ctx.weights)
        tab = L.ScalarTerms()
        tab.n = n
        for i, w in enumerate(ctx.weights):
            tab.w[i] = w
        g = _c(g).reshape(1)
        out = torch.empty(n, device=g.device)
        check(lib.rtg_scalar_fanout(C.byref(tab), _p(g), _p(out), _stream()), 'scalar fan-out')
        return (None, *[out[i].reshape(ctx.shapes[i]) if ctx.needs_input_grad[1 + i] else None for i in range(n)])


def weighted_sum(terms, weights=None):
    """sum_i weights[i] * terms[i] of device scalars (the loss totals of the train step) as one autograd node"""
    terms = list(terms)
    weights = [1.0] * len(terms) if weights is None else [float(w) for w in weights]
    assert len(terms) == len(weights) and len(terms) >= 1
    if len(terms) > L.MAX_SCALAR_TERMS or not all(t.is_cuda and t.dtype == torch.float32 and t.numel() == 1 for t in terms):
        total = terms[0] * weights[0]
        for t, w in zip(terms[1:], weights[1:]):
            total = total + t * w
        return total
    return WSumFn.apply(weights, *terms)


def multi_loss(kind, a_list, b_list=None, weights=None, target=0.0):
    n = len(a_list)
    if b_list is None:
        b_list = [None] * n
    if weights is None:
        weights = [1.0] * n
    return MultiLossFn.apply(kind, [float(w) for w in weights], float(target), n, *a_list, *b_list)


class DynLossFn(torch.autograd.Function):
    """dynamic_loss (retunegan/models/loss.py:76-82) or, with env=True, envelope_loss (loss.py:66-72); gradient w.r.t.
    the generated wave only."""

    @staticmethod
    def forward(ctx, y, g, k, env=False):
        _need_cuda(y, g)
        y, g = _c(y), _c(g)
        rows, Lx = y.numel() // y.shape[-1], y.shape[-1]
        loss = torch.zeros(1, device=y.device)
        ws = torch.empty(256, device=y.device)
        fwd = lib.rtg_env_loss_fwd if env else lib.rtg_dyn_loss_fwd
        check(fwd(_p(y), _p(g), rows, Lx, k, 1.0, _p(ws), _p(loss), _stream()), 'dyn/env fwd')
        ctx.save_for_backward(y, g)
        ctx.cfg = (rows, Lx, k, env)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gout):
        y, g = ctx.saved_tensors
        rows, Lx, k, env = ctx.cfg
        dg = torch.empty_like(g)
        bwd = lib.rtg_env_loss_bwd if env else lib.rtg_dyn_loss_bwd
        check(bwd(_p(y), _p(g), rows, Lx, k, 1.0, _p(_c(gout).reshape(1)), _p(dg), _stream()), 'dyn/env bwd')
        return None, dg, None, None


class StripMirrorFn(torch.autograd.Function):
    """strip_mirror_loss (retunegan/models/loss.py:86-98) of a wave [B, 1, T]."""

    @staticmethod
    def forward(ctx, y):
        _need_cuda(y)
        y = _c(y)
        rows, Lx = y.numel() // y.shape[-1], y.shape[-1]
        loss = torch.zeros(1, device=y.device)
        ws = torch.empty(256, device=y.device)
        stats = torch.zeros(4, device=y.device)
        check(lib.rtg_strip_mirror_fwd(_p(y), rows, Lx, 1.0, _p(ws), _p(stats), _p(loss), _stream()), 'strip-mirror fwd')
        ctx.save_for_backward(y, stats)
        ctx.cfg = (rows, Lx)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gout):
        y, stats = ctx.saved_tensors
        rows, Lx = ctx.cfg
        dy = torch.empty_like(y)
        check(lib.rtg_strip_mirror_bwd(_p(y), rows, Lx, 1.0, _p(stats), _p(_c(gout).reshape(1)), _p(dy), _stream()),
              'strip-mirror bwd')
        return dy


# ---------------------------------------------------------------------------------------------------------------
# STFT
# ---------------------------------------------------------------------------------------------------------------
SPEC_FREQ_MAJOR = True     # the spectrogram maps are kept [B, 2, frames, F] in HBM (StftFn hands out the transposed view)


class MultiStftFn(torch.autograd.Function):
    """The resolutions `plans` of multi_stft_loss (retunegan/models/loss.py:30-52) on the generated wave y [B,T] — and, with
    `y_real` [B,T] (a constant), on the real wave too — in ONE forward launch (rtg_stft_forward_multi); the backward is one frame
    launch plus one overlap-add over all resolutions (rtg_stft_backward_multi).  Returns (mel_0, .., spec_0 | None, ..) of y,
    then the same of y_real (non-differentiable) when given.  spec as in StftFn."""

    @staticmethod
    def forward(ctx, y, y_real, want_spec, *plans):
        _need_cuda(y)
        y = _c(y)
        sigs = [y] + ([_c(y_real)] if y_real is not None else [])
        need_bwd = ctx.needs_input_grad[0]
        n = len(plans)
        if n * len(sigs) > L.STFT_MAX_JOBS:
            raise L.RtgError(f'multi-STFT: {n} resolutions x {len(sigs)} waves exceed {L.STFT_MAX_JOBS} jobs per launch')
        jobs = (L.StftFwdJob * (n * len(sigs)))()
        outs, saved, nbytes = [], [], 0
        spec_t = int(SPEC_FREQ_MAJOR)
        for si, sig in enumerate(sigs):
            B, T = sig.shape
            dev = sig.device
            mels, specs = [], []
            for pi, plan in enumerate(plans):
                frames, F = 1 + T // plan.hop, plan.n_fft // 2 + 1
                t = plan.tensors(dev)
                mel = torch.empty(B, plan.n_mel, frames, device=dev)
                spec = (torch.empty(B, 2, frames, F, device=dev) if spec_t else torch.empty(B, 2, F, frames, device=dev)) if want_spec else None
                keep = need_bwd and si == 0
                re = torch.empty(B, frames, F, device=dev) if keep else None
                im = torch.empty(B, frames, F, device=dev) if keep else None
                j = jobs[si * n + pi]
                j.d = L.StftDesc(B, T, plan.n_fft, plan.win, plan.hop, frames, plan.n_mel, spec_t)
                j.y, j.window, j.twiddle = sig.data_ptr(), t['window'].data_ptr(), t['twiddle'].data_ptr()
                j.mel_lo, j.mel_len, j.mel_woff, j.mel_w = (t[k].data_ptr() for k in ('mel_lo', 'mel_len', 'mel_woff', 'mel_w'))
                j.mel, j.spec = mel.data_ptr(), (spec.data_ptr() if want_spec else None)
                j.re, j.im = (re.data_ptr(), im.data_ptr()) if keep else (None, None)
                nbytes += 4 * (sig.numel() + mel.numel() + (spec.numel() if want_spec else 0) + (2 * re.numel() if keep else 0))
                mels.append(mel); specs.append(spec)
                if keep:
                    saved += [re, im]
            outs.append((mels, specs))
        B, T = y.shape
        check(timed_bw('stft_fwd', nbytes, lambda: lib.rtg_stft_forward_multi(len(jobs), jobs, _stream()),
                       f'{n} resolutions x {len(sigs)} waves B{B} T{T}'), 'stft fwd (multi)')
        ctx.plans, ctx.shape, ctx.spec_t, ctx.want_spec = plans, (B, T), spec_t, want_spec
        ctx.save_for_backward(*saved)
        ctx.set_materialize_grads(False)
        flat = []
        for si, (mels, specs) in enumerate(outs):
            flat += mels + specs
            if si == 1:
                ctx.mark_non_differentiable(*[t_ for t_ in mels + specs if t_ is not None])
        return tuple(flat)

    @staticmethod
    def backward(ctx, *grads):
        plans, (B, T), n = ctx.plans, ctx.shape, len(ctx.plans)
        saved = ctx.saved_tensors
        dmels, dspecs = grads[:n], grads[n:2 * n]
        live = [i for i in range(n) if dmels[i] is not None or dspecs[i] is not None]
        if not live:
            return (None,) * (3 + n)
        dev = saved[0].device
        jobs = (L.StftBwdJob * len(live))()
        hold, nbytes = [], 0
        for k, i in enumerate(live):
            plan = plans[i]
            frames = 1 + T // plan.hop
            t = plan.tensors(dev)
            re, im = saved[2 * i], saved[2 * i + 1]
            dm, ds = _c(dmels[i]), _c(dspecs[i])
            ws = torch.empty(B * frames * plan.win, device=dev)
            hold += [dm, ds, ws]
            j = jobs[k]
            j.d = L.StftDesc(B, T, plan.n_fft, plan.win, plan.hop, frames, plan.n_mel, ctx.spec_t)
            j.re, j.im = re.data_ptr(), im.data_ptr()
            j.dmel, j.dspec = (dm.data_ptr() if dm is not None else None), (ds.data_ptr() if ds is not None else None)
            j.window, j.twiddle = t['window'].data_ptr(), t['twiddle'].data_ptr()
            j.binmel_idx, j.binmel_w, j.frame_ws = t['binmel_idx'].data_ptr(), t['binmel_w'].data_ptr(), ws.data_ptr()
            nbytes += 4 * (2 * re.numel() + (dm.numel() if dm is not None else 0) + (ds.numel() if ds is not None else 0))
        dy = torch.empty(B, T, device=dev)
        nbytes += 4 * dy.numel()
        check(timed_bw('stft_bwd', nbytes, lambda: lib.rtg_stft_backward_multi(len(live), jobs, _p(dy), 0, _stream()),
                       f'{len(live)} resolutions B{B} T{T}'), 'stft bwd (multi)')
        return (dy, None, None) + (None,) * n


class StftFn(torch.autograd.Function):
    """y [B,T] -> (mel [B,n_mel,frames], spec or None) for one resolution (see rtg_stft_forward).  spec: [B,2,F,frames], or with
    SPEC_FREQ_MAJOR the tensor [B,2,frames,F] whose transposed view is that map (audio.stft_mel_spec hands the view out)."""

    @staticmethod
    def forward(ctx, y, plan, want_spec):
        _need_cuda(y)
        y = _c(y)
        B, T = y.shape
        frames = 1 + T // plan.hop
        F = plan.n_fft // 2 + 1
        dev = y.device
        t = plan.tensors(dev)
        need_bwd = ctx.needs_input_grad[0]
        mel = torch.empty(B, plan.n_mel, frames, device=dev)
        spec_t = int(SPEC_FREQ_MAJOR)
        spec = (torch.empty(B, 2, frames, F, device=dev) if spec_t else torch.empty(B, 2, F, frames, device=dev)) if want_spec else None
        re = torch.empty(B, frames, F, device=dev) if need_bwd else None
        im = torch.empty(B, frames, F, device=dev) if need_bwd else None
        d = L.StftDesc(B, T, plan.n_fft, plan.win, plan.hop, frames, plan.n_mel, spec_t)
        nbytes = 4 * (y.numel() + mel.numel() + (spec.numel() if want_spec else 0) + (2 * re.numel() if need_bwd else 0))
        check(timed_bw('stft_fwd', nbytes, lambda: lib.rtg_stft_forward(
            C.byref(d), _p(y), _p(t['window']), _p(t['twiddle']), _p(t['mel_lo']), _p(t['mel_len']), _p(t['mel_woff']),
            _p(t['mel_w']), _p(mel), _p(spec), _p(re), _p(im), _stream()), f'n_fft {plan.n_fft} B{B} T{T}'), 'stft fwd')
        ctx.plan, ctx.shape, ctx.spec_t = plan, (B, T, frames), spec_t
        ctx.save_for_backward(re, im)
        ctx.set_materialize_grads(False)
        return mel, spec

    @staticmethod
    def backward(ctx, dmel, dspec):
        re, im = ctx.saved_tensors
        plan = ctx.plan
        B, T, frames = ctx.shape
        if dmel is None and dspec is None:
            return None, None, None
        dev = re.device
        t = plan.tensors(dev)
        dy = torch.zeros(B, T, device=dev)
        ws = torch.empty(B * frames * plan.win, device=dev)
        d = L.StftDesc(B, T, plan.n_fft, plan.win, plan.hop, frames, plan.n_mel, ctx.spec_t)
        dmel_c, dspec_c = _c(dmel), _c(dspec)
        nbytes = 4 * (2 * re.numel() + (dmel_c.numel() if dmel_c is not None else 0) +
                      (dspec_c.numel() if dspec_c is not None else 0) + dy.numel())
        check(timed_bw('stft_bwd', nbytes, lambda: lib.rtg_stft_backward(
            C.byref(d), _p(re), _p(im), _p(dmel_c), _p(dspec_c), _p(t['window']), _p(t['twiddle']), _p(t['binmel_idx']),
            _p(t['binmel_w']), _p(ws), _p(dy), _stream()), f'n_fft {plan.n_fft} B{B} T{T}'), 'stft bwd')
        return dy, None, None
