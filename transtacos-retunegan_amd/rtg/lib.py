"""ctypes loader for librtg.so (include/rtg.h).  Fails loudly when the library is missing: there is no fallback."""
import ctypes as C
import os

import torch  # noqa: F401  FIRST: librtg.so must bind to the HIP runtime PyTorch ships and initialises (loading the
#                     library before torch pulls in /opt/rocm's libamdhip64 instead, and every launch then fails with
#                     hipErrorNoDevice once torch has set the device up through its own copy)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('RTG_DEV_LIB') or os.path.join(os.path.dirname(_HERE), 'librtg.so')   # (rtg/config.py: development builds)


class RtgError(RuntimeError):
    pass


class Conv1dDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in ('B', 'C1', 'C2', 'L_in', 'groups', 'Cg', 'Mg', 'K', 'stride', 'dil', 'pad', 'Q',
                                       'out_C', 'out_L', 'shuf_S', 'shuf_P', 'pre_mode')] + \
               [('pre_slope', C.c_float), ('mask_slope', C.c_float), ('out_scale', C.c_float), ('act', C.c_int),
                ('act_slope', C.c_float), ('accumulate', C.c_int), ('tile_m', C.c_int), ('out_split', C.c_int)] + \
               [(n, C.c_int) for n in ('h_in', 'h_k', 'h_stride', 'h_pad', 'h_n', 'h_mode', 'tap_major', 'tile_cfg', 'bf16',
                                       'wp16', 'io_bf16')] + [('enc_slope', C.c_float)]


class ConvPtrs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ('x1', 'x2', 'aux', 'wp', 'bias', 'mask', 'res', 'out', 'out2')]


MAX_GROUP = 4
WGRAD_MAX_GROUP = 6


class WgradDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in ('B', 'C1', 'C2', 'L_in', 'groups', 'Cg', 'Mg', 'K', 'stride', 'dil', 'pad', 'Q',
                                       'dy_L', 'pre_mode')] + \
               [('pre_slope', C.c_float), ('gy_mode', C.c_int), ('gy_slope', C.c_float), ('gy_scale', C.c_float),
                ('splits', C.c_int),
                ('part_stride', C.c_longlong)] + [(n, C.c_int) for n in ('h_in', 'h_k', 'h_stride', 'h_pad', 'h_n', 'shape_cfg', 'bf16',
                                                                         'io_bf16')]


class GconvDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in ('B', 'groups', 'Cg', 'Mg', 'K', 'stride', 'pad', 'L_in', 'L_out')] + \
               [('pre_slope', C.c_float)]


class WgradPtrs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ('x1', 'x2', 'dy', 'gy_aux', 'part')]


class NormJob(C.Structure):
    _fields_ = [('g_off', C.c_longlong), ('v_off', C.c_longlong), ('scale_off', C.c_longlong), ('rows', C.c_int),
                ('inner', C.c_int)]


class PackJob(C.Structure):
    _fields_ = [('v_off', C.c_longlong), ('scale_off', C.c_longlong), ('dst_off', C.c_longlong),
                ('dst_size', C.c_longlong)] + \
               [(n, C.c_int) for n in ('mode', 'groups', 'Mg', 'Cg', 'K', 'src_K', 'src_inner_c', 'S', 'tile_m', 'KH', 'tap_major', 'bf16',
                                       'frag16', 'first_block', 'n_blocks', 'src_T', 'kh_major')]


MAX_SCALAR_TERMS = 16


class ScalarTerms(C.Structure):
    _fields_ = [('p', C.c_void_p * MAX_SCALAR_TERMS), ('w', C.c_float * MAX_SCALAR_TERMS), ('n', C.c_int)]


class WnBwdJob(C.Structure):
    _fields_ = [('g_off', C.c_longlong), ('v_off', C.c_longlong), ('b_off', C.c_longlong), ('scale_off', C.c_longlong),
                ('part_off', C.c_longlong), ('part_stride', C.c_longlong), ('splits', C.c_int), ('rows', C.c_int),
                ('inner', C.c_int), ('t_rows', C.c_int), ('t_taps', C.c_int)]


class LossJob(C.Structure):
    _fields_ = [('a', C.c_void_p), ('b', C.c_void_p), ('da', C.c_void_p), ('db', C.c_void_p), ('n', C.c_longlong),
                ('w', C.c_float), ('target', C.c_float)]


class ResStackDesc(C.Structure):
    _fields_ = [('B', C.c_int), ('C', C.c_int), ('L', C.c_int), ('dil', C.c_int * 6), ('pre_slope', C.c_float),
                ('final_act', C.c_int), ('act_slope', C.c_float)]


PtrArray6 = C.c_void_p * 6


class StftDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in ('B', 'T', 'n_fft', 'win', 'hop', 'frames', 'n_mel', 'spec_T')]


STFT_MAX_JOBS = 8


class StftFwdJob(C.Structure):      # RtgStftFwdJob (ABI 11)
    _fields_ = [('d', StftDesc)] + [(n, C.c_void_p) for n in ('y', 'window', 'twiddle', 'mel_lo', 'mel_len', 'mel_woff', 'mel_w',
                                                               'mel', 'spec', 're', 'im')]


class StftBwdJob(C.Structure):      # RtgStftBwdJob (ABI 11)
    _fields_ = [('d', StftDesc)] + [(n, C.c_void_p) for n in ('re', 'im', 'dmel', 'dspec', 'window', 'twiddle', 'binmel_idx',
                                                               'binmel_w', 'frame_ws')]


PRE_NONE, PRE_LRELU, PRE_MUL_DLRELU, PRE_MUL_DTANH = 0, 1, 2, 3
ACT_NONE, ACT_LRELU, ACT_TANH = 0, 1, 2
PACK_FWD, PACK_DGRAD_S1, PACK_DGRAD_POLY, PACK_CONVT_POLY, PACK_DGRAD_2D, PACK_GCONV_FWD, PACK_GCONV_BWD, PACK_GMFMA_FWD = 0, 1, 2, 3, 4, 5, 6, 7
IO_X_BF16, IO_OUT_BF16, IO_MASK_BF16, IO_RES_BF16 = 1, 2, 4, 8
CK = 16
LOSS_L1, LOSS_L1_L1LOG, LOSS_MSE_TARGET, LOSS_MSE_REL, LOSS_L1_ENC = 0, 1, 2, 3, 4
MAX_LOSS_JOBS = 48

ABI_VERSION = 11           # RTG_ABI_VERSION of include/rtg.h these struct layouts / prototypes were written for
_P = C.c_void_p
_I, _F, _D, _LL, _ULL = C.c_int, C.c_float, C.c_double, C.c_longlong, C.c_ulonglong

# name -> (restype, argtypes); must list every symbol include/rtg.h declares (checked by tests/test_abi.py)
PROTOTYPES = {
    'rtg_conv1d': (_I, [C.POINTER(Conv1dDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'rtg_conv1d_group': (_I, [_I, C.POINTER(Conv1dDesc), C.POINTER(ConvPtrs), _P]),
    'rtg_conv1d_variant': (_I, [C.POINTER(Conv1dDesc)]),
    'rtg_conv1d_tile_candidates': (_I, [C.POINTER(Conv1dDesc), C.POINTER(C.c_int), _I]),
    'rtg_packed_size': (_LL, [_I, _I, _I, _I, _I]),
    'rtg_packed_size_frag16': (_LL, [_I, _I, _I]),
    'rtg_packed_size_frag16_bf16': (_LL, [_I, _I, _I]),
    'rtg_packed_size_tapmajor': (_LL, [_I, _I, _I, _I, _I]),
    'rtg_packed_size_bf16': (_LL, [_I, _I, _I, _I, _I]),
    'rtg_tapmajor_pays': (_I, [_I, _I, _I]),
    'rtg_conv1d_wgrad': (_I, [C.POINTER(WgradDesc), _P, _P, _P, _P, _P, _P]),
    'rtg_conv1d_wgrad_group': (_I, [_I, C.POINTER(WgradDesc), C.POINTER(WgradPtrs), _P]),
    'rtg_gconv_ok': (_I, [C.POINTER(GconvDesc)]),
    'rtg_gconv_workspace': (C.c_longlong, [C.POINTER(GconvDesc)]),
    'rtg_gconv_prepare': (_I, [C.POINTER(GconvDesc), _P, _P, _P, _P]),
    'rtg_gconv_forward': (_I, [C.POINTER(GconvDesc), _P, _P, _P, _P, _P]),
    'rtg_gconv_prepare_bwd': (_I, [C.POINTER(GconvDesc), _P, _P, _P, _P]),
    'rtg_gconv_backward_data': (_I, [C.POINTER(GconvDesc), _P, _P, _P, _P, _P, _P]),
    'rtg_gmfma_ok': (_I, [C.POINTER(GconvDesc)]),
    'rtg_gmfma_workspace': (C.c_longlong, [C.POINTER(GconvDesc)]),
    'rtg_gmfma_forward': (_I, [C.POINTER(GconvDesc), _P, _P, _P, _P, _P]),
    'rtg_wgrad_splits': (_I, [C.POINTER(WgradDesc)]),
    'rtg_wgrad_shape_candidates': (_I, [C.POINTER(WgradDesc), C.POINTER(C.c_int), _I]),
    'rtg_weightnorm_scales': (_I, [_P, _I, _I, _P, _P, _P]),
    'rtg_weights_pack': (_I, [_P, _I, _LL, _I, _P, _P, _P, _P]),
    'rtg_pack_job_blocks': (_I, [_P]),
    'rtg_pack_job_lds': (_I, [_P]),
    'rtg_weightnorm_backward': (_I, [_P, _I, _I, _I, _P, _P, _P, _P, _P]),
    'rtg_stft_forward': (_I, [C.POINTER(StftDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'rtg_stft_backward': (_I, [C.POINTER(StftDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'rtg_stft_forward_multi': (_I, [_I, C.POINTER(StftFwdJob), _P]),
    'rtg_stft_backward_multi': (_I, [_I, C.POINTER(StftBwdJob), _P, _I, _P]),
    'rtg_noise_lrelu_fwd': (_I, [_P, _P, _P, _P, _LL, _F, _ULL, _P, _P]),
    'rtg_noise_lrelu_bwd': (_I, [_P, _P, _P, _P, _P, _P, _I, _LL, _F, _ULL, _P, _P]),
    'rtg_noise_lrelu_bwd_acc': (_I, [_P, _P, _P, _P, _P, _P, _I, _LL, _F, _ULL, _P, _P, _P]),
    'rtg_channel_sum': (_I, [_P, _P, _I, _I, _I, _P, _P]),
    'rtg_axpby': (_I, [_P, _P, _P, _LL, _F, _F, _I, _P]),
    'rtg_lrelu_bwd': (_I, [_P, _P, _P, _LL, _F, _P]),
    'rtg_bf16_encode': (_I, [_P, _P, _LL, _F, _P]),
    'rtg_bf16_decode': (_I, [_P, _P, _LL, _F, _P]),
    'rtg_avgpool4s2_fwd': (_I, [_P, _P, _I, _I, _P]),
    'rtg_avgpool4s2_bwd': (_I, [_P, _P, _I, _I, _P]),
    'rtg_period_fold_fwd': (_I, [_P, _P, _I, _I, _I, _I, _P]),
    'rtg_period_fold_bwd': (_I, [_P, _P, _I, _I, _I, _I, _P]),
    'rtg_loss_fwd': (_I, [_I, _P, _I, _P, _P, _P]),
    'rtg_loss_bwd': (_I, [_I, _P, _I, _P, _P]),
    'rtg_dyn_loss_fwd': (_I, [_P, _P, _I, _I, _I, _F, _P, _P, _P]),
    'rtg_dyn_loss_bwd': (_I, [_P, _P, _I, _I, _I, _F, _P, _P, _P]),
    'rtg_env_loss_fwd': (_I, [_P, _P, _I, _I, _I, _F, _P, _P, _P]),
    'rtg_env_loss_bwd': (_I, [_P, _P, _I, _I, _I, _F, _P, _P, _P]),
    'rtg_strip_mirror_fwd': (_I, [_P, _I, _I, _F, _P, _P, _P, _P]),
    'rtg_strip_mirror_bwd': (_I, [_P, _I, _I, _F, _P, _P, _P, _P]),
    'rtg_adamw': (_I, [_P, _P, _P, _P, _LL, _P, _P, _D, _D, _D, _D, _D, _F, _P]),
    'rtg_resstack_ok': (_I, [C.POINTER(ResStackDesc)]),
    'rtg_resstack_forward': (_I, [C.POINTER(ResStackDesc), _P, C.POINTER(PtrArray6), C.POINTER(PtrArray6), C.POINTER(PtrArray6), _P]),
    'rtg_resstack_backward': (_I, [C.POINTER(ResStackDesc), _P, _P, C.POINTER(PtrArray6), C.POINTER(PtrArray6), C.POINTER(PtrArray6), _P]),
    'rtg_stream_create': (_I, [_I, C.POINTER(C.c_void_p)]),
    'rtg_stream_destroy': (_I, [_P]),
    'rtg_stream_end_capture': (_I, [_P]),
    'rtg_scalar_wsum': (_I, [C.POINTER(ScalarTerms), _P, _P]),
    'rtg_scalar_fanout': (_I, [C.POINTER(ScalarTerms), _P, _P, _P]),
    'rtg_abi_version': (_I, []),
    'rtg_build_info': (C.c_char_p, []),
}


class _Lib:
    """Lazy handle: importing the package works without the .so (CPU-only host logic and tests), calling any
    kernel without it raises."""

    def __init__(self):
        self._dll = None

    def load(self):
        if self._dll is None:
            if not os.path.exists(LIB_PATH):
                raise RtgError(f'{LIB_PATH} is missing: run `python transtacos-retunegan_amd/build.py` '
                               '(there is no non-HIP fallback for the RetuneGAN hot path)')
            dll = C.CDLL(LIB_PATH)
            for name, (res, args) in PROTOTYPES.items():
                try:
                    fn = getattr(dll, name)      # AttributeError if the symbol is not exported
                except AttributeError:
                    raise RtgError(f'{LIB_PATH} does not export {name}: stale build — run '
                                   '`python transtacos-retunegan_amd/build.py`') from None
                fn.restype, fn.argtypes = res, args
            if hasattr(dll, 'rtg_abi_version') and dll.rtg_abi_version() != ABI_VERSION:
                raise RtgError(f'{LIB_PATH} has ABI version {dll.rtg_abi_version()}, these bindings are for '
                               f'{ABI_VERSION}: stale build — run `python transtacos-retunegan_amd/build.py`')
            if not os.environ.get('RTG_DEV_LIB') and hasattr(dll, 'rtg_build_info') and b'ABLATION' in dll.rtg_build_info():
                raise RtgError(f'{LIB_PATH} was compiled with an ablation / diagnostic define (its results are wrong by '
                               'design): rebuild with `python transtacos-retunegan_amd/build.py --force`')
            self._dll = dll
        return self._dll

    def __getattr__(self, name):                 # first use of a symbol only: the function is cached on the instance
        fn = getattr(self.load(), name)
        if not name.startswith('_'):
            self.__dict__[name] = fn
        return fn


lib = _Lib()


def check(status, what=''):
    if status != 0:
        raise RtgError(f'librtg call {what} failed with status {status}')


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def assign_pack_blocks(jobs):
    """fill first_block / n_blocks of a list of PackJob (rtg_pack_job_blocks per job) -> (the launch's total_blocks, its
    lds_floats = the largest slab any job stages)"""
    total, lds = 0, 0
    for j in jobs:
        lds = max(lds, lib.rtg_pack_job_lds(C.byref(j)))
        n = lib.rtg_pack_job_blocks(C.byref(j))
        if n < 1:
            raise RtgError(f'rtg_pack_job_blocks: invalid pack job (mode {j.mode}, {j.Mg} x {j.Cg} x {j.K})')
        j.first_block, j.n_blocks = total, n
        total += n
    return total, lds


def current_stream_ptr():
    """raw handle of torch's current HIP stream on the current device as a ctypes pointer (torch.cuda.current_stream()
    builds a Stream object per call: ~5 us, 900 times per train step)"""
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


_OWN_STREAMS = []


def new_stream(priority=0, device=None):
    """A HIP stream of the library's own (rtg_stream_create), wrapped for torch: NOT one of the 32 pooled streams
    torch.cuda.Stream() hands out round robin, which ProcessGroupNCCL also draws its collective stream from — a pooled
    stream used for HIP-graph capture can be the RCCL stream itself, and the process group's watchdog then dies polling an
    event of a capturing stream (hipErrorCapturedEvent, round 4).  Lives as long as the process."""
    dev = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    with torch.cuda.device(dev):
        h = C.c_void_p()
        check(lib.rtg_stream_create(int(priority), C.byref(h)), 'stream_create')
    s = torch.cuda.ExternalStream(h.value, device=dev)
    _OWN_STREAMS.append(s)
    return s
