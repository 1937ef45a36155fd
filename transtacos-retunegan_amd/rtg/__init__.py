"""rtg — binding layer between the reference-shaped Python API (hparam / models / audio / train in the parent
directory) and librtg.so (hand-written gfx950 kernels behind the C ABI of include/rtg.h).

    rtg.lib   ctypes prototypes of every entry point of include/rtg.h (lib.lib is the lazy library handle)
    rtg.bank  flat parameter / gradient / packed-weight buffers of a model (weight-norm prep and backward)
    rtg.ops   torch.autograd.Function wrappers that launch the kernels on the current stream
"""
from . import config  # noqa: F401  (first: sets the HIP-runtime defaults before anything can initialise the runtime)
from .lib import check, RtgError  # noqa: F401
