"""rtg — binding layer between the reference-shaped Python API (hparam / models / audio / train in the parent
directory) and librtg.so (hand-written gfx950 kernels behind the C ABI of include/rtg.h)."""
from .lib import lib, check, RtgError  # noqa: F401
