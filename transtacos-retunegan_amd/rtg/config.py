"""The environment switches of the package — all of them, in one place (round 5: the A/B knobs whose measurements are settled
became constants; DESIGN.md records the numbers).  `get(name)` is what the modules call; `non_default()` is what bench.py
prints into its record, so that a number measured under a switch says so."""
import os

# name -> (default, what it does)
KNOBS = {
    'RTG_TUNE': ('1', '0: no timing of block shapes in the first train step, the library\'s heuristics everywhere'),
    'RTG_STREAMS': ('1', '0: sub-networks run one after the other on one stream (profiling: a launch has the chip to itself)'),
    'RTG_GRAPH': ('1', 'bench.py: 0 = time the eager step instead of the HIP-graph replay'),
    'RTG_DP_CUT': ('update', 'exchange policy a Trainer starts with: update | disc (train.default_exchange); bench.py times both when unset'),
    'RTG_DP_FORCE': ('0', '1: the data-parallel machinery over a process group of ONE rank (tests, rehearsal on a one-GPU box)'),
    'RTG_CAPTURE_MODE': ('thread_local', 'HIP-graph capture error mode (global: what torch defaults to; dies under a live RCCL watchdog)'),
    'RTG_BENCH_REHEARSE': ('0', 'bench.py: 1 = all ranks on cuda:0, gradients over gloo (rehearses --gpus N on a one-GPU box)'),
    'RTG_BENCH_VERBOSE': ('0', 'bench.py: 1 = progress lines on stderr also for a single rank'),
    'RTG_TEST_FAIL_CAPTURE': ('0', 'test hook: an illegal synchronous copy inside the capture (bench.py\'s eager fallback)'),
    'RTG_DEV_LIB': ('', 'path of a development build of librtg.so to load instead of the in-tree one (ablation builds)'),
    'RTG_EXTRA_FLAGS': ('', 'build.py: extra hipcc flags (ablation defines are refused there)'),
    # a switch of the HIP RUNTIME, not of this package: kernel arguments in device memory instead of host-coherent memory — the
    # command processor fetches them faster, which a step of 670 launches (115 of them one dependent chain) feels: round 6,
    # same-box A/B of the replayed config-2 step, four pairs: 26.35 / 26.43 / 26.55 / 26.42 ms with it, 26.72 / 26.65 / 26.85 /
    # 27.20 without (profiles/r06_ab_dev_kernarg.txt).  Read when the runtime initialises: the package sets it on import
    # (apply_runtime_env) unless the environment already says otherwise; a process that touched the GPU before is not affected.
    'HIP_FORCE_DEV_KERNARG': ('1', 'HIP runtime: 1 = kernel arguments in device memory (set by the package on import if unset; 0 restores the runtime default)'),
}


def apply_runtime_env():
    """defaults of HIP-runtime switches this package wants, for variables the caller has not set (before the runtime initialises)"""
    os.environ.setdefault('HIP_FORCE_DEV_KERNARG', KNOBS['HIP_FORCE_DEV_KERNARG'][0])


apply_runtime_env()


def get(name):
    """the switch's value (its default when unset); unknown names are a programming error"""
    return os.environ.get(name, KNOBS[name][0])


def non_default():
    """{name: value} of the switches set to something else than their default"""
    return {k: os.environ[k] for k, (dflt, _) in KNOBS.items() if k in os.environ and os.environ[k] != dflt}
