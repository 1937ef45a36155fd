"""Per-problem choice of the MFMA kernels' block shape by timing.

The conv / wgrad kernels come in several block shapes (wave tile x wave grid).  Which one is fastest for a layer depends
on how its rows, clips and output channels quantise onto 256 CUs, and a static score misses by 10-20 % on the
discriminator layers.  So the first train step times the candidates the library lists for each distinct descriptor
(`rtg_conv1d_tile_candidates`, `rtg_wgrad_shape_candidates`) on the live tensors and keeps the fastest; later steps look
the choice up.  The block shape never changes a result bit (the summation order of every output element is the same for
all shapes), so tuning is invisible to the parity tests.  RTG_TUNE=0 keeps the library's heuristic everywhere.

While `ACTIVE` is set the sub-networks run serially on one stream (models.layers.fork_join) so that the timings are not
disturbed by neighbours; `train.Trainer` sets it for its first step and again after a step that met an untuned problem.
"""
import ctypes as C
import os

import torch

from . import config
from .lib import lib

ENABLED = config.get('RTG_TUNE') != '0'
ACTIVE = False
MISSED = False
REPS = 3
MAX_CANDS = 48            # candidate block shapes asked of rtg_conv1d_tile_candidates (general + resconv + dconv codes)
_conv, _wgrad, _group, _wgroup, _alt = {}, {}, {}, {}, {}


def _time(launch):
    """average duration of `launch` (ms); None if it fails.  Short launches are repeated more often: three repetitions of
    a 15 us kernel measure the event bracket as much as the kernel, and a wrong pick among candidates 2-3 % apart costs
    the small layers of UNet-G more than the extra repetitions cost the tuning step."""
    st = launch()                                   # warm-up: code object load, caches
    if st:
        return None
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        launch()
    e1.record()
    e1.synchronize()
    t = e0.elapsed_time(e1) / REPS
    if t < 0.1:                                     # < 100 us per launch: measure again over ~1 ms
        n = min(40, max(REPS, int(1.0 / max(t, 0.005))))
        e0.record()
        for _ in range(n):
            launch()
        e1.record()
        e1.synchronize()
        t = e0.elapsed_time(e1) / n
    return t


def _miss():
    global MISSED
    MISSED = MISSED or ENABLED
    return 0


def conv_cfg(d, launch):
    """tile_cfg for descriptor `d`; `launch()` runs rtg_conv1d with `d` as it stands and returns its status."""
    d.tile_cfg = 0
    key = bytes(d)
    cfg = _conv.get(key)
    if cfg is not None:
        return cfg
    if not (ENABLED and ACTIVE):
        return _miss()
    cands = (C.c_int * MAX_CANDS)()
    n = lib.rtg_conv1d_tile_candidates(C.byref(d), cands, MAX_CANDS)
    best, best_t = 0, None
    for c in cands[:max(n, 0)]:
        d.tile_cfg = c
        t = _time(launch)
        if t is not None and (best_t is None or t < best_t):
            best, best_t = c, t
    d.tile_cfg = 0
    _conv[key] = best
    return best


def wgrad_cfg(wd, run):
    """shape_cfg for wgrad descriptor `wd`; `run(part)` launches rtg_conv1d_wgrad with `wd` into the scratch `part`."""
    wd.shape_cfg, wd.splits, wd.part_stride = 0, 1, 0
    key = bytes(wd)
    cfg = _wgrad.get(key)
    if cfg is not None:
        return cfg
    if not (ENABLED and ACTIVE):
        return _miss()
    cands = (C.c_int * 12)()
    n = lib.rtg_wgrad_shape_candidates(C.byref(wd), cands, 12)
    need = wd.groups * wd.Mg * (wd.Cg * wd.K + 1)
    best, best_t = 0, None
    for c in cands[:max(n, 0)]:
        wd.shape_cfg = c
        splits = lib.rtg_wgrad_splits(C.byref(wd))
        if splits < 1:
            continue
        part = torch.empty(splits * need, device='cuda')
        wd.splits, wd.part_stride = splits, need
        t = _time(lambda: run(part))
        if t is not None and (best_t is None or t < best_t):
            best, best_t = c, t
    wd.shape_cfg, wd.splits, wd.part_stride = 0, 1, 0
    _wgrad[key] = best
    return best


def wgrad_cfg_any(wd, run):
    """(bf16, shape_cfg) for a bf16 weight-gradient descriptor whose layer may just as well run its fp32 shapes (the
    thin-group layers: exact-fit fp32 tiles against bf16 tiles that are mostly padding): every candidate of either
    arithmetic is timed.  Stored in the weight-gradient table as shape_cfg (+ 1000 for an fp32 pick)."""
    wd.bf16, wd.shape_cfg, wd.splits, wd.part_stride = 1, 0, 1, 0
    key = b'any' + bytes(wd)
    code = _wgrad.get(key)
    if code is None:
        if not (ENABLED and ACTIVE):
            _miss()
            return 1, 0
        need = wd.groups * wd.Mg * (wd.Cg * wd.K + 1)
        best, best_t = 0, None
        for bf in (1, 0):
            wd.bf16 = bf
            cands = (C.c_int * 12)()
            n = lib.rtg_wgrad_shape_candidates(C.byref(wd), cands, 12)
            for c in cands[:max(n, 0)]:
                wd.shape_cfg, wd.splits, wd.part_stride = c, 1, 0
                splits = lib.rtg_wgrad_splits(C.byref(wd))
                if splits < 1:
                    continue
                part = torch.empty(splits * need, device='cuda')
                wd.splits, wd.part_stride = splits, need
                t = _time(lambda: run(part))
                if t is not None and (best_t is None or t < best_t):
                    best, best_t = c + (0 if bf else 1000), t
        code = _wgrad[key] = best
    wd.bf16, wd.shape_cfg, wd.splits, wd.part_stride = (0 if code >= 1000 else 1), 0, 1, 0
    return wd.bf16, code % 1000


def wgrad_group_cfg(wds, run_group, run_singles):
    """How to run the weight gradients of n layers that may share one launch (rtg_conv1d_wgrad_group).  -> 0: one by
    one (each with its own tuned shape), else shape_cfg + 16 * d with the members' split counts divided by n (d = 1) or
    as for a lone launch (d = 0).  `run_group(code)` / `run_singles()` launch into scratch partials, return a status."""
    key = b''.join(bytes(w) for w in wds)
    cfg = _wgroup.get(key)
    if cfg is not None:
        return cfg
    if not (ENABLED and ACTIVE):
        return _miss()
    lists = []
    for w in wds:
        cands = (C.c_int * 12)()
        k = lib.rtg_wgrad_shape_candidates(C.byref(w), cands, 12)
        lists.append([c for c in cands[:max(k, 0)] if 1 <= c <= 6])
    common = [c for c in lists[0] if all(c in l for l in lists[1:])]
    best, best_t = 0, _time(run_singles)
    for c in common:
        for d in (0, 1):
            t = _time(lambda: run_group(c + 16 * d))
            if t is not None and (best_t is None or t < best_t):
                best, best_t = c + 16 * d, t
    _wgroup[key] = best
    return best


def alt_choice(key, launches):
    """index of the fastest of several complete ways to run one problem (`launches`: callables returning a status, each
    already tuned in itself); 0 (the first) until timed"""
    c = _alt.get(key)
    if c is not None:
        return c
    if not (ENABLED and ACTIVE):
        return _miss()
    best, best_t = 0, None
    for i, f in enumerate(launches):
        t = _time(f)
        if t is not None and (best_t is None or t < best_t):
            best, best_t = i, t
    _alt[key] = best
    return best


def group_cfg(darr, n, launch):
    """common tile_cfg for the n descriptors of a grouped launch (ctypes array `darr`); `launch()` runs rtg_conv1d_group
    on it as it stands.  0: the members have no block shape in common."""
    for i in range(n):
        darr[i].tile_cfg = 0
    key = b''.join(bytes(darr[i]) for i in range(n))
    cfg = _group.get(key)
    if cfg is not None:
        return cfg
    lists = []
    for i in range(n):
        cands = (C.c_int * MAX_CANDS)()
        k = lib.rtg_conv1d_tile_candidates(C.byref(darr[i]), cands, MAX_CANDS)
        lists.append([c for c in cands[:max(k, 0)] if 0 < c < 7000])     # 7001 / 7002: rtg_resconv, not a group member
    common = [c for c in lists[0] if all(c in l for l in lists[1:])]
    if not common:
        _group[key] = 0
        return 0
    if not (ENABLED and ACTIVE):
        _miss()
        return common[0]
    best, best_t = common[0], None
    for c in common:
        for i in range(n):
            darr[i].tile_cfg = c
        t = _time(launch)
        if t is not None and (best_t is None or t < best_t):
            best, best_t = c, t
    for i in range(n):
        darr[i].tile_cfg = 0
    _group[key] = best
    return best


_TABLES = ('_conv', '_wgrad', '_group', '_wgroup', '_alt')


def export_tables():
    """the pick tables as one picklable object (keys: descriptor bytes — shapes, no pointers)"""
    g = globals()
    return {n: dict(g[n]) for n in _TABLES}


def import_tables(tables):
    """take another rank's picks (train.DataParallel.sync_tuner); picks of problems only this process met are kept"""
    g = globals()
    for n in _TABLES:
        g[n].update(tables.get(n, {}))


def digest():
    """short hash of all picks (bench.py prints it per rank: equal digests = every rank runs the same kernels)"""
    import hashlib
    g = globals()
    picks = repr([sorted((k.hex() if isinstance(k, bytes) else repr(k), v) for k, v in g[n].items()) for n in _TABLES])
    return hashlib.sha256(picks.encode()).hexdigest()[:6]


def stats():
    return {'conv_problems': len(_conv), 'wgrad_problems': len(_wgrad), 'group_problems': len(_group), 'wgrad_group_problems': len(_wgroup)}
