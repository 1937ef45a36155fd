"""Constants and helpers the reference exposes from retunegan/utils.py (re-exported by `from models import *`).
Plotting helpers are not part of the hot path and are left out."""
import glob
import os

import torch

import hparam as hp

LRELU_SLOPE = 0.15            # retunegan/utils.py:11
PI = 3.14159265358979         # retunegan/utils.py:12


def init_weights(m, mean=0.0, std=0.01):
    """utils.py:26-29.  On a weight-normed conv the reference's call only rewrites the derived `.weight`, which the
    next forward recomputes: a no-op on parameters that still consumes RNG.  Our layers mirror that (burn_init_rng)."""
    if hasattr(m, 'burn_init_rng'):
        m.burn_init_rng()


def get_padding(kernel_size, dilation=1):
    return (kernel_size * dilation - dilation) // 2


def get_same_padding(kernel_size, dilation=1):
    return dilation * (kernel_size // 2)


def get_param_cnt(model):
    return sum(p.numel() for p in model.parameters())


def load_checkpoint(fp, device):
    assert os.path.isfile(fp)
    print(f"Loading '{fp}'")
    ckpt = torch.load(fp, map_location=device)
    print("Complete.")
    return ckpt


def save_checkpoint(fp, obj):
    print(f"Saving checkpoint to {fp}")
    torch.save(obj, fp)
    print("Complete.")


def scan_checkpoint(dp, prefix):
    cp_list = glob.glob(os.path.join(dp, prefix + '*'))
    return len(cp_list) and sorted(cp_list)[-1] or None
