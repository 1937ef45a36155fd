#!/usr/bin/env python3
"""Per-job timing of rtg_weights_pack (dev tool): every pack job of a model's bank launched alone."""
import ctypes as C
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, 'transtacos-retunegan_amd'))
import torch  # noqa: E402
from rtg import lib as L  # noqa: E402
from rtg.lib import lib  # noqa: E402
from rtg import bank as B  # noqa: E402
import models  # noqa: E402

torch.manual_seed(0)
which = sys.argv[1] if len(sys.argv) > 1 else 'mpd'
m = {'g': models.Generator_RefineGAN_small, 'msd': models.MultiScaleDiscriminator, 'mpd': models.MultiPeriodDiscriminator,
     'mtd': models.MultiStftDiscriminator}[which]().cuda()
bank = m.bank()
bank._run_prep()
torch.cuda.synchronize()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
tab = bank.pack_table
sz = C.sizeof(L.PackJob)
jobs = (L.PackJob * bank.n_pack).from_buffer_copy(bytes(tab.cpu().numpy().tobytes()))


def run(ptr, n, mx, lds, reps=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(2):
        lib.rtg_weights_pack(ptr, n, mx, lds, C.c_void_p(bank.flat.data_ptr()), C.c_void_p(bank.scales.data_ptr()),
                             C.c_void_p(bank.packed.data_ptr()), st)
    e0.record()
    for _ in range(reps):
        lib.rtg_weights_pack(ptr, n, mx, lds, C.c_void_p(bank.flat.data_ptr()), C.c_void_p(bank.scales.data_ptr()),
                             C.c_void_p(bank.packed.data_ptr()), st)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


tot = run(C.c_void_p(tab.data_ptr()), bank.n_pack, bank.pack_blocks, bank.pack_lds)
print(f'{which}: all {bank.n_pack} jobs in one launch: {tot:.1f} us, packed {bank.packed.numel() * 4 / 1e6:.1f} MB, '
      f'params {bank.n_params * 4 / 1e6:.1f} MB -> {(bank.packed.numel() + bank.n_params) * 4 / tot / 1e3:.0f} GB/s')
s = 0.0
for i, j in enumerate(jobs):
    one = (L.PackJob * 1).from_buffer_copy(bytes(j))
    nblk, lds1 = L.assign_pack_blocks(one)
    one_d = torch.frombuffer(bytearray(bytes(one)), dtype=torch.uint8).cuda()
    us = run(C.c_void_p(one_d.data_ptr()), 1, nblk, lds1)
    s += us
    print(f'{i:3d} mode{j.mode} g{j.groups} Mg{j.Mg} Cg{j.Cg} K{j.K} srcK{j.src_K} S{j.S} tm{j.tile_m} KH{j.KH} tap{j.tap_major} '
          f'bf{j.bf16} f16_{j.frag16}  {j.dst_size * 4 / 1e6:7.2f} MB  {us:7.1f} us  {j.dst_size * 4 / us / 1e3:7.0f} GB/s')
print(f'sum of single launches {s:.1f} us')
