"""dev: the loss terms of one bf16 train step (G + MSD + MPD), product vs the bf16-rounding oracle"""
import os, sys
import torch
REPO = os.path.join(os.path.dirname(__file__), '..', '..')
sys.path.insert(0, os.path.join(REPO, 'transtacos-retunegan_amd')); sys.path.insert(0, os.path.join(REPO, 'oracle'))
import hparam as hp
hp.compute_dtype = 'bf16'
hp.bf16_maps = os.environ.get('MAPS', '1') == '1'
import rtg_oracle as oracle
from train import Trainer
torch.manual_seed(3)
tr = Trainer(use_mpd=True, use_mtd=False, d_train_times=2, dev='cuda:0')
nets = (oracle.Generator(), oracle.MSD(), oracle.MPD())
for m in (tr.generator, *tr.discs):
    oracle.det_fill(m)
for m in nets:
    oracle.det_fill(m)
for prod, om in zip((tr.generator, tr.msd, tr.mpd), nets):
    flags = {ly.name: (ly.fwd_bf, ly.maps_bf) for ly in prod.bank().layers}
    for name, mod in om.named_modules():
        if name in flags:
            mod.bf16 = bool(flags[name][0]); mod.store_bf16 = bool(flags[name][1])
x, y_tmpl, y = oracle.golden_inputs()
opts = oracle.make_optimizers(nets[0], list(nets[1:]))
dl, gl = tr.train_step(x.cuda(), y_tmpl.cuda(), y.cuda())
odl, ogl = oracle.train_step(nets[0], *opts, x, y_tmpl, y, nets[1], nets[2], None, 2)
print('D product', {k: round(v.item(), 5) for k, v in dl.items()})
print('D oracle ', {k: round(v.item(), 5) for k, v in odl.items()})
print('G product', {k: round(v.item(), 5) for k, v in gl.items() if v is not None})
print('G oracle ', {k: round(v.item(), 5) if torch.is_tensor(v) else v for k, v in ogl.items()})
