import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, 'transtacos-retunegan_amd')); sys.path.insert(0, REPO)
import torch, bench
import hparam as hp
from train import Trainer
from rtg import tune
torch.manual_seed(1)
tr = Trainer(use_mpd=True, use_mtd=False, d_train_times=1, dev='cuda')
bank = tr.generator.bank()
for ly in bank.layers:
    if ly.fwd16 or ly.bwd16:
        print('G layer with fragment image:', ly.name, ly.fwd16, ly.bwd16, ly.fwd_op, ly.bwd_op)
data = bench.synthetic_batch(32, 8192, 1, 'cuda')
tr.train_step(*data)
torch.cuda.synchronize()
print('picks > 9000:', sum(1 for v in tune._conv.values() if v > 9000), 'of', len(tune._conv))
from rtg.lib import Conv1dDesc
import ctypes as C
for k, v in tune._conv.items():
    d = Conv1dDesc.from_buffer_copy(k)
    if d.wp16 and d.L_in <= 64 and d.stride == 1 and d.B * d.L_in <= 8192 and d.Cg >= 128:
        print('  desc B', d.B, 'C1', d.C1, 'C2', d.C2, 'Mg', d.Mg, 'L', d.L_in, 'K', d.K, 'dil', d.dil, 'split', d.out_split, 'pre', d.pre_mode, '-> pick', v)
