import sys, time, os, faulthandler; faulthandler.enable()
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/transtacos-retunegan_amd')
import torch, bench
import hparam as hp
from train import Trainer
torch.manual_seed(hp.randseed)
tr = Trainer(use_mpd=True, use_mtd=False, d_train_times=2, dev='cuda')
data = bench.synthetic_batch(32, 8192, 1, 'cuda')
for _ in range(3): tr.train_step(*data)
torch.cuda.synchronize()
t=time.time()
for _ in range(10): dl, gl = tr.train_step(*data)
torch.cuda.synchronize(); print('eager ms/step', (time.time()-t)*100, gl['gen_all'].item(), dl['disc_all'].item())
if len(sys.argv) > 1:
    if sys.argv[1] == 'del': del dl, gl
    elif sys.argv[1] == 'deldl': del dl
    elif sys.argv[1] == 'delgl': del gl
    elif sys.argv[1] == 'detach':
        dl = {k: v.detach() for k, v in dl.items()}; gl = {k: (v.detach() if v is not None else None) for k, v in gl.items()}
t=time.time(); dl, gl = tr.train_step_graphed(*data); torch.cuda.synchronize(); print('capture+first replay s', time.time()-t)
t=time.time()
for _ in range(10): dl, gl = tr.train_step_graphed(*data)
torch.cuda.synchronize(); print('graph ms/step', (time.time()-t)*100, gl['gen_all'].item(), dl['disc_all'].item())
