import sys, os, numpy as np, torch, importlib.util
REPO=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO,'transtacos-retunegan_amd'))
spec=importlib.util.spec_from_file_location('rtg_oracle',os.path.join(REPO,'oracle','rtg_oracle.py')); O=importlib.util.module_from_spec(spec); spec.loader.exec_module(O)
gold=dict(np.load(os.path.join(REPO,'tests/golden/retunegan_b2_t8192.npz')))
from models import MultiStftDiscriminator, multi_stft_loss
DEV='cuda'
torch.manual_seed(1)
mtd=MultiStftDiscriminator(); O.det_fill(mtd); mtd.to(DEV).train()
_,_,y=O.golden_inputs(); yd=torch.from_numpy(gold['y_hat'])
oS,oSg=O.multi_stft_loss(y,yd,ret_specs=True)
def cmp(tag, lr, lg):
    for i,(r,g) in enumerate(zip(lr,lg)):
        er=np.abs(r.detach().cpu().numpy()-gold[f'mtd_logit_r{i}']).max(); eg=np.abs(g.detach().cpu().numpy()-gold[f'mtd_logit_g{i}']).max()
        print(tag,i,'max abs err r',er,'g',eg,'scale',np.abs(gold[f'mtd_logit_r{i}']).max())
with torch.no_grad():
    lr,lg,fr,fg=mtd([s.to(DEV) for s in oS],[s.to(DEV) for s in oSg])
cmp('nograd cpu-spectra',lr,lg)
lr,lg,fr,fg=mtd([s.to(DEV) for s in oS],[s.to(DEV) for s in oSg])
cmp('grad   cpu-spectra',lr,lg)
S,Sg=multi_stft_loss(y.to(DEV),yd.to(DEV),ret_specs=True)
def canon(spec, ospec):
    ph, oph = spec[:, 1], ospec[:, 1]
    cut = (ph.abs() > 1 - 1e-3) & (oph.abs() > 1 - 1e-3) & (torch.sign(ph) != torch.sign(oph))
    out = spec.clone(); out[:, 1] = torch.where(cut, oph, ph); return out
fx=[canon(a,b.to(DEV)) for a,b in zip(S+Sg,oS+oSg)]
for a,b in zip(fx,oS+oSg):
    print('after canon max diff ph',(a[:,1].cpu()-b[:,1]).abs().max().item(),'logS',(a[:,0].cpu()-b[:,0]).abs().max().item())
with torch.no_grad():
    lr,lg,fr,fg=mtd(fx[:3],fx[3:])
cmp('nograd canon',lr,lg)
lr,lg,fr,fg=mtd(fx[:3],fx[3:])
cmp('grad canon',lr,lg)
olr,olg,_,_=O.MTD.__call__ if False else (None,None,None,None)
omtd=O.MTD(); O.det_fill(omtd); omtd.train()
with torch.no_grad():
    olr,olg,_,_=omtd([f.cpu() for f in fx[:3]],[f.cpu() for f in fx[3:]])
cmp('oracle on canon',olr,olg)
