import sys, os, numpy as np, torch, importlib.util
REPO=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO,'transtacos-retunegan_amd'))
spec=importlib.util.spec_from_file_location('rtg_oracle',os.path.join(REPO,'oracle','rtg_oracle.py')); O=importlib.util.module_from_spec(spec); spec.loader.exec_module(O)
gold=dict(np.load(os.path.join(REPO,'tests/golden/retunegan_b2_t8192.npz')))
from models import MultiStftDiscriminator, multi_stft_loss, MultiScaleDiscriminator, discriminator_loss
DEV='cuda'
_,_,y=O.golden_inputs(); yd=torch.from_numpy(gold['y_hat'])
S,Sg=multi_stft_loss(y.to(DEV),yd.to(DEV),ret_specs=True)
oS,oSg=O.multi_stft_loss(y,yd,ret_specs=True)
for i,(a,b) in enumerate(zip(S+Sg,oS+oSg)):
    a=a.cpu()
    d=(a[:,1]-b[:,1]).abs()
    dm=torch.minimum(d,2-d)
    print(i,'shape',tuple(a.shape),'frac |dphase|>0.1 raw',(d>0.1).float().mean().item(),'mod2',(dm>0.1).float().mean().item(),
          'max mod2',dm.max().item(),'logS diff max',(a[:,0]-b[:,0]).abs().max().item(), 'min logS',b[:,0].min().item())
    # where are raw diffs: per frame
    fr=(d>0.1).float().sum(dim=(0,1))
    nz=torch.nonzero(fr).flatten().tolist()
    print('   frames with wraps:',nz[:10],'...',nz[-5:], 'counts', fr[nz[:3]].tolist())
    w=(d>0.1)
    print('   |ph| at wraps gpu min',a[:,1][w].abs().min().item() if w.any() else None,'cpu min',b[:,1][w].abs().min().item() if w.any() else None)
# NaN propagation
msd=MultiScaleDiscriminator().to(DEV)
yb=y.clone().to(DEV); yb[0,0,100]=float('nan')
with torch.no_grad():
    lr,lg,fr,fg=msd(yb,yd.to(DEV))
for i,f in enumerate(fr[0]): print('fmap',i,'nan count',torch.isnan(f).sum().item())
print('logit nan',[torch.isnan(l).sum().item() for l in lr], 'loss',discriminator_loss(lr,lg).item())
from models import MultiPeriodDiscriminator
mpd=MultiPeriodDiscriminator().to(DEV)
with torch.no_grad():
    lr,lg,fr,fg=mpd(yb,yd.to(DEV))
for i,f in enumerate(fr[0]): print('mpd fmap',i,'nan count',torch.isnan(f).sum().item())
print('mpd loss',discriminator_loss(lr,lg).item())
