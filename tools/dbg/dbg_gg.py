import os, sys, numpy as np, torch, importlib.util
REPO=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO,'transtacos-retunegan_amd'))
spec=importlib.util.spec_from_file_location('rtg_oracle',os.path.join(REPO,'oracle','rtg_oracle.py')); O=importlib.util.module_from_spec(spec); spec.loader.exec_module(O)
from models import Generator_RefineGAN_small
from rtg import ops
torch.manual_seed(7)
g=Generator_RefineGAN_small(); og=O.Generator()
O.det_fill(g); O.det_fill(og); g.to('cuda').train(); og.train()
x,y_tmpl,_=O.golden_inputs(seed=3)
dy=torch.randn(2,1,8192,generator=torch.Generator().manual_seed(5))
og.zero_grad(); yo=og(x,y_tmpl); yo.backward(dy)
op=dict(og.named_parameters())
for fused in (True, False):
    ops.RESSTACK=fused
    g.zero_grad(); yg=g(x.cuda(),y_tmpl.cuda()); yg.backward(dy.cuda()); torch.cuda.synchronize()
    err={n:(p.grad.cpu()-op[n].grad).norm().item()/(op[n].grad.norm().item()+1e-20) for n,p in g.named_parameters() if n!='noise.w'}
    vals=sorted(err.values())
    print('fused' if fused else 'unfused','fwd maxdiff',(yg.detach().cpu()-yo.detach()).abs().max().item(),'median',vals[len(vals)//2],'max',vals[-1])
    top=sorted(err.items(), key=lambda kv:-kv[1])[:6]; print('   worst',[(n,round(e,5)) for n,e in top])
    dec=[e for n,e in err.items() if n.startswith(('resblocks','merge','ups','conv_post'))]; enc=[e for n,e in err.items() if n.startswith(('resblock.','downs','conv_pre','conv_fuse'))]
    print('   decoder median',sorted(dec)[len(dec)//2],'encoder median',sorted(enc)[len(enc)//2])
