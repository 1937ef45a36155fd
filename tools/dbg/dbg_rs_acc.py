import os, sys, numpy as np, torch, importlib.util
REPO=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO,'transtacos-retunegan_amd')); sys.path.insert(0, os.path.join(REPO,'tests'))
spec=importlib.util.spec_from_file_location('rtg_oracle',os.path.join(REPO,'oracle','rtg_oracle.py')); O=importlib.util.module_from_spec(spec); spec.loader.exec_module(O)
from test_resstack_gpu import _net
from rtg import ops
import torch.nn.functional as F
for seed in range(3):
    torch.manual_seed(seed)
    net=_net(128); ref64=O.ResidualStack(128).double(); ref32=O.ResidualStack(128)
    sd={k[len('stack.'):]:v for k,v in net.state_dict().items()}
    ref64.load_state_dict({k:v.double() for k,v in sd.items()}); ref32.load_state_dict(sd)
    net.to('cuda')
    x=torch.randn(8,128,32)
    y64=F.leaky_relu(ref64(x.double()),0.15); y32=F.leaky_relu(ref32(x),0.15)
    res={}
    for fused in (True,False):
        ops.RESSTACK=fused
        with torch.no_grad(): y=net(x.cuda(),0.15).cpu()
        res[fused]=((y.double()-y64).norm()/y64.norm()).item()
    ops.RESSTACK=True
    print(seed,'fused',res[True],'unfused',res[False],'cpu fp32',((y32.double()-y64).norm()/y64.norm()).item())
