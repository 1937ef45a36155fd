import sys, os, copy
sys.path.insert(0, 'transtacos-retunegan_amd'); sys.path.insert(0, 'oracle')
import torch, numpy as np
import torch.nn.functional as F
import rtg_oracle as O
from models import Generator_RefineGAN_small
from models.generator import conv, _Mean3, ACT_LRELU, ACT_TANH, LRELU_SLOPE
torch.manual_seed(7)
g = Generator_RefineGAN_small(); og = O.Generator()
O.det_fill(g); O.det_fill(og); g.to('cuda')
og = og.double()
x, y_tmpl, _ = O.golden_inputs(seed=3)
dy = torch.randn(2, 1, 8192, generator=torch.Generator().manual_seed(5))

def keep(lst, name, t):
    t.retain_grad(); lst.append((name, t)); return t

# product forward with retained intermediates
P = []
self = g
tok = self.token()
xg, yg = x.cuda(), y_tmpl.cuda()
o = []
y = keep(P, 'pre', conv(tok, self.conv_pre, yg, act=ACT_LRELU, act_slope=LRELU_SLOPE))
for i in range(3):
    o.append(y)
    y = keep(P, f'down{i}', conv(tok, self.downs[i], y))
    y = keep(P, f'rs{i}', self.resblock[i].run(tok, y, final_act_slope=LRELU_SLOPE))
z = keep(P, 'fuse', conv(tok, self.conv_fuse, xg, y))
for i in range(3):
    z = keep(P, f'up{i}', conv(tok, self.ups[i], z, pre_slope=LRELU_SLOPE))
    z = keep(P, f'merge{i}', conv(tok, self.merge[i], z, o[2 - i]))
    z = keep(P, f'noiseA{i}', self.noise(z))
    brs = [keep(P, f'br{i}{j}', self.resblocks[i * 3 + j].run(tok, z)) for j in range(3)]
    z = keep(P, f'mean{i}', _Mean3.apply(*brs))
    z = keep(P, f'noiseB{i}', self.noise(z))
out = conv(tok, self.conv_post, z, pre_slope=LRELU_SLOPE, act=ACT_TANH)
out.backward(dy.cuda())

# oracle f64 with the same intermediates
R = []
self = og
x6, y6 = x.double(), y_tmpl.double()
o = []
y = keep(R, 'pre', F.leaky_relu(self.conv_pre(y6), 0.15))
for i in range(3):
    o.append(y)
    y = keep(R, f'down{i}', self.downs[i](y))
    y = keep(R, f'rs{i}', F.leaky_relu(self.resblock[i](y), 0.15))
z = keep(R, 'fuse', self.conv_fuse(torch.cat([x6, y], 1)))
for i in range(3):
    z = keep(R, f'up{i}', self.ups[i](F.leaky_relu(z, 0.15)))
    z = keep(R, f'merge{i}', self.merge[i](torch.cat([z, o[2 - i]], 1)))
    z = keep(R, f'noiseA{i}', F.leaky_relu(z, 0.15))
    brs = [keep(R, f'br{i}{j}', self.resblocks[i * 3 + j](z)) for j in range(3)]
    z = keep(R, f'mean{i}', sum(brs) / 3)
    z = keep(R, f'noiseB{i}', F.leaky_relu(z, 0.15))
out6 = torch.tanh(self.conv_post(F.leaky_relu(z, 0.15)))
out6.backward(dy.double())
for (n, a), (m, b) in zip(P, R):
    assert n == m
    fe = (a.detach().cpu().double() - b.detach()).abs().max().item() / (b.abs().max().item() + 1e-30)
    ge = (a.grad.cpu().double() - b.grad).abs().max().item() / (b.grad.abs().max().item() + 1e-30)
    d = (a.grad.cpu().double() - b.grad).abs()
    idx = np.unravel_index(d.argmax().item(), d.shape)
    print(f'{n:10s} fwd {fe:.2e} grad {ge:.2e}  worst at {idx} of {tuple(d.shape)}')
PD, RD = dict(P), dict(R)
for name, idx in (('noiseB1', (0, 28, 756)), ('mean1', (0, 28, 756)), ('noiseB0', (0, 12, 61)), ('mean0', (0, 12, 61))):
    print(name, idx, 'gpu fwd', PD[name][idx].item(), 'f64 fwd', RD[name][idx].item(),
          'gpu grad', PD[name].grad[idx].item(), 'f64 grad', RD[name].grad[idx].item())
for (n, a), (m, b) in zip(P, R):
    d = (a.grad.cpu().double() - b.grad)
    print(f'{n:10s} L2-rel grad err {d.norm().item() / b.grad.norm().item():.2e}')
