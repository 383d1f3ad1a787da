import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch
from oracle import hang2020_np as O, prng
from conftest import rel_l2
from test_hip_parity import make, grads_of, dev
bands, classes, B, seed = 20, 7, 9, 5
x = prng.uniform01(seed + 1, 1, (B, bands, 11, 11)); y = prng.randint(seed + 1, 2, (B,), classes); w=np.ones(classes,np.float32)
res={}
for prec in ("fp32","bf16"):
    m,p = make("hang", bands, classes, seed, precision=prec); m.train()
    lg = m(torch.from_numpy(x).to(dev())); loss=torch.nn.functional.cross_entropy(lg, torch.from_numpy(y).to(dev())); loss.backward()
    res[prec]=(lg.detach().cpu().numpy(), grads_of(m))
def orc(q):
    O.set_conv_operand_quantizer(q)
    lg,c,_=O.hang2020_fwd(p,x,True,np.float64); l,dl=O.weighted_cross_entropy(lg,y,w); g=O.hang2020_bwd(p,c,dl,np.float64)
    O.set_conv_operand_quantizer(None); return lg,g
lg0,g0=orc(None); lgq,gq=orc(O.bf16_round)
print("logits: bf16hip-vs-exact %.2e  bf16hip-vs-emul %.2e  emul-vs-exact %.2e"%(rel_l2(res['bf16'][0],lg0),rel_l2(res['bf16'][0],lgq),rel_l2(lgq,lg0)))
for k in g0:
    if k.endswith('conv_layer.bias') or k=='alpha': continue
    a=res['bf16'][1][k]
    print("%-55s hip-vs-exact %.2e hip-vs-emul %.2e emul-vs-exact %.2e"%(k,rel_l2(a,g0[k]),rel_l2(a,gq[k]),rel_l2(gq[k],g0[k])))
