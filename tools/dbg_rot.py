import sys; sys.path.insert(0, ".")
import torch
from tests import golden
from oracle import ops
from scratchpad_amd import _native
g = golden.load("rotary"); dtype=torch.float16
hs=int(g["c0_head_size"]); mp=int(g["c0_max_pos"])
cache = ops.rope_cos_sin_cache(mp, float(g["c0_base"]), hs, None, dtype)
pos = torch.from_numpy(g["c0_positions"]); q = torch.from_numpy(g["c0_q"]).to(dtype); k=torch.from_numpy(g["c0_k"]).to(dtype)
qr, kr = ops.rotary_embedding(pos, q, k, hs, cache, True)
qg, kg = q.cuda(), k.cuda()
_native.rotary_embedding(pos.cuda(), qg, kg, hs, cache.cuda(), True)
d = (qg.cpu().float()-qr.float()).abs()
print("max", d.max().item(), "n", (d>0).sum().item(), "of", d.numel())
for a,b in (d>0).nonzero()[:10].tolist():
    h=b//hs; e=b%hs; j=e%(hs//2)
    x1=q[a,h*hs+j].item(); x2=q[a,h*hs+hs//2+j].item(); c=cache[pos[a],j].item(); s=cache[pos[a],hs//2+j].item()
    print(a,b,"hip",qg[a,b].item(),"ref",qr[a,b].item(),"x1",x1,"x2",x2,"c",c,"s",s, "pos", pos[a].item())
