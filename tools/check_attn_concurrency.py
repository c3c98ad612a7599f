import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spann3r_amd import ops
from spann3r_amd.engine import _rope_tables
dev = "cuda"
torch.manual_seed(0)
B, P, C, heads = 1, 196, 768, 12
npad = 256
cos, sin = _rope_tables(64, 100.0, dev)
pos = torch.stack(torch.meshgrid(torch.arange(14), torch.arange(14), indexing="ij"), -1).reshape(-1, 2).to(torch.int32).to(dev)
def make():
    x = ops.PackedAct.from_dense(torch.randn(P, C, device=dev).to(torch.bfloat16))
    W = ops.PackedWeight((torch.randn(3 * C, C, device=dev) * 0.05).to(torch.bfloat16))
    b = torch.randn(3 * C, device=dev)
    qkp = torch.zeros(ops.packed_shape(npad, 2 * C, torch.bfloat16), device=dev, dtype=torch.bfloat16)
    vtp = torch.zeros(heads * npad * 64, device=dev, dtype=torch.bfloat16)
    ao = ops.PackedAct(P, C, torch.bfloat16, dev)
    return x, W, b, qkp, vtp, ao
def run(t):
    x, W, b, qkp, vtp, ao = t
    ops.proj_rope_vt(x, W, b, qkp, 0, vtp, npad, M=P, N=3 * C, K=C, lda=C, rope_cols=2 * C, pos=pos, cos=cos, sin=sin, tokens=P, heads=heads, qkv_packed=True)
    ops.attention_packed(qkp, 2 * C, 0, npad, qkp, 2 * C, C, npad, vtp, ao, C, B=1, heads=heads, Nq=P, Nk=P, scale=0.125)
t1, t2 = make(), make()
run(t1); run(t2); torch.cuda.synchronize()
ref1, ref2 = t1[5].data.clone(), t2[5].data.clone()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
bad = 0
for it in range(200):
    t1[5].data.zero_(); t2[5].data.zero_(); torch.cuda.synchronize()
    with torch.cuda.stream(s1):
        for _ in range(3): run(t1)
    with torch.cuda.stream(s2):
        for _ in range(3): run(t2)
    torch.cuda.synchronize()
    if not (torch.equal(t1[5].data, ref1) and torch.equal(t2[5].data, ref2)):
        bad += 1
print("concurrent proj+attention mismatches: %d / 200" % bad)
