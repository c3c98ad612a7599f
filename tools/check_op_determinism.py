import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spann3r_amd import ops
dev = "cuda"
torch.manual_seed(0)
heads, N = 16, 196
C = heads * 64; npad = 256
qkp = ops.PackedAct(npad, 2 * C, torch.bfloat16, dev); qkp.data.normal_()
vtp = torch.randn(heads * npad * 64, device=dev).to(torch.bfloat16)
outs = []
for i in range(5):
    ao = torch.empty(N, C, device=dev)
    ops.attention_packed(qkp, 2 * C, 0, npad, qkp, 2 * C, C, npad, vtp, ao, C, B=1, heads=heads, Nq=N, Nk=N, scale=0.125)
    outs.append(ao.clone())
print("attention_packed repeat diffs:", [float((o - outs[0]).abs().max()) for o in outs[1:]])
M, Cx, Nn = 196, 1024, 3072
x = torch.randn(M, Cx, device=dev)
A0 = torch.randn(M, 256, device=dev); W0 = (torch.randn(Cx, 256, device=dev) * 0.1).to(torch.bfloat16)
W = ops.PackedWeight((torch.randn(Nn, Cx, device=dev) * 0.05).to(torch.bfloat16)); s_n = torch.randn(Nn, device=dev); b = torch.randn(Nn, device=dev)
outs = []
for i in range(5):
    xx = x.clone(); xp = ops.PackedAct(M, Cx, torch.bfloat16, dev); st = torch.empty(M, Cx // 32, 2, device=dev)
    ops.gemm(A0, W0, xx, M=M, N=Cx, K=256, lda=256, ldc=Cx, res1=xx, ldr1=Cx, stats_out=st, c2=xp)
    y = torch.empty(M, Nn, device=dev)
    ops.gemm(xp, W, y, M=M, N=Nn, K=Cx, lda=Cx, ldc=Nn, bias=b, ln=ops.LnFold(st, Cx, s_n))
    outs.append((xx.clone(), st.clone(), y.clone()))
for j in range(3):
    print("lnfold part", j, [float((o[j] - outs[0][j]).abs().max()) for o in outs[1:]])
