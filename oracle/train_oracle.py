"""CPU oracle of the spatial-memory read in train mode.  TEST INFRASTRUCTURE ONLY.

Plain differentiable torch restatement of SpatialMemory.memory_read (spann3r/model.py:145-183) as Spann3R.forward builds
it for training (:475: attn_thresh = 0, mem_dropout): norm_q / norm_k / norm_v LayerNorms (eps 1e-5, :22-24 of the
constructor call sites), affinity / sqrt(C), softmax, dropout as an explicit mask (0 or 1/(1-p)), attn . LN_v(mem_v), + feat.
The tests run it in float64 and let autograd produce the reference gradients of the HIP backward.  Pinned against the
unmodified reference module in tests/test_train.py::test_oracle_matches_reference_memory_read (build container only) and by
the forward parity of the eval path (tests/golden/memory_bank.npz), which shares every line but the dropout."""
import torch
import torch.nn.functional as F


def memory_read_train(feat, mem_k, mem_v, norm_q, norm_k, norm_v, mask=None, eps=1e-5):
    C = feat.shape[-1]
    q = F.layer_norm(feat, (C,), norm_q[0], norm_q[1], eps)
    k = F.layer_norm(mem_k, (C,), norm_k[0], norm_k[1], eps)
    v = F.layer_norm(mem_v, (C,), norm_v[0], norm_v[1], eps)
    aff = torch.einsum("bpc,bxc->bpx", q, k) / torch.sqrt(torch.tensor(float(C), dtype=feat.dtype))
    attn = torch.softmax(aff, dim=-1)
    if mask is not None:
        attn = attn * mask
    return torch.einsum("bpx,bxc->bpc", attn, v) + feat
