"""Host-side logic that needs no GPU: the C-ABI surface, the drop-in boundary (constructor, state dict, error
behaviour) and the multi-process sequence sharding (gloo, world size 2)."""
import argparse
import ctypes
import os
import re
import subprocess
import sys
import tempfile

import pytest
import torch

from conftest import REPO


# ----------------------------------------------------------------------------- C-ABI
def _header():
    return open(os.path.join(REPO, "include", "spann3r_hip.h")).read()


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    from spann3r_amd import lib
    if not os.path.exists(lib.LIB_PATH):
        g.build()
    l = lib.load()
    declared = set(re.findall(r"^\s*(?:int|int64_t|const char\*)\s+(sp3_\w+)\s*\(", _header(), flags=re.M))
    assert len(declared) >= 24
    for name in declared:
        assert hasattr(l, name), "missing export " + name
    assert declared == set(lib.EXPORTS), declared ^ set(lib.EXPORTS)
    assert l.sp3_version() >= 1


def test_ctypes_structs_match_the_header():
    """sizeof / offsetof of the descriptor structs, as gcc sees include/spann3r_hip.h, equal the ctypes mirror."""
    from spann3r_amd import lib
    pairs = [("sp3_gemm_desc", lib.GemmDesc), ("sp3_reduce_ln_desc", lib.ReduceLnDesc), ("sp3_head_part", lib.HeadPart),
             ("sp3_attn_bwd_desc", lib.AttnBwdDesc), ("sp3_bank_write_desc", lib.BankWriteDesc),
             ("sp3_attn_qproj_desc", lib.AttnQProjDesc)]
    lines = ['#include "%s"' % os.path.join(REPO, "include", "spann3r_hip.h"), "#include <stdio.h>", "#include <stddef.h>", "int main(){"]
    exp = []
    for cname, cls in pairs:
        fields = [f[0] for f in cls._fields_]
        lines.append('printf("%%zu\\n", sizeof(%s));' % cname)
        lines += ['printf("%%zu\\n", offsetof(%s, %s));' % (cname, f) for f in fields]
        exp += [ctypes.sizeof(cls)] + [getattr(cls, f).offset for f in fields]
    lines.append("return 0;}")
    with tempfile.TemporaryDirectory() as d:
        c, exe = os.path.join(d, "p.c"), os.path.join(d, "p")
        open(c, "w").write("\n".join(lines) + "\n")
        subprocess.check_call(["gcc", c, "-o", exe])
        vals = [int(x) for x in subprocess.check_output([exe]).split()]
    assert vals == exp


def test_training_host_side_choices():
    """host logic of the training path: RoPE tables are the reference's own fp32 tables (croco/models/pos_embed.py:118-129 as restated by
    the oracle), the bf16 GEMM tile rule follows the sweep"""
    import torch
    from spann3r_amd import train as T
    from oracle import spann3r_oracle as O
    for hd in (64, 48, 32):
        c, s_, n = T._rope_table(100.0, hd, "cpu")
        oc, os_ = O.rope_tables(n, hd // 2, 100.0)
        assert c.shape == (256, hd // 4) and torch.equal(c, oc) and torch.equal(s_, os_)
    assert T._bf16_tile(196, 768) == -1                 # few rows: the library's own choice
    assert T._bf16_tile(784, 768) == 25 and T._bf16_tile(768, 768) == 25 and T._bf16_tile(784, 1024) == 25
    assert T._bf16_tile(784, 2304) == 21 and T._bf16_tile(3072, 768) == 21 and T._bf16_tile(784, 4096) == 21
    assert T._bf16_tile(3920, 4096) == -1 and T._bf16_tile(200704, 256) == -1


def test_argument_validation_reports_errors_without_a_gpu():
    """Every entry point validates before it launches; failures come back as rc != 0 + sp3_last_error()."""
    from spann3r_amd import lib
    l = lib.load()
    d = lib.GemmDesc()
    assert l.sp3_gemm(ctypes.byref(d), None) != 0
    assert b"null A/W/C" in l.sp3_last_error()
    with pytest.raises(RuntimeError, match="sp3_layernorm"):
        lib.check(l.sp3_layernorm(None, 0, None, None, 1e-6, None, 0, 0, 4, 1024, None), "sp3_layernorm")
    assert l.sp3_rope_2d(1, 0, 1, 1, 1, 6, 0, 0, 0, 1, 100.0, 1.0, None) != 0      # D % 4 != 0, curope.cpp's check
    assert b"multiple of 4" in l.sp3_last_error()


# ----------------------------------------------------------------------------- drop-in boundary
def test_state_dict_contract():
    """1101 keys (8 aliases), 658.69 M parameters, same key order as the reference."""
    from spann3r_amd.config import FULL
    from spann3r_amd.weights import param_spec, alias_of
    spec = param_spec(FULL)
    assert len(spec) == 1101
    aliases = [k for k in spec if alias_of(k)]
    assert len(aliases) == 8
    n = sum(int(torch.tensor(s).prod()) for k, s in spec.items() if not alias_of(k))
    assert abs(n / 1e6 - 658.69) < 0.01
    assert spec["dust3r.enc_blocks.23.attn.qkv.weight"] == (3072, 1024)
    assert spec["dust3r.downstream_head1.dpt.act_postprocess.0.1.weight"] == (96, 96, 4, 4)
    assert spec["attn_head_2.0.weight"] == (1792, 1792)


def test_constructor_from_reference_checkpoint(tiny_sd):
    """Spann3R(dus3r_name=<file>) reads the DUSt3R checkpoint the way dust3r/model.py:27-51,94-101 does: geometry from
    the constructor string, dec_blocks duplicated into dec_blocks2 when absent, pos_patch_embed cloned from patch_embed."""
    from spann3r_amd import Spann3R, TINY
    from spann3r_amd.config import Spann3RConfig
    dust3r = {k[len("dust3r."):]: v for k, v in tiny_sd.items() if k.startswith("dust3r.") and "dec_blocks2" not in k}
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "dust3r.pth")
        torch.save({"args": argparse.Namespace(model=TINY.ctor_string()), "model": dust3r}, path)
        m = Spann3R(dus3r_name=path, use_feat=False, init_weights=False)
    assert m.cfg == TINY and hasattr(m, "dust3r") and not m.training is False
    sd = m.state_dict()
    assert list(sd.keys()) == list(tiny_sd.keys())
    assert torch.equal(sd["dust3r.dec_blocks2.3.mlp.fc1.weight"], tiny_sd["dust3r.dec_blocks.3.mlp.fc1.weight"])
    assert torch.equal(sd["pos_patch_embed.proj.weight"], tiny_sd["dust3r.patch_embed.proj.weight"])
    assert sd["dust3r.downstream_head1.dpt.scratch.layer_rn.2.weight"].data_ptr() == \
        sd["dust3r.downstream_head1.dpt.scratch.layer3_rn.weight"].data_ptr()
    assert Spann3RConfig.from_ctor_string(TINY.ctor_string("PatchEmbedDust3R")) == TINY
    with pytest.raises(ValueError):
        Spann3RConfig.from_ctor_string("AsymmetricCroCo3DStereo(enc_embed_dim=768, dec_depth=12)")
    both = Spann3R(dus3r_name=None, cfg=TINY, use_feat=True, mem_pos_enc=True, init_weights=False)
    assert both.cfg.use_feat and both.cfg.mem_pos_enc         # RoPE on the 48-wide heads: engine.narrow_head_slots
    from spann3r_amd.engine import narrow_head_slots
    slots = narrow_head_slots(48).tolist()
    assert len(set(slots)) == 48 and max(slots) < 64 and all((s & 15) < 12 for s in slots)
    assert all(slots[d] ^ 16 == slots[d + 12] for a in (0, 24) for d in range(a, a + 12))      # rotary partners stay partners
    uf = Spann3R(dus3r_name=None, cfg=TINY, use_feat=True, init_weights=False)       # spann3r/model.py:225,239: 768-wide value encoder
    keys = set(uf.state_dict().keys())
    assert "pos_patch_embed.proj.weight" not in keys and tuple(uf.state_dict()["value_out.weight"].shape) == (1024, 768)
    assert tuple(uf.state_dict()["value_encoder.5.mlp.fc1.weight"].shape) == (3072, 768)
    with pytest.raises(FileNotFoundError):                    # a mistyped path must not silently build synthetic weights
        Spann3R(dus3r_name="/nonexistent/DUSt3R.pth")


def test_true_shape_routing():
    """Host logic of Spann3R._forward: which frame lists take the static (hipGraph) path.  Every reference caller sends
    `true_shape` (CPU int32, = image shape, or its transpose for dataset-rotated portraits)."""
    from spann3r_amd import Spann3R, TINY
    m = Spann3R(dus3r_name=None, cfg=TINY, init_weights=False)
    img = torch.zeros(2, 3, 48, 80)
    ts = lambda h, w: torch.tensor([[h, w]] * 2, dtype=torch.int32)
    assert m._uniform_true_hw([{"img": img}] * 3) == (48, 80)
    assert m._uniform_true_hw([{"img": img, "true_shape": ts(48, 80)}] * 3) == (48, 80)
    assert m._uniform_true_hw([{"img": img, "true_shape": ts(80, 48)}] * 3) == (80, 48)         # rotated portrait
    assert m._uniform_true_hw([{"img": img, "true_shape": torch.tensor([[48, 80]])}] * 2) == (48, 80)   # demo.py:109
    assert m._uniform_true_hw([{"img": img, "true_shape": ts(48, 80)}, {"img": img}]) == (48, 80)
    # -> general path: shapes that change inside the sequence or the batch, or that do not match the image
    assert m._uniform_true_hw([{"img": img, "true_shape": ts(48, 80)}, {"img": img, "true_shape": ts(80, 48)}]) is None
    assert m._uniform_true_hw([{"img": img, "true_shape": torch.tensor([[48, 80], [80, 48]])}] * 2) is None
    assert m._uniform_true_hw([{"img": img, "true_shape": ts(80, 48)}, {"img": img}]) is None
    assert m._uniform_true_hw([{"img": img, "true_shape": ts(32, 120)}] * 2) is None
    assert m._uniform_true_hw([{"img": img}, {"img": torch.zeros(2, 3, 80, 48)}]) is None
    assert m._uniform_true_hw([{"img": torch.zeros(1, 3, 50, 80)}] * 2) is None                 # not a multiple of the patch


def test_no_cpu_fallback(tiny_sd):
    from spann3r_amd import Spann3R, TINY
    m = Spann3R(dus3r_name=None, cfg=TINY, init_weights=False).eval()
    m.load_state_dict(tiny_sd)
    frames = [{"img": torch.zeros(1, 3, 32, 32)}] * 2
    with pytest.raises(RuntimeError, match="MI355X"):
        m(frames)
    m.train()
    with pytest.raises(RuntimeError, match="MI355X"):             # a training step (grad enabled) is HIP-only as well
        m(frames)
    with torch.no_grad(), pytest.raises(RuntimeError, match="MI355X"):    # train mode without a tape (active dropout): the same HIP-only ops
        m(frames)


def test_bench_flop_model():
    sys.path.insert(0, REPO)
    import bench
    assert abs(bench.flops_per_sequence(10, 224) / 1e9 - 3260.5) < 1.0
    assert abs(bench.flops_per_sequence(50, 512) / 1e9 - 1.061e5) / 1.061e5 < 0.01


# ----------------------------------------------------------------------------- multi-process sharding (gloo, world 2)
WORKER = r'''
import os, sys
sys.path.insert(0, %(repo)r)
import torch, torch.distributed as dist
from spann3r_amd.runner import shard, gather_stats, aggregate, run_sequences, make_sequence
dist.init_process_group(backend="gloo", init_method="env://")
rank, world = dist.get_rank(), dist.get_world_size()
ids = shard(6, rank, world)
seqs = [make_sequence(s, 3, 32, 48) for s in ids]
# a real forward-shaped callable on the CPU: frames -> (preds, preds_all), the oracle's port of Spann3R.forward on the
# tiny geometry (the HIP model needs a GPU; the sharding / timing / gather path around it is the same)
from oracle import spann3r_oracle as O
from spann3r_amd.config import TINY
from spann3r_amd.weights import synth_state_dict
sd = synth_state_dict(0, TINY)
def fwd(seq):
    preds, preds_all = O.forward(seq, sd, TINY)
    assert len(preds) == len(seq) and len(preds_all) == len(seq) - 1 and tuple(preds[0]["pts3d"].shape) == (1, 32, 48, 3)
    return preds, preds_all
frames, seconds, last = run_sequences(fwd, seqs)
assert torch.isfinite(last[0][-1]["conf"]).all()
dist.barrier()
stats = gather_stats(frames, seconds, extra=[float(sum(ids))])
fps, tot, mx = aggregate(stats)
if rank == 0:
    assert stats.shape == (2, 3), stats.shape
    assert tot == 18 and sorted(stats[:, 2].tolist()) == [6.0, 9.0], stats
    assert mx == float(stats[:, 1].max()) and fps == tot / mx
    # different ranks ran different sequences (seeded per sequence id)
    print("OK", tot)
dist.destroy_process_group()
'''


def test_two_rank_sharding_and_stats_gather():
    with tempfile.TemporaryDirectory() as d:
        script = os.path.join(d, "w.py")
        open(script, "w").write(WORKER % {"repo": REPO})
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", WORLD_SIZE="2")
        procs = [subprocess.Popen([sys.executable, script], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                                  stderr=subprocess.STDOUT) for r in range(2)]
        outs = [p.communicate(timeout=600)[0].decode() for p in procs]
        assert all(p.returncode == 0 for p in procs), outs
        assert "OK 18" in outs[0]
    from spann3r_amd.runner import shard
    assert shard(8, 3, 8) == [3] and shard(10, 1, 4) == [1, 5, 9]


GRAD_WORKER = r'''
import os, sys
sys.path.insert(0, %(repo)r)
import torch, torch.distributed as dist
from spann3r_amd.runner import GradReducer
dist.init_process_group(backend="gloo", init_method="env://")
rank, world = dist.get_rank(), dist.get_world_size()
torch.manual_seed(0)
params = [torch.nn.Parameter(torch.zeros(s)) for s in ((300, 7), (5,), (64, 64), (1000,), (3, 3))]
frozen = torch.nn.Parameter(torch.zeros(4), requires_grad=False)
g = torch.Generator().manual_seed(100 + rank)
for i, p in enumerate(params):
    if not (rank == 1 and i == 1):                       # an unused parameter on one rank: contributes zeros
        p.grad = torch.randn(p.shape, generator=g)
mine = [None if p.grad is None else p.grad.clone() for p in params]
red = GradReducer(params + [frozen], bucket_mb=0.01)     # ~2600 floats per bucket -> several buckets
assert len(red.buckets) >= 3 and red.buckets[0][0] is params[-1]
# flat buckets: every .grad is a view of its bucket, the existing values were adopted
for (flat, items) in red.flat_buffers():
    for p, o in items:
        assert p.grad.data_ptr() == flat.data_ptr() + 4 * o and o %% 1024 == 0
red.start(); red.finish()
# reference: gather every rank's gradients and average
for i, p in enumerate(params):
    loc = mine[i] if mine[i] is not None else torch.zeros(p.shape)
    allg = [torch.zeros_like(loc) for _ in range(world)]
    dist.all_gather(allg, loc)
    assert torch.allclose(p.grad, sum(allg) / world, atol=1e-7), i
assert frozen.grad is None and red.unused_everywhere() == []
# finish() alone (no prepare / start) still reduces: ranks never diverge silently
for p in params:
    p.grad.fill_(float(rank + 1))
red.finish()
assert all(torch.allclose(p.grad, torch.full_like(p.grad, (1 + world) / 2.0)) for p in params)
# overlap=True: the collectives start from inside backward, bucket by bucket, in reverse layer order
torch.manual_seed(1)
net = torch.nn.Sequential(torch.nn.Linear(40, 64), torch.nn.Tanh(), torch.nn.Linear(64, 64), torch.nn.Tanh(), torch.nn.Linear(64, 8))
unused = torch.nn.Parameter(torch.ones(5))
red2 = GradReducer(list(net.parameters()) + [unused], bucket_mb=0.012, overlap=True)
assert len(red2.buckets) >= 3
x = torch.randn(16, 40, generator=torch.Generator().manual_seed(200 + rank))
red2.prepare()
net(x).square().mean().backward()
started = red2.launched_in_backward
red2.finish()
assert started == 0, started                               # the unused parameter sits in bucket 0: everything waits for finish()
ref = torch.nn.Sequential(torch.nn.Linear(40, 64), torch.nn.Tanh(), torch.nn.Linear(64, 64), torch.nn.Tanh(), torch.nn.Linear(64, 8))
ref.load_state_dict(net.state_dict())
tot = [torch.zeros_like(p) for p in ref.parameters()]
for r in range(world):
    ref.zero_grad()
    ref(torch.randn(16, 40, generator=torch.Generator().manual_seed(200 + r))).square().mean().backward()
    for t, p in zip(tot, ref.parameters()):
        t += p.grad
for p, t in zip(net.parameters(), tot):
    assert torch.allclose(p.grad, t / world, atol=1e-7)
assert torch.equal(unused.grad, torch.zeros(5))            # contributed zeros on every rank
assert [q is unused for q in red2.unused_everywhere()] == [True]
# a parameter unused on ONE rank only (a head that rank 1's batch never reaches): hooks fire in different patterns on the
# two ranks, the collectives must still pair up bucket by bucket (ADVICE r2: issue order, not readiness order)
torch.manual_seed(2)
trunk, head_a, head_b = torch.nn.Linear(30, 50), torch.nn.Linear(50, 50), torch.nn.Linear(50, 50)
allp = list(trunk.parameters()) + list(head_a.parameters()) + list(head_b.parameters())
red3 = GradReducer(allp, bucket_mb=0.008, overlap=True)
assert len(red3.buckets) >= 3
for step in range(2):
    red3.zero_grad()
    red3.prepare()
    xin = torch.randn(8, 30, generator=torch.Generator().manual_seed(300 + 10 * step + rank))
    h = torch.tanh(trunk(xin))
    out = head_a(h) if rank == 0 else head_a(h) + head_b(h) * 0.5        # rank 0 never touches head_b
    out.square().mean().backward()
    red3.finish()
    want = [torch.zeros_like(p) for p in allp]
    for r in range(world):
        for p in allp:
            p_ = p.detach()
        t2, a2, b2 = torch.nn.Linear(30, 50), torch.nn.Linear(50, 50), torch.nn.Linear(50, 50)
        t2.load_state_dict(trunk.state_dict()); a2.load_state_dict(head_a.state_dict()); b2.load_state_dict(head_b.state_dict())
        xin2 = torch.randn(8, 30, generator=torch.Generator().manual_seed(300 + 10 * step + r))
        h2 = torch.tanh(t2(xin2))
        o2 = a2(h2) if r == 0 else a2(h2) + b2(h2) * 0.5
        o2.square().mean().backward()
        for w_, q in zip(want, list(t2.parameters()) + list(a2.parameters()) + list(b2.parameters())):
            if q.grad is not None:
                w_ += q.grad
    for p, w_ in zip(allp, want):
        assert torch.allclose(p.grad, w_ / world, atol=1e-6), step
    assert red3.unused_everywhere() == []
# capture mode (train.TrainStep(graph=True) with a process group): the gradient all-reduces still run, the used-parameter bitmap
# is NOT exchanged (no host -> device upload inside a hipGraph capture): every rank replays the same kernel sequence, so the local
# used-set is the global one
torch.manual_seed(3)
net4 = torch.nn.Sequential(torch.nn.Linear(20, 32), torch.nn.Tanh(), torch.nn.Linear(32, 4))
spare = torch.nn.Parameter(torch.ones(7))
red4 = GradReducer(list(net4.parameters()) + [spare], bucket_mb=0.004, overlap=True)
red4.capture = True
red4.zero_grad()
red4.prepare()
net4(torch.randn(6, 20, generator=torch.Generator().manual_seed(400 + rank))).square().mean().backward()
red4.finish()
assert red4._used is None and red4._used_work is None
assert [q is spare for q in red4.unused_everywhere()] == [True]
ref4 = torch.nn.Sequential(torch.nn.Linear(20, 32), torch.nn.Tanh(), torch.nn.Linear(32, 4))
ref4.load_state_dict(net4.state_dict())
tot4 = [torch.zeros_like(p) for p in ref4.parameters()]
for r in range(world):
    ref4.zero_grad()
    ref4(torch.randn(6, 20, generator=torch.Generator().manual_seed(400 + r))).square().mean().backward()
    for t, p in zip(tot4, ref4.parameters()):
        t += p.grad
for p, t in zip(net4.parameters(), tot4):
    assert torch.allclose(p.grad, t / world, atol=1e-7)
# gradient accumulation (train.TrainStep(accum_iter=3), spann3r/training.py:228-233): the first backwards of a window only add into the
# buckets (prepare(arm=False): no collective may start), the last one reduces the SUMS -- equal to DDP's all-reduce after every
# backward, since the mean over ranks is linear.  A parameter only an early iteration touches still counts as used.
torch.manual_seed(4)
net5 = torch.nn.Sequential(torch.nn.Linear(12, 24), torch.nn.Tanh(), torch.nn.Linear(24, 6))
early, never = torch.nn.Parameter(torch.ones(3)), torch.nn.Parameter(torch.ones(2))
red5 = GradReducer(list(net5.parameters()) + [early, never], bucket_mb=0.002, overlap=True)
assert len(red5.buckets) >= 2 and red5.untouched() == []
A = 3
red5.zero_grad()
for it in range(A):
    red5.prepare(arm=(it == A - 1))
    xin = torch.randn(5, 12, generator=torch.Generator().manual_seed(500 + 10 * it + rank))
    loss = net5(xin).square().mean() + (early.sum() * 0.25 if it == 0 else 0.0)
    (loss * (1.0 / A)).backward()
    if it < A - 1:
        assert red5.launched_in_backward == 0 and not red5._armed and red5._work == [None] * len(red5.buckets)
red5.finish()
ref5 = torch.nn.Sequential(torch.nn.Linear(12, 24), torch.nn.Tanh(), torch.nn.Linear(24, 6))
ref5.load_state_dict(net5.state_dict())
tot5 = [torch.zeros_like(p) for p in ref5.parameters()]
for r in range(world):
    for it in range(A):
        ref5.zero_grad()
        (ref5(torch.randn(5, 12, generator=torch.Generator().manual_seed(500 + 10 * it + r))).square().mean() * (1.0 / A)).backward()
        for t, p in zip(tot5, ref5.parameters()):
            t += p.grad
for p, t in zip(net5.parameters(), tot5):
    assert torch.allclose(p.grad, t / world, atol=1e-7)
assert torch.allclose(early.grad, torch.full((3,), 0.25 / A)) and torch.equal(never.grad, torch.zeros(2))
assert [q is never for q in red5.unused_everywhere()] == [True] and [q is never for q in red5.untouched()] == [True]
if rank == 0:
    print("OK grads", started)
dist.destroy_process_group()
'''


def test_two_rank_gradient_reducer():
    """bucketed all-reduce averaging (DDP-equivalent, training.py:322-325) over gloo, world size 2"""
    with tempfile.TemporaryDirectory() as d:
        script = os.path.join(d, "g.py")
        open(script, "w").write(GRAD_WORKER % {"repo": REPO})
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29579", WORLD_SIZE="2")
        procs = [subprocess.Popen([sys.executable, script], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                                  stderr=subprocess.STDOUT) for r in range(2)]
        outs = [p.communicate(timeout=300)[0].decode() for p in procs]
        assert all(p.returncode == 0 for p in procs), outs
        assert "OK grads" in outs[0]
    from spann3r_amd.runner import GradReducer
    p = torch.nn.Parameter(torch.ones(3))
    p.grad = torch.full((3,), 2.0)
    GradReducer([p]).reduce()                            # no process group: a no-op
    assert torch.equal(p.grad, torch.full((3,), 2.0))


@pytest.mark.skipif(not os.path.isdir("/root/reference/croco/models"), reason="needs the reference checkout (build container only)")
def test_curope_shim_is_selected_by_unmodified_reference():
    """shims/curope satisfies the reference's only native FFI: with it on sys.path the UNMODIFIED
    croco/models/pos_embed.py:106-110 picks cuRoPE2D (curope2d.py:6-9 `import curope`) instead of the torch fallback,
    and the module it binds is ours."""
    code = (
        "import sys; sys.dont_write_bytecode = True\n"
        "sys.path[:0] = [%r, %r, '/root/reference']\n"
        "import dust3r.utils.path_to_croco\n"
        "import models.pos_embed as pe, models.curope.curope2d as c2d, curope\n"
        "assert pe.RoPE2D is c2d.cuRoPE2D, pe.RoPE2D\n"
        "assert c2d._kernels is curope and curope.__file__.startswith(%r), curope.__file__\n"
        "import torch\n"
        "try:\n"
        "    curope.rope_2d(torch.zeros(1, 4, 2, 64), torch.zeros(1, 4, 2, dtype=torch.int64), 100.0, 1.0)\n"
        "except RuntimeError as e:\n"
        "    assert 'GPU' in str(e), e\n"
        "else:\n"
        "    raise SystemExit('CPU tensors must be refused')\n"
        "print('shim ok')\n" % (os.path.join(REPO, "shims"), REPO, os.path.join(REPO, "shims")))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "shim ok" in out.stdout, out.stdout + out.stderr
    assert "slow pytorch version" not in out.stdout


def test_lr_schedule_and_accumulation_window_vs_reference():
    """host logic of the training loop (spann3r/training.py:205-233): the per-iteration learning rate against samples of the reference's
    unmodified `adjust_learning_rate` (tests/golden/lr_schedule.npz: warm-up, cosine, warmup_epochs = 0, lr_scale groups), and which
    iterations of an accumulation window zero / update -- no GPU involved"""
    import argparse
    import numpy as np
    from conftest import load_golden
    from spann3r_amd import train as T
    g = load_golden("lr_schedule.npz")
    for k, (lr, min_lr, warm, epochs) in enumerate(g["sets"]):
        args = argparse.Namespace(lr=float(lr), min_lr=float(min_lr), warmup_epochs=int(warm), epochs=int(epochs))
        for e, (ret, lr0, lr1) in zip(g["epochs%d" % k], g["lr%d" % k]):
            mine = T.scheduled_lr(float(e), args)
            assert mine == ret == lr0, (k, e, mine, ret)                     # same expression, same doubles
            assert abs(mine * 0.65 - lr1) <= 1e-18 + 1e-15 * abs(lr1)        # a group's lr_scale multiplies it (misc.py:473-477)

    class Step(T.TrainStep):                                                 # the window bookkeeping without a model
        def __init__(self, a):
            self.accum_iter, self.data_iter_step = a, 0

    for a in (1, 2, 3):
        st, seen = Step(a), []
        for i in range(7):
            seen.append(st._window())
            st.data_iter_step += 1
        # training.py:229-233: update (and zero_grad afterwards, i.e. the NEXT iteration starts a window) when (i + 1) % accum_iter == 0
        assert [w[1] for w in seen] == [(i + 1) % a == 0 for i in range(7)]
        assert [w[0] for w in seen] == [i % a == 0 for i in range(7)]
        st.reset_iteration()
        assert st._window() == (True, a == 1)


def test_pair_indices_vs_reference_make_pairs():
    """runner.pair_indices against dust3r/image_pairs.py make_pairs run on the unmodified reference (tests/golden/make_golden.py pairs):
    every scene graph (complete, swin[-W], oneref[-R], prev), prefilter (seqN, cycN) and symmetrize, pair ORDER included"""
    import json
    from spann3r_amd.runner import pair_indices
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "pairs.json")))
    assert len(g) > 200
    for key, ref in g.items():
        n, sg, pf, sym = key.split("|")
        got = pair_indices(int(n), sg, None if pf == "None" else pf, bool(int(sym)))
        assert [list(p) for p in got] == ref, key
    with pytest.raises(ValueError):
        pair_indices(4, "star")


def test_xcd_maps_are_bijections():
    """The round-5 workgroup -> tile maps (csrc/gemm.hip GemmArgs.xcd_slices, csrc/attention.hip attn_tile_of, csrc/conv.hip conv_tile_of,
    csrc/gemm_sm.hip bm_kernel's (problem, N-tile) pairs) restated in Python: over the launch's grid every tile is produced exactly
    once, surplus workgroups exit, and the property each map exists for holds (the sharers of an operand run on ONE XCD = workgroup
    id % 8; the many-row GEMM's workgroups per XCD differ by at most one N-tile's worth)."""
    # split-K: grid (gx, 1, S), S % 8 == 0, gx % 8 == 0 -> (tile b, slice kz); XCD x owns slices {x, x + 8, ..}
    for gx, S in ((32, 16), (8, 8), (40, 24)):
        seen = {}
        for z in range(S):
            for x in range(gx):
                L = x + gx * z
                xc, jj = L & 7, L >> 3
                q = jj // gx
                kz, b = xc + 8 * q, jj - q * gx
                assert (b, kz) not in seen and 0 <= b < gx and 0 <= kz < S
                seen[(b, kz)] = L % 8
        assert len(seen) == gx * S and all(xcd == kz % 8 for (_, kz), xcd in seen.items())
    # attention: grid (gx, pairs padded to 8, 1) -> (query tile, head, batch); a pair's query tiles on one XCD
    for gx, heads, B in ((13, 12, 2), (13, 16, 1), (32, 16, 2), (7, 3, 5), (1, 1, 1), (64, 12, 1)):
        NP = heads * B
        gy = (NP + 7) // 8 * 8
        seen = {}
        for y in range(gy):
            for x in range(gx):
                L = x + gx * y
                xc, jj = L & 7, L >> 3
                pq = jj // gx
                pair = xc + 8 * pq
                if pair >= NP:
                    continue
                qt, b = jj - pq * gx, pair // heads
                h = pair - b * heads
                assert (qt, h, b) not in seen and 0 <= qt < gx and 0 <= h < heads and 0 <= b < B
                seen[(qt, h, b)] = L % 8
        assert len(seen) == gx * NP
        for h in range(heads):
            for b in range(B):
                assert len({seen[(qt, h, b)] for qt in range(gx)}) == 1
    # 3x3 convolution tiles: 1-D grid of ceil(T / 8) * 8 * NB -> (pixel tile, channel block); a tile's channel blocks on one XCD
    for T, NB in ((98, 4), (392, 2), (28, 8), (1, 4), (9, 1)):
        seen = {}
        for L in range((T + 7) // 8 * 8 * NB):
            xc, jj = L & 7, L >> 3
            q = jj // NB
            nb, t = jj - q * NB, q * 8 + xc
            if t >= T:
                continue
            assert (t, nb) not in seen and 0 <= nb < NB
            seen[(t, nb)] = L % 8
        assert len(seen) == T * NB
        for t in range(T):
            assert len({seen[(t, nb)] for nb in range(NB)}) == 1
    # many-row GEMM: grid (8, mt, nzb), pair p = 8 z + x over (problem, N-tile); per XCD the N-tile counts differ by at most one
    for nt, G in ((18, 2), (12, 2), (24, 1), (12, 1), (7, 2)):
        nzb = (G * nt + 7) // 8
        seen, per_xcd = set(), [0] * 8
        for z in range(nzb):
            for x in range(8):
                p = z * 8 + x
                if p >= nt * G:
                    continue
                grp = 1 if p >= nt else 0
                tile_n = p - grp * nt
                assert (grp, tile_n) not in seen and 0 <= tile_n < nt and grp < G
                seen.add((grp, tile_n))
                per_xcd[x] += 1
        assert len(seen) == nt * G and max(per_xcd) - min(per_xcd) <= 1
    # round 6, many M-tiles (the 512 x 512 whole-sequence encoder).  xm = 2: grid (8, NT, nzm), workgroup l = y + NT zm of an XCD's
    # dispatch order -> blocks of 8 N-tiles x 4 M-tile slots (NT % 8 == 0, nzm % 4 == 0): every (N-tile, slot) once, and each run of 32
    # consecutive workgroups of an XCD shares 4 A tiles and 8 W panels
    for NT, nzm in ((32, 8), (24, 8), (16, 4), (32, 12)):
        seen = {}
        for zm in range(nzm):
            for y in range(NT):
                l = y + NT * zm
                b, r, nb8 = l >> 5, l & 31, NT >> 3
                yy, zz = (b % nb8) * 8 + (r & 7), (b // nb8) * 4 + (r >> 3)
                assert (yy, zz) not in seen and 0 <= yy < NT and 0 <= zz < nzm
                seen[(yy, zz)] = l
        assert len(seen) == NT * nzm
        inv = sorted(seen.items(), key=lambda kv: kv[1])
        for b0 in range(0, NT * nzm, 32):
            blk = [k for k, _ in inv[b0:b0 + 32]]
            assert len({y for y, _ in blk}) == 8 and len({z for _, z in blk}) == 4
    # mblk: grid (8, mt, nz), l = tile_m + mt z -> blocks of 8 M-tiles x all nz z-slots (mt % 8 == 0, 2 <= nz <= 4)
    for mt, nz in ((64, 3), (64, 4), (16, 2), (24, 3)):
        seen = set()
        for z in range(nz):
            for y in range(mt):
                l = y + mt * z
                per = 8 * nz
                b = l // per
                r = l - b * per
                tm, zz = b * 8 + (r & 7), r >> 3
                assert (tm, zz) not in seen and 0 <= tm < mt and 0 <= zz < nz
                seen.add((tm, zz))
        assert len(seen) == mt * nz


def test_bench_kernel_symbols():
    """bench.py books every profiler key under the __global__ template it instantiates (roofline.family): the lean tile names of
    spann3r_amd/ops.py against the kernel families of csrc/gemm_sm.hip; paired and single launches of one instance share a key"""
    import bench
    from spann3r_amd.ops import _TILE_NAMES
    want = {}
    for t, n in _TILE_NAMES.items():
        if t < 30:
            want[n] = "gemm_kernel"
        elif t in (40, 41):
            want[n] = "conv_sm_kernel"
        elif t == 44:
            want[n] = "pv_kernel"
        elif t == 45:
            want[n] = "bm_kernel"
        elif t == 46:
            want[n] = "pvs_kernel"
        elif t < 50:
            want[n] = "sm_kernel"
        else:
            want[n] = "bm_kernel"
    for n, sym in want.items():
        for form in ("gemm", "gemm2"):
            assert bench.kernel_symbol("%s<Abf16,Wbf16,plain,%s>" % (form, n)) == sym, n
    assert bench.instance_key("gemm2<Abf16,Wbf16,plain,lean-rope-k768-32x32xk4>") == "gemm<Abf16,Wbf16,plain,lean-rope-k768-32x32xk4>"
    assert bench.instance_key("attention_packed<bf16>") == "attention_packed<bf16>"
    assert bench.kernel_symbol("attention_packed<bf16>") == "attention_packed_kernel" and bench.kernel_symbol("conv3x3_tile") == "conv3x3_tile_kernel"
