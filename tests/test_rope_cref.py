"""2-D RoPE three ways: the C restatement of the reference's CPU routine (oracle/rope2d_ref.c, curope.cpp:11-47),
the torch oracle (the reference's fallback formulation, pos_embed.py:112-159) and -- on the GPU -- sp3_rope_2d."""
import ctypes
import os
import subprocess

import pytest
import torch

from conftest import REPO, rel_err
from oracle import spann3r_oracle as O


def _cref():
    so = os.path.join(REPO, "oracle", "_ref", "librope2d_ref.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-s", "-C", os.path.join(REPO, "oracle")])
    lib = ctypes.CDLL(so)
    lib.rope2d_ref.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.c_float] * 2
    return lib


def _case(B=2, H=3, nh=5, nw=7, D=64):
    g = torch.Generator().manual_seed(3)
    tok = torch.rand(B, H, nh * nw, D, generator=g) * 2 - 1
    return tok, O.positions(B, nh, nw)


@pytest.mark.parametrize("fwd", [1.0, -1.0])
def test_c_restatement_equals_torch_oracle(fwd):
    tok, pos = _case()
    bnhd = tok.transpose(1, 2).contiguous()           # [B,N,H,D], what curope is handed (curope2d.py:39)
    B, N, H, D = bnhd.shape
    _cref().rope2d_ref(bnhd.data_ptr(), pos.contiguous().data_ptr(), B, N, H, D, 100.0, fwd)
    ref = O.rope2d(tok, pos, 100.0, fwd)
    assert rel_err(bnhd.transpose(1, 2), ref) < 2e-6
    # forward then backward is the identity (the autograd contract of cuRoPE2D_func, curope2d.py:12-29)
    _cref().rope2d_ref(bnhd.data_ptr(), pos.contiguous().data_ptr(), B, N, H, D, 100.0, -fwd)
    assert rel_err(bnhd.transpose(1, 2), tok) < 2e-6


def _real_reference():
    """oracle/_ref/curope_ref*.so = the reference's own curope.cpp CPU routine compiled in the build container."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("build_ref", os.path.join(REPO, "oracle", "build_ref.py"))
    br = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(br)
    if not br.built():
        br.build()
    mod = br.load_ref()
    if mod is None:
        pytest.skip("oracle/_ref/curope_ref not built (no /root/reference on this machine and no prebuilt file)")
    return mod


@pytest.mark.parametrize("fwd", [1.0, -1.0])
def test_real_reference_routine_pins_the_oracle(fwd):
    """The REAL reference code (curope.cpp:11-69) against both restatements."""
    ref_mod = _real_reference()
    tok, pos = _case()
    a = tok.transpose(1, 2).contiguous()
    ref_mod.rope_2d(a, pos, 100.0, fwd)                         # in place on [B,N,H,D], like cuRoPE2D_func.forward
    b = tok.transpose(1, 2).contiguous()
    B, N, H, D = b.shape
    _cref().rope2d_ref(b.data_ptr(), pos.contiguous().data_ptr(), B, N, H, D, 100.0, fwd)
    assert torch.equal(a, b)                                    # same arithmetic, same libm: bit-exact
    assert rel_err(a.transpose(1, 2), O.rope2d(tok, pos, 100.0, fwd)) < 2e-6
    with pytest.raises(RuntimeError):                           # the TORCH_CHECK convention our binding mirrors
        ref_mod.rope_2d(a, pos[:1], 100.0, fwd)


@pytest.mark.gpu
def test_hip_rope_equals_real_reference():
    from spann3r_amd import ops
    ref_mod = _real_reference()
    tok, pos = _case()
    a = tok.transpose(1, 2).contiguous()
    ref_mod.rope_2d(a, pos, 100.0, 1.0)
    t = tok.clone().cuda()
    ops.rope_2d(t.transpose(1, 2), pos.cuda(), 100.0, 1.0)
    assert rel_err(t.cpu(), a.transpose(1, 2)) < 1e-5


@pytest.mark.gpu
def test_hip_rope_equals_c_restatement():
    from spann3r_amd import ops
    tok, pos = _case()
    bnhd = tok.transpose(1, 2).contiguous()
    B, N, H, D = bnhd.shape
    _cref().rope2d_ref(bnhd.data_ptr(), pos.contiguous().data_ptr(), B, N, H, D, 100.0, 1.0)
    t = tok.clone().cuda()
    ops.rope_2d(t.transpose(1, 2), pos.cuda(), 100.0, 1.0)
    assert rel_err(t.cpu(), bnhd.transpose(1, 2)) < 1e-5


@pytest.mark.gpu
def test_curope_dropin_module():
    """spann3r_amd.curope.cuRoPE2D has the reference module's semantics (curope2d.py:32-40): [B,H,N,D] in, in place,
    and the autograd function's backward undoes the forward rotation."""
    from spann3r_amd.curope import cuRoPE2D, cuRoPE2D_func
    tok, pos = _case()
    rope = cuRoPE2D(freq=100.0)
    t = tok.clone().cuda()
    out = rope(t, pos.cuda())
    assert out.data_ptr() == t.data_ptr()
    assert rel_err(out.cpu(), O.rope2d(tok, pos, 100.0, 1.0)) < 1e-5
    x = tok.clone().cuda().transpose(1, 2).contiguous().requires_grad_(True)
    y = cuRoPE2D_func.apply(x.clone(), pos.cuda(), 100.0, 1.0)
    g = torch.ones_like(y)
    y.backward(g.clone())
    # d/dx sum(rope(x)) = rope^-1 applied to ones
    ones = torch.ones_like(tok)
    assert rel_err(x.grad.cpu().transpose(1, 2), O.rope2d(ones, pos, 100.0, -1.0)) < 1e-5
