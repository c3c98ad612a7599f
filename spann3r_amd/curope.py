"""`models.curope`-compatible surface on top of the MI355X kernel `sp3_rope_2d`.

The reference selects its native RoPE by importing `cuRoPE2D` (croco/models/pos_embed.py:106-110); the compiled
extension it wraps exports one routine, `rope_2d` (croco/models/curope/curope.cpp:49-69).  With `shims/` on sys.path the
reference's own wrapper runs unchanged on `rope_2d` below; the classes here are for code that imports this package
directly (spann3r_amd.train) and are written against the kernel's contract, not the reference's wrapper:

* `rope_2d(tokens[B,N,H,D], positions[B,N,2] int64, base, fwd)` rotates `tokens` in place (fwd = +1) or applies the
  inverse rotation (fwd = -1).  A rotation is orthogonal, so the vector-Jacobian product of the forward op is the
  inverse rotation of the incoming gradient.
* `Rope2D` keeps the incoming gradient intact (the reference rotates `grad_res` in place, which is only safe when
  nobody else holds that tensor): the backward rotates a private copy.
"""
import torch

from . import ops


def rope_2d(tokens, positions, base, fwd):
    ops.rope_2d(tokens, positions, base, fwd)


class Rope2D(torch.autograd.Function):
    """y = R(pos) x on a [B,N,H,D] tensor, in place; dx = R(pos)^T dy, out of place."""

    @staticmethod
    def forward(ctx, x_bnhd, positions, base, sign=1.0):
        ops.rope_2d(x_bnhd, positions, base, sign)
        ctx.mark_dirty(x_bnhd)
        ctx.rot = (positions, float(base), float(sign))      # integer positions: nothing for autograd to track
        return x_bnhd

    @staticmethod
    def backward(ctx, dy):
        positions, base, sign = ctx.rot
        dx = dy.clone(memory_format=torch.contiguous_format)
        ops.rope_2d(dx, positions, base, -sign)
        return dx, None, None, None


cuRoPE2D_func = Rope2D           # the name the reference's wrapper module uses for its autograd function


class cuRoPE2D(torch.nn.Module):
    """Module form with the reference's constructor (`freq`, `F0`) and call convention: tokens arrive as [B,H,N,D] and are
    rotated in place through their [B,N,H,D] view; the same tensor is returned."""

    def __init__(self, freq=100.0, F0=1.0):
        super().__init__()
        self.base, self.F0 = freq, F0

    def forward(self, tokens, positions):
        Rope2D.apply(tokens.transpose(1, 2), positions, self.base, self.F0)
        return tokens
