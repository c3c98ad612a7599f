"""Drop-in for the reference's `models.curope` package (croco/models/curope/{__init__,curope2d}.py) on top of the
MI355X kernel `sp3_rope_2d`.  `croco/models/pos_embed.py:106-110` only needs `cuRoPE2D` to be importable."""
import torch

from . import ops


def rope_2d(tokens, positions, base, fwd):
    """curope.rope_2d(tokens[B,N,H,D] (modified in place), positions[B,N,2] int64, base, fwd)  -- curope.cpp:49-69"""
    ops.rope_2d(tokens, positions, base, fwd)


class cuRoPE2D_func(torch.autograd.Function):
    """curope2d.py:12-29: forward and backward both run the kernel in place (backward with -F0)."""

    @staticmethod
    def forward(ctx, tokens, positions, base, F0=1):
        ctx.save_for_backward(positions)
        ctx.saved_base = base
        ctx.saved_F0 = F0
        rope_2d(tokens, positions, base, F0)
        ctx.mark_dirty(tokens)
        return tokens

    @staticmethod
    def backward(ctx, grad_res):
        positions, base, F0 = ctx.saved_tensors[0], ctx.saved_base, ctx.saved_F0
        rope_2d(grad_res, positions, base, -F0)
        ctx.mark_dirty(grad_res)
        return grad_res, None, None, None


class cuRoPE2D(torch.nn.Module):
    """curope2d.py:32-40: tokens [B,H,N,D]; the kernel sees the [B,N,H,D] transposed view."""

    def __init__(self, freq=100.0, F0=1.0):
        super().__init__()
        self.base = freq
        self.F0 = F0

    def forward(self, tokens, positions):
        cuRoPE2D_func.apply(tokens.transpose(1, 2), positions, self.base, self.F0)
        return tokens
