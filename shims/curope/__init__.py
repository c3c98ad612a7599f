"""Drop-in for the reference's compiled `curope` extension module (croco/models/curope/curope.cpp, PYBIND11 module
`curope` exporting `rope_2d`).  Put this directory's parent on sys.path (or PYTHONPATH) BEFORE importing the reference:

    PYTHONPATH=/path/to/repo/shims:/path/to/repo python demo.py ...

The unmodified croco/models/curope/curope2d.py:6-9 then resolves `import curope as _kernels` to this module, so
croco/models/pos_embed.py:106-110 selects cuRoPE2D (no "slow pytorch version" warning) and every RoPE call of the
reference's own blocks runs the MI355X kernel sp3_rope_2d through the C-ABI.  Same contract as curope.cpp:49-69:
tokens [B,N,H,D] modified in place (fp32 / bf16, stride(3) == 1), positions [B,N,2] int64 contiguous, RuntimeError on a
shape / device mismatch; GPU tensors only (the reference's CPU branch is the reference's own torch fallback)."""
from spann3r_amd.curope import rope_2d  # noqa: F401

__all__ = ["rope_2d"]
