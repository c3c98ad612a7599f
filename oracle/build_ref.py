#!/usr/bin/env python
"""Compile the reference's own curope CPU routine (curope.cpp:11-69) from /root/reference into oracle/_ref/curope_ref*.so.
TEST INFRASTRUCTURE ONLY.  Runs in the build container (where /root/reference exists); the GPU box uses the prebuilt
file that travels with the repo snapshot.  Nothing is copied from the reference: its source is compiled in place."""
import glob
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SRC = "/root/reference/croco/models/curope/curope.cpp"
OUT = os.path.join(HERE, "_ref")


def built():
    return sorted(glob.glob(os.path.join(OUT, "curope_ref*.so")))


def build(verbose=False):
    if built():
        return built()[0]
    if not os.path.exists(REF_SRC):
        return None
    os.makedirs(OUT, exist_ok=True)
    from torch.utils.cpp_extension import load
    load(name="curope_ref", sources=[REF_SRC, os.path.join(HERE, "curope_cuda_stub.cpp")], build_directory=OUT,
         extra_cflags=["-O2"], with_cuda=False, verbose=verbose)
    return built()[0]


def load_ref():
    """import the prebuilt module (None if it was never built)"""
    so = built()
    if not so:
        return None
    import importlib.util
    import torch  # noqa: F401  (the extension links against libtorch)
    spec = importlib.util.spec_from_file_location("curope_ref", so[0])
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    print(build(verbose="-v" in sys.argv))
