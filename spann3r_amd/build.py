"""Build libspann3r_hip.so in-tree with hipcc for gfx950 (no torch linkage: pure C-ABI)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libspann3r_hip.so")
SOURCES = ["error.cpp", "gemm.hip", "gemm_sm.hip", "norm_rope.hip", "attention.hip", "memory.hip", "dpt.hip", "conv.hip", "preproc.hip", "loss.hip", "train.hip", "train2.hip", "postproc.hip"]


STAMP = LIB + ".srchash"


def _deps():
    return [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "gemm_sm.h"),
                                                      os.path.join(HERE, "..", "include", "spann3r_hip.h"), os.path.abspath(__file__)]


def source_hash():
    """SHA-256 over every source the library is built from (and this recipe, flags included)"""
    import hashlib
    h = hashlib.sha256()
    for d in _deps():
        h.update(os.path.basename(d).encode())
        with open(d, "rb") as f:
            h.update(f.read())
    h.update(os.environ.get("SP3_HIPCC_EXTRA", "").encode())
    return h.hexdigest()


def needs_build():
    """True unless the .so was built from exactly these sources: the hash recorded next to it (not its mtime -- a stale library
    that merely looks newer than the tree must not pass) equals the tree's."""
    if not os.path.exists(LIB) or not os.path.exists(STAMP):
        return True
    try:
        return open(STAMP).read().strip() != source_hash()
    except OSError:
        return True


# Packed-FP32 VALU ops (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32, formed by the SLP vectoriser in the GEMM epilogues) are
# switched off for the device code: on MI355X with ROCm 7.2 the low half of a v_pk_fma_f32 result came out wrong for
# lanes 48..63 of a wave, rarely (about 1 forward in 50) and only while kernels of three streams shared the CUs; the same
# source without packed ops is bit-stable over thousands of runs (tools/check_decoder_ws.py, DESIGN.md "packed FP32").
# The host pass prints a harmless "not a recognized feature" note for the flag.
DEVICE_FLAGS = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]


def build(force=False, verbose=True, variant=None, extra=()):
    """variant (kernel A/B experiments): builds libspann3r_hip.<variant>.so next to the product library with the extra
    compiler flags; select it with SP3_LIB_PATH (spann3r_amd/lib.py).  The product build ignores it."""
    if variant is None and not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    bdir = os.path.join(HERE, "build" if variant is None else "build_" + variant)
    lib_out = LIB if variant is None else os.path.join(HERE, "libspann3r_hip.%s.so" % variant)
    os.makedirs(bdir, exist_ok=True)
    for s in SOURCES:
        o = os.path.join(bdir, s.rsplit(".", 1)[0] + ".o")
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-c",
               os.path.join(CSRC, s), "-o", o] + DEVICE_FLAGS + os.environ.get("SP3_HIPCC_EXTRA", "").split() + list(extra)
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(o)
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError("hipcc failed on %s" % s)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_out] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    if variant is None:
        with open(STAMP, "w") as f:
            f.write(source_hash() + "\n")
    return lib_out


if __name__ == "__main__":
    # python -m spann3r_amd.build [--force] | --variant NAME [extra hipcc flags ...]
    if "--variant" in sys.argv:
        i = sys.argv.index("--variant")
        print(build(variant=sys.argv[i + 1], extra=sys.argv[i + 2:], verbose=False))
    else:
        build(force="--force" in sys.argv)
        print(LIB)
