#!/usr/bin/env python
"""profiles/pmc_hbm_traffic.json from the two PMC passes of tools/gpu_run.sh pmcbench (FETCH_SIZE / WRITE_SIZE per (kernel, grid),
tools/rocpd_pmc_grid.py --json): HBM bytes per launch of the bf16 GEMM instantiations, keyed as bench.py names them.
   python tools/pmc_traffic_json.py gpurun_out/<tag>/pmc_FETCH_SIZE.json gpurun_out/<tag>/pmc_WRITE_SIZE.json "<build description>" """
import json
import re
import sys

NAMES = {(2, 2, 1, 1, 4, 3, 0): "32x32xk4", (7, 1, 1, 4, 2, 4, 2): "112x64wreg8", (4, 4, 4, 2, 1, 3, -1): "256x128pipe",
         (4, 4, 2, 2, 2, 4, -1): "128x128pipe", (4, 4, 2, 1, 2, 3, -1): "128x64pipe", (4, 2, 1, 1, 2, 4, -1): "64x32pipe"}


# the lean instances (csrc/gemm_sm.hip): template arguments -> the tile name spann3r_amd/ops.py gives them
SM = {(3, 4, 8, 16, 0, 2): 30, (2, 2, 4, 12, 0, 2): 31, (4, 4, 8, 16, 0, 0): 32, (2, 4, 4, 12, 0, 0): 33, (2, 2, 8, 16, 0, 1): 34,
      (2, 2, 8, 64, 4, 1): 35, (3, 2, 6, 12, 0, 1): 36, (3, 2, 8, 48, 3, 1): 37, (4, 2, 7, 28, 0, 1): 38, (3, 2, 8, 16, 0, 1): 39,
      (4, 2, 7, 28, 0, 0): 42, (2, 2, 8, 16, 0, 3): 43}
BM = {(4, 2, 4, 16, 3, 2): 50, (2, 2, 4, 16, 3, 2): 51, (2, 2, 4, 12, 3, 2): 52, (4, 2, 4, 16, 3, 0): 53, (2, 2, 4, 16, 3, 0): 54,
      (2, 2, 4, 12, 3, 0): 55, (2, 2, 2, 16, 3, 1): 56, (2, 2, 2, 64, 3, 1): 57, (2, 2, 2, 12, 3, 1): 58, (2, 2, 2, 48, 3, 1): 59,
      (2, 2, 2, 28, 3, 1): 60, (4, 2, 4, 16, 3, 1): 61, (4, 2, 4, 64, 3, 1): 62, (2, 2, 2, 48, 4, 1): 59, (4, 2, 4, 12, 3, 2): 63,
      (4, 2, 4, 12, 3, 0): 64, (4, 2, 4, 16, 3, 4): 45}


def lean_key(row):
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from spann3r_amd.ops import _TILE_NAMES
    n = row["kernel"]
    m = re.search(r"\b(sm|bm)_kernel<([^>]*)>", n)
    if m:
        ints = tuple(int(x) for x in re.findall(r"-?\d+", m.group(2)))[:6]
        tile = (SM if m.group(1) == "sm" else BM).get(ints)
        if tile is None:
            return None
        # sp3_gemm2 (q/k/v + cross k/v projections of a decoder layer in one launch): the only paired lean launch of the 224x224 step
        pair = tile == 31 and row["grid"][2] == 30
        return ("gemm2" if pair else "gemm") + "<Abf16,Wbf16,plain,%s>" % _TILE_NAMES[tile]
    if "conv3x3_tile_kernel" in n or "conv3x3_wide_kernel" in n:        # both pixel tiles of sp3_conv3x3_tile (bench.py times them as one op)
        return "conv3x3_tile"
    if "conv_sm_kernel" in n:
        # (the demangler prints __bf16 template arguments as garbage: take the dtype from the spelling, the tile from the workgroup size)
        bf = "float" not in n and "conv_sm_kernelIf" not in n
        return "gemm<A%s,Wbf16,conv3x3,%s>" % ("bf16" if bf else "f32", _TILE_NAMES[40 if row["grid"][3] == 768 else 41])
    if "attention_packed_kernel" in n:
        # (the third template argument = k-blocks of the q projection computed inside the launch: the decoder's cross-attention)
        m = re.search(r"attention_packed_kernel<\s*\d+,\s*\d+,\s*(\d+)", n) or re.search(r"attention_packed_kernelILi\d+ELi\d+ELi(\d+)E", n)
        return "attention_packed_qproj<bf16>" if (m and int(m.group(1)) > 0) else "attention_packed<bf16>"
    if "pvs_kernel" in n:
        return "gemm<Abf16,Wbf16,softmax,lean-prob-pv-256x128>"
    if re.search(r"\bpv_kernel", n):
        return "gemm<Af32,Wbf16,softmax,lean-softmax-pv-16x64xk8>"
    return None


def key_of(row):
    k = lean_key(row)
    if k:
        return k
    m = re.search(r"gemm_kernelIDF16bDF16bLi0E((?:Lin?\d+E)+)", row["kernel"])
    if not m:
        return None
    ints = [(-int(x[1:]) if x.startswith("n") else int(x)) for x in re.findall(r"Li(n?\d+)E", m.group(1))]
    name = NAMES.get(tuple(ints[:7]))
    if name is None:
        return None
    # sp3_gemm2 (two differently shaped groups in one launch): grid.y = 4 in the per-frame step (2 + 2 problems)
    return ("gemm2" if row["grid"][1] == 4 else "gemm") + "<Abf16,Wbf16,plain,%s>" % name


def collect(path):
    acc = {}
    for r in json.load(open(path))["rows"]:
        k = key_of(r)
        if k:
            a = acc.setdefault(k, [0.0, 0])
            a[0] += r["per_launch"] * r["launches"]
            a[1] += r["launches"]
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items()}


def main():
    fetch, write = collect(sys.argv[1]), collect(sys.argv[2])
    out = {"build": sys.argv[3],
           "command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline "
                      "--no-profile --no-extras --no-graphs (tools/gpu_run.sh pmcbench; tools/pmc_traffic_json.py)",
           "units": "FETCH_SIZE / WRITE_SIZE are KiB per launch; traffic_bytes = 2 * FETCH * 1024 (gfx950: FETCH_SIZE counts 128-B requests as "
                    "64 B, MI355X_MICROARCH.md) + WRITE * 1024",
           "note": "memory-side (fabric) requests of the 8 per-XCD L2s, Infinity-Cache hits included (MI355X_MICROARCH.md): every XCD fetches its "
                   "own copy of the activation panel, so a 196-row launch counts W + 8 x A + C; launches of one kernel are split by grid "
                   "(tools/rocpd_pmc_grid.py) so sp3_gemm2 pairs are separate",
           "kernels": {}}
    for k, (f, n) in fetch.items():
        if k not in write:
            # (the WRITE_SIZE pass did not see this kernel -- a pass that ended early, a name the demangler spelled differently: no number
            #  is better than an understated one; bench.py skips entries without traffic_bytes)
            print("pmc_traffic_json: %s has no WRITE_SIZE row: traffic left out" % k, file=sys.stderr)
            out["kernels"][k] = {"fetch_kib_per_launch": f, "write_kib_per_launch": None, "traffic_bytes": None, "launches": n}
            continue
        w = write[k][0]
        out["kernels"][k] = {"fetch_kib_per_launch": f, "write_kib_per_launch": w, "traffic_bytes": int(round(2 * f * 1024 + w * 1024)), "launches": n}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
