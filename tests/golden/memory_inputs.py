"""Deterministic (key, value, query) inputs for the stand-alone memory fixture
(shared by tests/golden/make_golden.py and the tests; no reference import needed)."""
import torch

from spann3r_amd.weights import hash_uniform, _stream_id


def memory_inputs(step, P=196, C=1024, seed=7):
    """Deterministic (key, value, query) for the stand-alone memory fixture."""
    def t(name, scale):
        u = hash_uniform(P * C, _stream_id(seed, "%s%d" % (name, step)))
        return torch.from_numpy(u).reshape(1, P, C) * scale
    k, v, q = t("k", 1.0), t("v", 1.0), t("q", 1.0)
    # keys drift slowly so that attention is not uniform; frames 11 and 12 are near-duplicates of
    # frames 10 and 11 -> the similarity gate (spann3r/model.py:97-118) must skip them
    base = torch.from_numpy(hash_uniform(P * C, _stream_id(seed, "base"))).reshape(1, P, C)
    k = base * 1.5 + k * (0.8 if step not in (11, 12) else 0.05)
    if step in (11, 12):
        kprev = torch.from_numpy(hash_uniform(P * C, _stream_id(seed, "k%d" % 10))).reshape(1, P, C)
        k = base * 1.5 + kprev * 0.8 + k
    q = base * 1.5 + q * 0.8
    return k, v, q
