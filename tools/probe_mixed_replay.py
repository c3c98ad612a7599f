#!/usr/bin/env python
"""Does a sequence whose steps are PARTLY replayed from hipGraphs and partly launched eagerly equal the all-eager one?
(round 6: a run with MAX_GRAPHS = 16 and per-step fallback to eager launches failed the config-2 fp32 fixture in two-graph mode;
the runner now never mixes inside a sequence -- this probe keeps the question answerable.)  Needs an MI355X."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from spann3r_amd import Spann3R, TINY  # noqa: E402
from spann3r_amd.weights import synth_frames, synth_state_dict  # noqa: E402


def main():
    for precision in ("fp32", "bf16"):
        for single in (True, False):
            m = Spann3R(dus3r_name=None, cfg=TINY, init_weights=False)
            m.load_state_dict(synth_state_dict(0, TINY))
            m = m.cuda().eval().set_precision(precision)
            m.single_graph_step = single
            frames = [{"img": f["img"].cuda()} for f in synth_frames(8, 48, 64, seed=5)]
            m.use_graphs = False
            ref = m(frames)[0]
            m.use_graphs = True
            for _ in range(3):
                out = m(frames)[0]
            same = all(torch.equal(a[k], b[k]) for a, b in zip(ref, out) for k in a)
            run = next(iter(m._runners.values()))
            orig = run._graphed
            for pattern in ("tail of step 3 eager", "part1 of step 3 eager", "steps 5.. eager", "enc eager"):
                def wrapped(key, fn, ug, per_length=False, pattern=pattern):
                    kind = key[0]
                    step_keys = [k for k in run.graphs if k[0] in ("first", "step", "whole")]
                    idx = None
                    if kind in ("first", "step", "whole", "tail"):
                        same_kind = [k for k in run.graphs if k[0] == kind or (kind in ("first", "step") and k[0] in ("first", "step"))]
                        idx = same_kind.index(key) if key in same_kind else None
                    eager = ((pattern == "tail of step 3 eager" and kind == "tail" and idx == 3) or
                             (pattern == "part1 of step 3 eager" and kind in ("step", "whole") and idx == 3) or
                             (pattern == "steps 5.. eager" and idx is not None and idx >= 5) or
                             (pattern == "enc eager" and kind == "enc"))
                    if eager:
                        fn()
                    else:
                        orig(key, fn, ug, per_length)
                run._graphed = wrapped
                got = m(frames)[0]
                run._graphed = orig
                worst = max(float((a[k] - b[k]).abs().max() / b[k].abs().max()) for a, b in zip(got, ref) for k in a)
                print("%s single_graph_step=%s: all-replay == eager: %s; %-24s -> worst rel diff vs eager %.3e" % (precision, single, same, pattern, worst))


if __name__ == "__main__":
    main()
