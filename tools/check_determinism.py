import sys; sys.path.insert(0, '/root/repo')
import torch
from spann3r_amd import Spann3R, TINY, FULL
from spann3r_amd.weights import synth_state_dict, synth_frames
cfg = FULL if len(sys.argv) > 1 and sys.argv[1] == "full" else TINY
m = Spann3R(dus3r_name=None, cfg=cfg, init_weights=False); m.load_state_dict(synth_state_dict(0, cfg)); m = m.cuda().eval()
S = 224 if cfg is FULL else 64
NF = int(sys.argv[3]) if len(sys.argv) > 3 else 5      # >= 6 frames at 224: the encoder's LDS-staged tiles (M >= 1024)
frames = [{"img": f["img"].cuda()} for f in synth_frames(NF, S, S)]
for prec in ("fp32", "bf16"):
    m.set_precision(prec)
    for graphs in (False, True):
        m.use_graphs = graphs
        N = int(sys.argv[2]) if len(sys.argv) > 2 else 4
        ref = m(frames)[0]
        bad = 0
        for j in range(1, N):
            out = m(frames)[0]
            d = max(float((a["conf"] - b["conf"]).abs().max()) for a, b in zip(ref, out))
            bad += d != 0.0
        print(prec, "graphs" if graphs else "eager", "runs differing from run 0: %d of %d" % (bad, N - 1))
