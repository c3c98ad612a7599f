import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spann3r_amd import Spann3R, FULL, ops
from spann3r_amd.weights import synth_state_dict, synth_frames
m = Spann3R(dus3r_name=None, cfg=FULL, init_weights=False); m.load_state_dict(synth_state_dict(0, FULL)); m = m.cuda().eval()
frames = [{"img": f["img"].cuda()} for f in synth_frames(3, 224, 224)]
m.set_precision("bf16"); m.use_graphs = False
m(frames)
run = list(m._runners.values())[0]
eng = m.engine
has_next = len(sys.argv) > 1 and sys.argv[1] == "next"
snaps = []
for it in range(6):
    run.mem.reset()
    run.img_pair[:1].copy_(frames[0]["img"]); run.img_pair[1:].copy_(frames[1]["img"]); run.img_next.copy_(frames[2]["img"])
    torch.cuda.synchronize()
    run._first(has_next)
    torch.cuda.synchronize()
    snap = {}
    for k, t in eng._ws.items():
        d = t.data if isinstance(t, ops.PackedAct) else t
        snap[str(k[:2])] = d.clone()
    snap["k1"] = run.k1.clone(); snap["v"] = run.v.clone()
    snaps.append(snap)
for it in range(1, 6):
    bad = sorted(k for k in snaps[0] if snaps[0][k].shape == snaps[it][k].shape and not torch.equal(snaps[0][k], snaps[it][k]))
    print("run", it, "differing:", len(bad), [b for b in bad if "dec" in b and ("_l1'" in b or "_l0" in b or "_l2'" in b)][:6], [b for b in bad if "dec" not in b and "dpt" not in b][:12])
