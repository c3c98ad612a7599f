import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spann3r_amd import Spann3R, FULL, ops
from spann3r_amd.weights import synth_state_dict, synth_frames
m = Spann3R(dus3r_name=None, cfg=FULL, init_weights=False); m.load_state_dict(synth_state_dict(0, FULL)); m = m.cuda().eval()
frames = [{"img": f["img"].cuda()} for f in synth_frames(3, 224, 224)]
m.set_precision("bf16"); m.use_graphs = False
res = []
for it in range(4):
    preds, _, mem = m(frames, return_memory=True)
    run = list(m._runners.values())[0]
    res.append(dict(p0=preds[0]["conf"].clone(), p1=preds[1]["conf"].clone(), p2=preds[2]["conf"].clone(),
                    k=mem.mem_k.clone(), v=mem.mem_v.clone(), k2=run.k2.clone(), k1=run.k1.clone(), fuse=run.fuse.clone(),
                    khat=run.mem.bank["k_hat"][:, :392].clone(), vhat=run.mem.bank["v_hat_t"][:, :, :392].clone(),
                    attn=mem.mem_attn.clone(), feat_pre=run.feat_pre.clone(), featpair=run.featpair.clone()))
for it in range(1, 4):
    print("run", it, {k: float((res[0][k].float() - res[it][k].float()).abs().max()) for k in res[0]})
    print("   mem_k rows 0-195 diff %.3g, rows 196-391 diff %.3g" % (float((res[0]["k"][:, :196] - res[it]["k"][:, :196]).abs().max()), float((res[0]["k"][:, 196:] - res[it]["k"][:, 196:]).abs().max())))
