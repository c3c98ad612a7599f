import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spann3r_amd import Spann3R, FULL, ops
from spann3r_amd.weights import synth_state_dict
m = Spann3R(dus3r_name=None, cfg=FULL, init_weights=False); m.load_state_dict(synth_state_dict(0, FULL)); m = m.cuda().eval()
m.set_precision("bf16")
eng = m.engine
torch.manual_seed(0)
f1 = torch.randn(1, 196, 1024, device="cuda"); f2 = torch.randn(1, 196, 1024, device="cuda")
img = torch.randn(1, 3, 224, 224, device="cuda"); fo = torch.empty(1, 196, 1024, device="cuda")
eng.positions(1, 14, 14)
mode = sys.argv[1] if len(sys.argv) > 1 else "enc"
def run(streams, conc):
    main = torch.cuda.current_stream()
    st = eng.side_streams()
    if conc:
        st[3].wait_stream(main)
        with torch.cuda.stream(st[3]):
            if mode == "enc":
                eng.encode_image(img, out=fo, tag="_pre")
            elif mode == "attn":
                ao = eng.wsp("attn_out_pre", 196, 1024)
                for i in range(100):
                    eng._attn_core(eng.wsp("vit_xpA_pre", 196, 1024), eng.stats("vit_stA_pre", 196, 1024), 196, 1, 196, 1024, 16, "enc%d." % (i % 24), eng.positions(1, 14, 14)[1], ao, tag="_pre")
            elif mode == "attnonly":
                ao = eng.wsp("attn_out_pre", 196, 1024)
                qkp = eng.ws("qkp_pre", ops.packed_shape(256, 2048, eng.wdt), eng.wdt, zero=True)
                vtp = eng.ws("vtp_pre", (16 * 256 * 64,), eng.wdt, zero=True)
                for i in range(200):
                    ops.attention_packed(qkp, 2048, 0, 256, qkp, 2048, 1024, 256, vtp, ao, 1024, B=1, heads=16, Nq=196, Nk=196, scale=0.125)
            elif mode == "update":
                for i in range(150):
                    eng._update(eng.wsp("attn_out_pre", 196, 1024), "enc%d.proj" % (i % 24), 196, 1024, 1024, eng.ws("vit_x_pre", (196, 1024)), eng.ws("vit_x_pre", (196, 1024)), eng.wsp("vit_xpB_pre", 196, 1024), eng.stats("vit_stB_pre", 196, 1024))
            else:   # plain GEMM traffic on the third stream
                for _ in range(150):
                    eng._mlp_fc1(eng.wsp("vit_xpA_pre", 196, 1024), eng.stats("vit_stA_pre", 196, 1024), 196, 1024, "enc0.", eng.wsp("mlp_hidden_pre", 196, 4096))
    d1, d2 = eng.decoder(f1, f2, 1, 14, 14, 14, 14, streams=st if streams else None)
    if conc:
        main.wait_stream(st[3])
    torch.cuda.synchronize()
    return [t.clone() for t in d1[1:]] + [t.clone() for t in d2[1:]]
ref = run(True, False)
for conc in (False, True):
    firsts = []
    for it in range(10):
        out = run(True, conc)
        diff = [i for i, (a, b) in enumerate(zip(ref, out)) if not torch.equal(a, b)]
        firsts.append(diff[0] if diff else -1)
    print("third stream busy:", conc, "first differing layer tensor per run:", firsts)
