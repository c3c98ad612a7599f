set -x
mkdir -p gpurun_out/r2d
timeout 900 python -m pytest tests/test_preprocess.py -x -q -m gpu > gpurun_out/r2d/pre.log 2>&1
tail -15 gpurun_out/r2d/pre.log
python - <<'PY' > gpurun_out/r2d/pre_time.log 2>&1
import time, numpy as np, torch
from spann3r_amd import preprocess as PP
rng = np.random.default_rng(0)
for hw, res in [((480,640),224), ((1080,1920),(512,384))]:
    rgb = rng.integers(0,256,(*hw,3),dtype=np.uint8)
    d = torch.from_numpy(rgb).cuda()
    for _ in range(3): PP.preprocess_image(d, res)
    torch.cuda.synchronize(); t=time.time()
    for _ in range(50): PP.preprocess_image(d, res)
    torch.cuda.synchronize(); dt=(time.time()-t)/50
    print(hw, res, "device-resident %.1f us/frame" % (dt*1e6))
    t=time.time()
    for _ in range(20): PP.preprocess_image(rgb, res)
    torch.cuda.synchronize(); dt=(time.time()-t)/20
    print(hw, res, "from host %.1f us/frame" % (dt*1e6))
    try:
        from PIL import Image
        from oracle import preprocess_oracle as PO
        p = PO.demo_plan(*hw, (res,res) if isinstance(res,int) else res)
        im = Image.fromarray(rgb); t=time.time()
        for _ in range(10):
            x = np.asarray(im.crop(p["crop0"]).resize(p["resize"], resample=Image.LANCZOS).crop(p["crop1"])).astype(np.float32)/255
        print(hw, res, "PIL crop+resize on host %.1f us/frame" % ((time.time()-t)/10*1e6))
    except Exception as e: print("PIL timing skipped", e)
PY
cat gpurun_out/r2d/pre_time.log
