"""Where Model.detect's wall time goes on one 1080p image: the steps of Model.detect_raw / scan_engine, each
followed by a device synchronisation (so the parts add up to more than one un-instrumented call)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import waldboost_amd as wb
from waldboost_amd import engine as E, _native as nat, channels as CH
from waldboost_amd.synth import synth_image
M = wb.load(os.path.join(ROOT, "tests/golden/models/cfg2_d2_T128.pb"))
imgs = [synth_image(1080, 1920, s) for s in range(4)]
for _ in range(3): M.detect(imgs[0])
torch.cuda.synchronize()
t0 = time.perf_counter()
N = 50
for i in range(N): M.detect(imgs[i % 4])
print(f"{'Model.detect, un-instrumented':45s} {(time.perf_counter() - t0) / N * 1e3:7.3f} ms")
acc = {}
def tick(name, t0, sync=True):
    if sync: torch.cuda.synchronize()
    t = time.perf_counter(); acc[name] = acc.get(name, 0) + (t - t0); return t
for i in range(N):
    img = imgs[i % 4]
    t = time.perf_counter()
    shrink, npo, smooth, spec = CH.read_opts(M.channel_opts)
    dm = M.device_cascade()
    eng = E.get_engine(1080, 1920, img.dtype, shrink, npo, smooth, 1, channels=spec)
    fused = eng.ranks_for(dm)
    t = tick("setup (opts, cached cascade, cached engine)", t)
    eng.load_images(img); t = tick("H2D: host launch part", t, sync=False)
    t = tick("H2D: until done", t)
    eng.run_channels(dm if fused else None, floats=not fused); t = tick("octaves + channels", t)
    stt = eng.run_cascade(dm, ranks=fused); t = tick("cascade", t)
    recs, alive = eng.fetch(dm, stt); t = tick("fetch (pack + D2H + one event wait)", t)
    d = recs.view(nat.DET_DTYPE).reshape(-1)
    d = d[np.lexsort((d["c"], d["r"], d["level"]))]
    inv = eng.inv_scales()[d["level"]]
    x1, y1 = d["c"].astype(np.float32), d["r"].astype(np.float32)
    x2, y2 = (d["c"].astype(np.int32) + 12).astype(np.float32), (d["r"].astype(np.int32) + 12).astype(np.float32)
    boxes = np.stack([x1 * inv, y1 * inv, x2 * inv, y2 * inv], 1)
    t = tick("host: order + boxes", t)
for k, v in acc.items():
    print(f"{k:45s} {v / N * 1e3:7.3f} ms")
print(f"{'sum':45s} {sum(acc.values()) / N * 1e3:7.3f} ms")
