"""Where Model.detect's wall time goes on one 1080p image (host-side steps, each followed by a sync)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import waldboost_amd as wb
from waldboost_amd import engine as E, _native as nat, channels as CH
from waldboost_amd.synth import synth_image
M = wb.load(os.path.join(ROOT, "tests/golden/models/cfg2_d2_T128.pb"))
img = synth_image(1080, 1920, 0)
for _ in range(3): M.detect(img)
acc = {}
def tick(name, t0):
    torch.cuda.synchronize(); t = time.perf_counter(); acc[name] = acc.get(name, 0) + (t - t0); return t
N = 50
for _ in range(N):
    t = time.perf_counter()
    shrink, npo, smooth, spec = CH.read_opts(M.channel_opts)
    dm = M.device_cascade()
    eng = E.get_engine(1080, 1920, img.dtype, shrink, npo, smooth, 1, channels=spec)
    t = tick("setup (opts, cached cascade, cached engine)", t)
    eng.load_images(img); t = tick("H2D copy", t)
    eng.run_channels(); t = tick("octaves + channels", t)
    stt = eng.run_cascade(dm); t = tick("cascade", t)
    eng.ensure_capacity(dm); t = tick("ensure_capacity (.item())", t)
    det = eng.sorted_detections(); t = tick("compact + sort", t)
    boxes, scores = eng.boxes(det, dm); t = tick("boxes kernel", t)
    alive = stt["alive"][0, :, :len(M)].cpu().numpy(); t = tick("alive D2H", t)
    d = det.cpu().numpy(); b = boxes.cpu().numpy(); s = scores.cpu().numpy(); t = tick("detections D2H", t)
for k, v in acc.items():
    print(f"{k:45s} {v / N * 1e3:7.3f} ms")
print(f"{'sum':45s} {sum(acc.values()) / N * 1e3:7.3f} ms")
