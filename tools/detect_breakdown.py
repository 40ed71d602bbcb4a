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
# the real call's host timeline (no extra synchronisation): where the host is at the end of each step
acc = {}
def lap(name, t0):
    t = time.perf_counter(); acc[name] = acc.get(name, 0) + (t - t0); return t
for i in range(N):
    img = imgs[i % 4]
    t = time.perf_counter()
    CH._validate_image(img)
    shrink, npo, smooth, spec = CH.read_opts(M.channel_opts)
    dm = M.device_cascade()
    eng = E.get_engine(1080, 1920, img.dtype, shrink, npo, smooth, 1, channels=spec)
    t = lap("validate, opts, cached cascade, cached engine", t)
    eng.load_images(img); t = lap("load_images (H2D from pageable memory: blocks)", t)
    fin = eng.detect_run(dm); t = lap("detect_run: graph replay (memset .. copies), wait for the GPU", t)
    keys, boxes_d, scores_d, alive, ordered = fin
    if ordered:
        b, s_ = boxes_d[:keys.size].copy(), scores_d[:keys.size].copy()
    else:
        ks = np.sort(keys)
        at = (ks & np.uint64((1 << 26) - 1)).astype(np.intp)
        b, s_ = boxes_d[at], scores_d[at]
    out = wb.Boxes(b); out.set_field("scores", s_)
    t = lap("host: boxes and scores out of the read-back buffer (ordered on the device), Boxes" if ordered else "host: sort keys, gather boxes and scores, Boxes", t)
print("host timeline of one call (no added synchronisation):")
for k, v in acc.items():
    print(f"  {k:55s} {v / N * 1e3:7.3f} ms")
print(f"  {'sum':55s} {sum(acc.values()) / N * 1e3:7.3f} ms")
