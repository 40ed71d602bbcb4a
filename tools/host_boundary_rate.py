"""Rate through the Python boundary (host ndarray in, host Boxes out): PCIe copies, kernels, ordering, boxes."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import waldboost_amd as wb
from waldboost_amd.engine import PyramidEngine
from waldboost_amd.synth import synth_image
M = wb.load(os.path.join(ROOT, "tests/golden/models/cfg2_d2_T128.pb"))
imgs = [synth_image(1080, 1920, s) for s in range(16)]
def rate(label):
    for _ in range(3):
        M.detect(imgs[0])
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        n = 0
        for im in imgs:
            n += len(M.detect(im))
        best = min(best, (time.perf_counter() - t0) / len(imgs))
    print(f"{label:60s} {best * 1e3:.3f} ms/image = {3045278 / best:.3e} windows/s ({n} detections per 16 images)")
rate("Model.detect, one 1080p image per call")
batch = np.stack(imgs)
M.detect_batch(batch); torch.cuda.synchronize()
t0 = time.perf_counter()
for rep in range(4):
    M.detect_batch(batch)
dt = (time.perf_counter() - t0) / (4 * len(imgs))
print(f"{'Model.detect_batch, 16 images per call':60s} {dt * 1e3:.3f} ms/image = {3045278 / dt:.3e} windows/s")
