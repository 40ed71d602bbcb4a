"""BASELINE configs[4]: 3840x2160 images, shrink=4, n_per_oct=12, 256-stage depth-2 cascade (~1e-4 survival).
Secondary workload (the shrink-4 pyramid is this build's extension): kernel times and windows/s, HIP events."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import waldboost_amd as wb
from waldboost_amd.engine import PyramidEngine
from waldboost_amd.synth import synth_image
M = wb.load(os.path.join(ROOT, "tests/golden/models/cfg5_d2_T256.pb"))
dm = M.device_cascade()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
e = PyramidEngine(2160, 3840, np.uint8, 4, 12, 1, batch=B, det_capacity=16384 * B)
e.load_images(np.stack([synth_image(2160, 3840, s) for s in range(B)]))
e.run(dm); n_det = e.ensure_capacity(dm); torch.cuda.synchronize()
def t(fn, it=10):
    fn(); torch.cuda.synchronize()
    best = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(it): fn()
        b.record(); torch.cuda.synchronize()
        best.append(a.elapsed_time(b) / it)
    return min(best)
n_loc = e.plan.n_loc(dm.m, dm.n)
ab = e.plan.algorithmic_bytes(1)
to, tc, tk = t(e.launch_octaves), t(e.launch_channels), t(lambda: e.run_cascade(dm))
g = e.capture(dm)
tg = t(g.replay)
print(f"cfg5 B={B}: {e.plan.n_levels} levels, {n_loc} windows/image, {n_det // B} detections/image, algorithmic {ab['total'] / 1e6:.1f} MB/image")
print(f"  octaves {to / B * 1e3:.1f}  channels {tc / B * 1e3:.1f}  cascade {tk / B * 1e3:.1f} us/image; whole step (graph) {tg / B * 1e3:.1f} us/image")
print(f"  {n_loc * B / tg / 1e-3:.3e} windows/s, {B * 2160 * 3840 / tg / 1e3:.0f} Mpx/s, {ab['total'] * B / tg / 1e6:.0f} GB/s algorithmic = {ab['total'] * B / tg / 1e6 / 8000:.3f} of 8 TB/s")
