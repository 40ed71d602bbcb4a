"""Diagnostic: time of the grad_mag channel kernel (WB_CHAN_DBG=1 / 2 / 8 / 16: returning after the resize, the
magnitudes, the first and the second triangle pass)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from waldboost_amd.engine import PyramidEngine
from waldboost_amd.chanfunc import SPECS
from waldboost_amd.synth import synth_image
B = 8
key = sys.argv[1] if len(sys.argv) > 1 else "grad_mag"
dt = np.float32 if len(sys.argv) > 2 and sys.argv[2] == "f32" else np.uint8
imgs = np.stack([synth_image(1080, 1920, s, dt) for s in range(B)])
e = PyramidEngine(1080, 1920, dt, 2, 8, 1, batch=B, channels=SPECS[key])
e.load_images(imgs)
e.run_channels(); torch.cuda.synchronize()
best = []
for _ in range(3):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): e.launch_channels()
    b.record(); torch.cuda.synchronize()
    best.append(a.elapsed_time(b) / 10)
print(f"{key} {np.dtype(dt).name} WB_CHAN_DBG={os.environ.get('WB_CHAN_DBG', '0'):3s}: {min(best) / B * 1e3:7.1f} us per image")
