"""Diagnostic: HIP-event time of the channel and cascade kernels for the library selected by
WB_NATIVE_LIB (A/B runs of two builds on the same box: alternate processes, compare medians)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import waldboost_amd as wb
from waldboost_amd.engine import PyramidEngine
from waldboost_amd.synth import synth_image
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
M = wb.load(os.path.join(ROOT, "tests/golden/models/cfg2_d2_T128.pb"))
dm = M.device_cascade()
if os.environ.get("WB_CASC_JIT", "1") != "0":
    print("specialised:", dm.specialize())
e = PyramidEngine(1080, 1920, np.uint8, 2, 8, 1, batch=B, det_capacity=16384 * B)
e.load_images(np.stack([synth_image(1080, 1920, s) for s in range(B)]))
e.run(dm); torch.cuda.synchronize()
def t(fn, it=20):
    fn(); torch.cuda.synchronize()
    best = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(it): fn()
        b.record(); torch.cuda.synchronize()
        best.append(a.elapsed_time(b) / it)
    return min(best)
fused = e.ranks_for(dm)
us = lambda ms: ms * 1e3 / B
print(f"{os.path.basename(os.environ.get('WB_NATIVE_LIB', 'default')):12s} B={B} per image: channels(float32) {us(t(e.launch_channels)):.2f} us  "
      f"channels(ranks) {us(t(lambda: e.launch_channels(dm, floats=False))) if fused else float('nan'):.2f} us  "
      f"cascade {us(t(lambda: e.run_cascade(dm, ranks=fused))):.2f} us")
