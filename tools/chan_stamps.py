"""Diagnostic (STAMPS=1 build of csrc only): mean wall-clock per phase of the channel kernel."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from waldboost_amd import _native as nat
from waldboost_amd.engine import PyramidEngine
from waldboost_amd.synth import synth_image
lib = nat.load()
lib.wb_debug_channel_stamps.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
CFG5 = "cfg5" in sys.argv                                      # BASELINE configs[4]: 4K, shrink 4, 12 per octave
H, W, SHRINK, NPO = (2160, 3840, 4, 12) if CFG5 else (1080, 1920, 2, 8)
e = PyramidEngine(H, W, np.uint8, SHRINK, NPO, 1, batch=B)
e.load_images(np.stack([synth_image(H, W, s) for s in range(B)]))
import waldboost_amd as wb
M = wb.load(os.path.join(ROOT, "tests/golden/models/cfg5_d2_T256.pb" if CFG5 else "tests/golden/models/cfg2_d2_T128.pb"))
dm = M.device_cascade()
RANKS = len(sys.argv) > 2 and sys.argv[2] == "ranks"          # the fused detection form: ranks only
e.run_channels(dm if RANKS else None, floats=not RANKS); torch.cuda.synchronize()
for _ in range(3):
    e.launch_channels(dm if RANKS else None, floats=not RANKS)
torch.cuda.synchronize()
n_wg = min(e.n_chan_tiles * B, 1 << 17)
out = (C.c_double * 7)(); life = C.c_double()
lib.wb_debug_channel_stamps(n_wg, out, C.byref(life))
names = ["tile/level/extent loads", "stage source patch + barrier", "row loop (resample)", "leftover columns + barrier",
         "step 2 (gradients, projection, shrink) + barrier", "step 3 (smooth)", "stores (+ ranks)"]
print(f"{'ranks' if RANKS else 'float32 channels'} B={B}: {n_wg} workgroups, mean lifetime {life.value:.2f} us")
for n, v in zip(names, out):
    print(f"  {n:50s} {v:6.2f} us")
