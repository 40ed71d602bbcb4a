"""Diagnostic (STAMPS=1 build of csrc only): mean wall-clock per phase of the cascade tile kernel."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import waldboost_amd as wb
from waldboost_amd import _native as nat
from waldboost_amd.engine import PyramidEngine
from waldboost_amd.synth import synth_image
lib = nat.load()
lib.wb_debug_cascade_stamps.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
CH = sys.argv[2] if len(sys.argv) > 2 else "grad_hist"
M = wb.load(os.path.join(ROOT, "tests/golden/models", {"grad_hist": "cfg2_d2_T128.pb", "grad_hist_4_u1": "cfg2_gh4u1_d2_T128.pb"}[CH]))
dm = M.device_cascade()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
e = PyramidEngine(1080, 1920, np.uint8, 2, 8, 1, batch=B, det_capacity=16384 * B, channels=wb.channels.channel_spec(M.channel_opts["channels"]))
e.load_images(np.stack([synth_image(1080, 1920, s) for s in range(B)]))
e.run(dm); torch.cuda.synchronize()
fused = e.ranks_for(dm)                      # the detection path's form: ranks in the byte tile when the model allows
for _ in range(3):
    stt = e.run_cascade(dm, ranks=fused)
torch.cuda.synchronize()
n_wg = min(stt["n_tiles"] * B, 1 << 16)
out = (C.c_double * 7)(); life = C.c_double()
lib.wb_debug_cascade_stamps(n_wg, out, C.byref(life))
names = ["init+tile load+barrier", "phase A (stages 0-7) + queue", "segments to 8 + re-pack", "segments 8..", "stage-parallel tail", "epilogue barrier+atomic", "copy out"]
print(f"{CH} {'threshold ranks, byte tile' if fused else str(e.spec.dtype) + ' channels'} B={B}: {n_wg} workgroups, mean lifetime {life.value:.2f} us")
for n, v in zip(names, out):
    print(f"  {n:32s} {v:6.2f} us")
