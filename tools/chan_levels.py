"""Channel kernel time per pyramid level (fused rank form, batch B): one launch per level, HIP events."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import waldboost_amd as wb
from waldboost_amd import _native as nat
from waldboost_amd.engine import PyramidEngine
from waldboost_amd.synth import synth_image
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
e = PyramidEngine(1080, 1920, np.uint8, 2, 8, 1, batch=B)
e.load_images(np.stack([synth_image(1080, 1920, s % 4) for s in range(B)]))
M = wb.load(os.path.join(ROOT, "tests/golden/models/cfg2_d2_T128.pb"))
dm = M.device_cascade()
e.run_channels(dm, floats=False); torch.cuda.synchronize()
p = e.plan
tiles = p.chan_tiles()
tot = 0.0
print(f"B={B}; per level: scale, tiles per image, us per image, ns per tile")
for l in range(p.n_levels):
    sel = np.ascontiguousarray(tiles[tiles["level"] == l])
    if sel.size == 0: continue
    td = torch.from_numpy(sel.view(np.uint8).copy()).to(e.dev)
    def go():
        nat.check(e.lib.wb_channels_launch(nat.stream_ptr(), nat.ptr(e.img), p.H * p.W, nat.ptr(e.oct), p.oct_total, e.wb_dtype,
                                           e.batch, nat.ptr(e.levels), p.n_levels, nat.ptr(td), int(sel.size), nat.ptr(e.minmax),
                                           max(p.n_oct, 1), nat.ptr(e.taps), e.spec.func_id, p.shrink, p.smooth,
                                           e.cs_sn.ctypes.data_as(C.POINTER(C.c_double)), None, e.chn_stride, dm.handle,
                                           nat.ptr(e.rank), e.chn_stride), "wb_channels_launch")
    for _ in range(2): go()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): go()
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / 5 / B
    tot += us
    L = p.levels[l] if hasattr(p, "levels") else None
    print(f"level {l:2d}  tiles {sel.size:5d}  {us:7.2f} us/image  {us * 1e3 / sel.size:7.1f} ns/tile")
print(f"sum {tot:.2f} us per image")
