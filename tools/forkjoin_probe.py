"""Experiment: one 1080p image, the step's kernels as ONE chain (memset, octaves, channels, cascade) against a graph
whose channels / cascade work is split into level groups on parallel branches (octaves first, join at the end)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import waldboost_amd as wb
from waldboost_amd import _native as nat
from waldboost_amd.engine import PyramidEngine
from waldboost_amd.synth import synth_image
M = wb.load(os.path.join(ROOT, "tests/golden/models/cfg2_d2_T128.pb"))
dm = M.device_cascade()
e = PyramidEngine(1080, 1920, np.uint8, 2, 8, 1, batch=1)
e.load_images(synth_image(1080, 1920, 0)[None])
stt = e.run(dm); torch.cuda.synchronize()
ref = e.sorted_detections().cpu().numpy().copy()
p = e.plan
ctiles = p.chan_tiles()
ktiles = p.casc_tiles(dm.m, dm.n, dm.tile_rows, dm.tile_cols)
def subset(t, levels):
    sel = np.ascontiguousarray(t[np.isin(t["level"], levels)])
    return int(sel.size), torch.from_numpy(sel.view(np.uint8).copy()).to(e.dev)
def chan(n, td):
    nat.check(e.lib.wb_channels_launch(nat.stream_ptr(), nat.ptr(e.img), p.H * p.W, nat.ptr(e.oct), p.oct_total, e.wb_dtype, 1,
                                       nat.ptr(e.levels), p.n_levels, nat.ptr(td), n, nat.ptr(e.minmax), max(p.n_oct, 1), nat.ptr(e.taps),
                                       e.spec.func_id, p.shrink, p.smooth, e.cs_sn.ctypes.data_as(C.POINTER(C.c_double)), None,
                                       e.chn_stride, dm.handle, nat.ptr(e.rank), e.chn_stride), "chan")
def casc(n, td):
    nat.check(e.lib.wb_cascade_launch(nat.stream_ptr(), dm.handle, nat.ptr(e.rank), nat.WB_DTYPE_RANK8, e.chn_stride, 1, nat.ptr(e.levels),
                                      p.n_levels, nat.ptr(td), n, nat.ptr(e.detb.recs), nat.ptr(e.detb.counts), e.detb.cap,
                                      nat.ptr(stt["alive"])), "casc")
KEEP = []          # the tile tables a captured graph reads must outlive it
def build(groups):
    parts = [(subset(ctiles, g), subset(ktiles, g)) for g in groups]
    KEEP.append(parts)
    side = [torch.cuda.Stream() for _ in parts[1:]]
    KEEP.append(side)
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        main = torch.cuda.current_stream()
        e.reset_step(stt, octaves=True)
        e.launch_octaves()
        for st in side: st.wait_stream(main)
        for k, ((cn, ct), (kn, kt)) in enumerate(parts):
            with torch.cuda.stream(main if k == 0 else side[k - 1]):
                if cn: chan(cn, ct)
                if kn: casc(kn, kt)
        for st in side: main.wait_stream(st)
    return g
def timeit(g, name):
    g.replay(); torch.cuda.synchronize()
    got = e.sorted_detections().cpu().numpy()
    ok = np.array_equal(got, ref)
    ts = []
    for _ in range(30):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); a.record(); g.replay(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
    print(f"{name:60s} {np.median(ts):7.1f} us   same detections: {ok}")
L = list(range(p.n_levels))
timeit(build([L]), "one chain")
timeit(build([L[:8], L[8:]]), "two branches: levels 0-7 | 8..")
timeit(build([L[:4], L[4:8], L[8:]]), "three branches: 0-3 | 4-7 | 8..")
timeit(build([L[0:1], L[1:3], L[3:8], L[8:]]), "four branches: 0 | 1-2 | 3-7 | 8..")
