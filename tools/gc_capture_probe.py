"""Diagnostic: which objects, when freed in the middle of a stream capture, invalidate it (torch on ROCm).  One child
process per kind of object: freeing a CUDAGraph throws from its destructor and ends the process.
Measured (torch 2.10 / ROCm 7): page-locked tensor, device tensor: the capture survives; CUDAGraph: "operation not
permitted when stream is capturing" -- hence engine.capturing (collector paused during every capture)."""
import gc, os, subprocess, sys
if len(sys.argv) == 1:
    for k in range(6):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), str(k)], capture_output=True, text=True)
        out = [l for l in r.stdout.splitlines() if "capture" in l]
        err = [l for l in r.stderr.splitlines() if "what():" in l]
        print(out[0] if out else f"case {k}: the process ended with code {r.returncode}: {err[0].strip() if err else r.stderr[-200:]}")
    sys.exit(0)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import waldboost_amd as wb
from waldboost_amd.engine import PyramidEngine
from waldboost_amd.synth import synth_image
M = wb.load(os.path.join(ROOT, "tests/golden/models/cfg2_d2_T128.pb"))
dm = M.device_cascade()
def engine():
    e = PyramidEngine(200, 260, np.uint8, 2, 8, 1, batch=1, det_capacity=16384)
    e.load_images(synth_image(200, 260, 1)[None])
    e.run(dm)
    return e
e = engine()
torch.cuda.synchronize()
def make_pinned():
    t = torch.empty(1 << 16, dtype=torch.uint8).pin_memory()
    t.copy_(torch.zeros(1 << 16, dtype=torch.uint8, device="cuda"), non_blocking=True)
    return t
def make_graph():
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        e.run(dm)
    return g
def make_engine_with_step():
    e2 = engine()
    stt = e2.batch_enqueue(dm); stt = e2.batch_enqueue(dm)      # second call captures and keeps its step
    e2.fetch(dm, stt)                                            # (page-locked read-back buffers)
    return e2
CASES = (("page-locked tensor", make_pinned), ("device tensor", lambda: torch.zeros(1 << 20, device="cuda")),
                   ("CUDAGraph", make_graph), ("stream", torch.cuda.Stream), ("event (recorded)", lambda: (lambda ev: (ev.record(), ev)[1])(torch.cuda.Event())),
                   ("engine that keeps a captured step", make_engine_with_step))
for name, make in CASES[int(sys.argv[1]):int(sys.argv[1]) + 1]:
    obj = [make()]
    torch.cuda.synchronize()
    gc.collect()
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g):
            e.run(dm)
            obj.clear()                     # freed in the middle of the capture
            gc.collect()
            e.run(dm)
        g.replay(); torch.cuda.synchronize()
        print(f"{name:40s}: capture survives")
    except Exception as ex:
        print(f"{name:40s}: capture INVALIDATED -- {str(ex).splitlines()[0][:120]}")
        torch.cuda.synchronize()
