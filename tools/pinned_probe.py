"""Where a Model.detect call on a page-locked image spends its time: the upload alone (default stream / side stream),
is_pinned(), the whole call."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import waldboost_amd as wb
from waldboost_amd.synth import synth_image
M = wb.load(os.path.join(ROOT, "tests/golden/models/cfg2_d2_T128.pb"))
imgs = [synth_image(1080, 1920, 7000 + i) for i in range(4)]
pins = [torch.from_numpy(im).pin_memory() for im in imgs]
pimgs = [t.numpy() for t in pins]
dev = torch.empty((1080, 1920), dtype=torch.uint8, device="cuda")
def timeit(fn, n=50):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n): fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
print("is_pinned(from_numpy(view))      %.4f ms" % timeit(lambda i=0: torch.from_numpy(pimgs[i % 4]).is_pinned()))
print("copy_ pinned -> device, default  %.4f ms" % timeit(lambda i=0: dev.copy_(torch.from_numpy(pimgs[i % 4]), non_blocking=True)))
print("copy_ pageable -> device         %.4f ms" % timeit(lambda i=0: dev.copy_(torch.from_numpy(imgs[i % 4]), non_blocking=True)))
s = torch.cuda.Stream()
def side(i=0):
    with torch.cuda.stream(s):
        dev.copy_(torch.from_numpy(pimgs[i % 4]), non_blocking=True)
print("copy_ pinned -> device, side     %.4f ms" % timeit(side))
for k in range(3): M.detect(imgs[0])
print("Model.detect pageable            %.4f ms" % timeit(lambda i=0: M.detect(imgs[i % 4]), 40))
print("Model.detect pinned              %.4f ms" % timeit(lambda i=0: M.detect(pimgs[i % 4]), 40))
def on_side(i=0):
    with torch.cuda.stream(s):
        M.detect(pimgs[i % 4])
for k in range(3): on_side()
print("Model.detect pinned, side stream %.4f ms" % timeit(on_side, 40))
print("Model.detect pinned tensor       %.4f ms" % timeit(lambda i=0: M.detect(pins[i % 4]), 40))
print("Model.detect device tensor       %.4f ms" % timeit(lambda i=0: M.detect(dev), 40))
# as bench.py does it: eight page-locked copies made AFTER the model has detected on pageable arrays; per-call times
imgs8 = [synth_image(1080, 1920, 7000 + i) for i in range(8)]
for im in imgs8[:3]: M.detect(im)
pins8 = [torch.from_numpy(im).pin_memory() for im in imgs8]
p8 = [t.numpy() for t in pins8]
ts = []
for i in range(24):
    t0 = time.perf_counter(); M.detect(p8[i % 8]); ts.append((time.perf_counter() - t0) * 1e3)
print("per-call ms, pinned, bench order:", " ".join("%.3f" % t for t in ts))
