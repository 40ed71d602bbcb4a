"""Diagnostic: Model.detect_stream over many images (two shapes alternating in runs) -- device and host memory must stay flat."""
import os, sys, time, resource
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import waldboost_amd as wb
from waldboost_amd.synth import synth_image
M = wb.load(os.path.join(ROOT, "tests/golden/models/cfg2_d2_T128.pb"))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1
pool = [synth_image(540, 960, i) for i in range(8)] + [synth_image(480, 640, 100 + i) for i in range(8)]
def source(n):
    for i in range(n):
        yield pool[(i // 50 % 2) * 8 + i % 8]
rss = lambda: resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024
marks = []
t0 = time.perf_counter()
for i, bx in enumerate(M.detect_stream(source(N), lanes=3, batch=batch)):
    if i % (N // 5) == 0:
        marks.append((i, torch.cuda.memory_allocated() / 2**20, torch.cuda.memory_reserved() / 2**20, rss()))
dt = time.perf_counter() - t0
marks.append((N, torch.cuda.memory_allocated() / 2**20, torch.cuda.memory_reserved() / 2**20, rss()))
for m in marks:
    print("image %6d: device allocated %8.1f MiB reserved %8.1f MiB, host max RSS %8.1f MiB" % m)
print(f"{N} images, batch {batch}: {dt / N * 1e3:.4f} ms per image; n_loc {M.n_loc}")
