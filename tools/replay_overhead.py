"""Diagnostic: host cost of a hipGraph replay of one batch-1 step vs the GPU time of the step."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import waldboost_amd as wb
from waldboost_amd.engine import PyramidEngine
from waldboost_amd.synth import synth_image
M = wb.load(os.path.join(ROOT, "tests/golden/models/cfg2_d2_T128.pb"))
dm = M.device_cascade()
P = 4
engines = []
for i in range(P):
    e = PyramidEngine(1080, 1920, np.uint8, 2, 8, 1, batch=1, det_capacity=16384)
    e.load_images(synth_image(1080, 1920, i)[None]); e.run(dm); engines.append(e)
graphs = [e.capture(dm) for e in engines]
streams = [torch.cuda.Stream() for _ in range(P)]
torch.cuda.synchronize()
N = 400
for mode in ("4 streams", "1 stream"):
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(N):
            j = i % P
            if mode == "4 streams":
                with torch.cuda.stream(streams[j]): graphs[j].replay()
            else:
                graphs[j].replay()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    print(f"{mode}: issue {1e6 * (t1 - t0) / N:.1f} us/step on the host, total {1e6 * (t2 - t0) / N:.1f} us/step")
# host-only cost: replay an empty-ish graph
g = torch.cuda.CUDAGraph()
x = torch.zeros(1, device="cuda")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    x.add_(1)
torch.cuda.synchronize()
with torch.cuda.graph(g):
    x.add_(1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(2000): g.replay()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"1-kernel graph: issue {1e6 * (t1 - t0) / 2000:.1f} us, total {1e6 * (t2 - t0) / 2000:.1f} us per replay")

# ---- one graph holding the P steps as parallel branches (fork/join inside the capture)
side = [torch.cuda.Stream() for _ in range(P)]
big = torch.cuda.CUDAGraph()
torch.cuda.synchronize()
with torch.cuda.graph(big):
    cur = torch.cuda.current_stream()
    for j in range(P):
        side[j].wait_stream(cur)
        with torch.cuda.stream(side[j]):
            engines[j].run(dm)
    for j in range(P):
        cur.wait_stream(side[j])
torch.cuda.synchronize()
for rep in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(N // P):
        big.replay()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
print(f"one {P}-branch graph: total {1e6 * (t2 - t0) / N:.1f} us/step")
# two such graphs alternating on two streams (the join of one overlaps the fork of the other)
big2 = torch.cuda.CUDAGraph()
engines2 = []
for i in range(P):
    e = PyramidEngine(1080, 1920, np.uint8, 2, 8, 1, batch=1, det_capacity=16384)
    e.load_images(synth_image(1080, 1920, 10 + i)[None]); e.run(dm); engines2.append(e)
torch.cuda.synchronize()
with torch.cuda.graph(big2):
    cur = torch.cuda.current_stream()
    for j in range(P):
        side[j].wait_stream(cur)
        with torch.cuda.stream(side[j]):
            engines2[j].run(dm)
    for j in range(P):
        cur.wait_stream(side[j])
torch.cuda.synchronize()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
for rep in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(N // (2 * P)):
        with torch.cuda.stream(sa): big.replay()
        with torch.cuda.stream(sb): big2.replay()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
print(f"two {P}-branch graphs on two streams: total {1e6 * (t2 - t0) / N:.1f} us/step")
