"""Experiment: how fast can K=20 steps on 4 streams be submitted?  (a) one replay per step, round-robin; (b) one graph
per stream holding its 5 steps, replayed from one host thread; (c) the same from 4 host threads."""
import os, sys, time, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import waldboost_amd as wb
from waldboost_amd.engine import PyramidEngine
from waldboost_amd.synth import synth_image
M = wb.load(os.path.join(ROOT, "tests/golden/models/cfg2_d2_T128.pb"))
dm = M.device_cascade()
P, K = 4, 20
engines = []
for i in range(P):
    e = PyramidEngine(1080, 1920, np.uint8, 2, 8, 1, batch=1, det_capacity=16384)
    e.load_images(synth_image(1080, 1920, i)[None])
    e.run(dm)
    engines.append(e)
torch.cuda.synchronize()
lanes = [torch.cuda.Stream() for _ in range(P)]
step_graphs = [e.capture(dm) for e in engines]
lane_graphs = []
for j, e in enumerate(engines):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=lanes[j]):
        for _ in range(K // P):
            e.run(dm)
    lane_graphs.append(g)
head_graphs, tail_graphs = [], []
for j, e in enumerate(engines):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=lanes[j]):
        e.run(dm)
    head_graphs.append(g)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=lanes[j]):
        for _ in range(K // P - 1):
            e.run(dm)
    tail_graphs.append(g)
# (e) ONE graph: the four streams' chains as parallel branches (fork from / join into the capture stream)
fork_graph = torch.cuda.CUDAGraph()
cap = torch.cuda.Stream()
with torch.cuda.graph(fork_graph, stream=cap):
    for st in lanes:
        st.wait_stream(cap)
    for j, e in enumerate(engines):
        with torch.cuda.stream(lanes[j]):
            for _ in range(K // P):
                e.run(dm)
    for st in lanes:
        cap.wait_stream(st)
torch.cuda.synchronize()

def region_e():
    fork_graph.replay()

def region_d():
    for j in range(P):
        with torch.cuda.stream(lanes[j]):
            head_graphs[j].replay()
    for j in range(P):
        with torch.cuda.stream(lanes[j]):
            tail_graphs[j].replay()

def region_a():
    for i in range(K):
        with torch.cuda.stream(lanes[i % P]):
            step_graphs[i % P].replay()
def region_b():
    for j in range(P):
        with torch.cuda.stream(lanes[j]):
            lane_graphs[j].replay()
start = threading.Barrier(P + 1)
done = threading.Barrier(P + 1)
stop = False
def worker(j):
    while True:
        start.wait()
        if stop: return
        with torch.cuda.stream(lanes[j]):
            lane_graphs[j].replay()
        done.wait()
threads = [threading.Thread(target=worker, args=(j,), daemon=True) for j in range(P)]
for t in threads: t.start()
def region_c():
    start.wait(); done.wait()
def timeit(fn, name):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(15):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); th = time.perf_counter(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0, th - t0))
    ts.sort()
    print(f"{name:50s} region {ts[len(ts)//2][0]*1e6:8.1f} us  ({ts[len(ts)//2][0]*1e6/K:6.1f} us/step)   host submit {np.median([h for _, h in ts])*1e6:7.1f} us")
timeit(region_a, "(a) one replay per step, round-robin")
timeit(region_b, "(b) one graph per stream, one host thread")
timeit(region_c, "(c) one graph per stream, one host thread each")
timeit(region_d, "(d) per stream: a 1-step graph, then a 4-step graph")
timeit(region_e, "(e) one graph, four parallel branches")
timeit(region_b, "(b) again")
timeit(region_e, "(e) again")
timeit(region_d, "(d) again")
stop = True; start.wait()
