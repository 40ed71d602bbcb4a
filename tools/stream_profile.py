"""Diagnostic: where the host time of Model.detect_stream goes (cProfile over 400 images of 1080p)."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import waldboost_amd as wb
from waldboost_amd.synth import synth_image
M = wb.load(os.path.join(ROOT, "tests/golden/models/cfg2_d2_T128.pb"))
imgs = [synth_image(1080, 1920, 7000 + i) for i in range(8)]
lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 3
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1
list(M.detect_stream((imgs[i % 8] for i in range(16 * 3 * batch)), lanes=lanes, batch=batch))
torch.cuda.synchronize()
N = 400
t0 = time.perf_counter()
n = sum(len(b) for b in M.detect_stream((imgs[i % 8] for i in range(N)), lanes=lanes, batch=batch))
dt = (time.perf_counter() - t0) / N
print(f"lanes {lanes} batch {batch}: {dt * 1e3:.4f} ms per image, {n / N:.1f} boxes per image")
if len(sys.argv) > 3:
    sys.exit(0)
pr = cProfile.Profile()
pr.enable()
n = sum(len(b) for b in M.detect_stream((imgs[i % 8] for i in range(N)), lanes=lanes, batch=batch))
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
