import os, sys
import numpy as np
sys.path.insert(0, "/root/repo")
import torch
import waldboost_amd as wb
from waldboost_amd.engine import PyramidEngine
from waldboost_amd.synth import synth_image
for B in (1, 64):
    for dt in (np.uint8, np.float32):
        e = PyramidEngine(1080, 1920, dt, 2, 8, 1, batch=B)
        e.load_images(np.stack([synth_image(1080, 1920, s, dt) for s in range(B)]))
        e.run_channels(); torch.cuda.synchronize()
        best = []
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(50): e.launch_octaves()
            b.record(); torch.cuda.synchronize()
            best.append(a.elapsed_time(b) / 50)
        print(f"octaves B={B} {np.dtype(dt).name}: {min(best) * 1e3:.2f} us per launch, {min(best) * 1e3 / B:.2f} us per image")
