"""Diagnostic: host time of the ways a pageable 1080p image can reach the device."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, time
from waldboost_amd.synth import synth_image
imgs = [synth_image(1080, 1920, i) for i in range(8)]
pin = torch.empty((1080, 1920), dtype=torch.uint8).pin_memory()
pv = pin.numpy()
dev = torch.empty((1080, 1920), dtype=torch.uint8, device="cuda")
for name, fn in (("np.copyto -> pinned", lambda im: np.copyto(pv, im)),
                 ("pinned.copy_(from_numpy)", lambda im: pin.copy_(torch.from_numpy(im))),
                 ("dev.copy_(pageable, non_blocking)", lambda im: dev.copy_(torch.from_numpy(im), non_blocking=True)),
                 ("np.copyto + dev.copy_(pinned, non_blocking)", lambda im: (np.copyto(pv, im), dev.copy_(pin, non_blocking=True)))):
    for im in imgs: fn(im)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(200): fn(imgs[k % 8])
    t = (time.perf_counter() - t0) / 200
    torch.cuda.synchronize()
    print(f"{name:45s} {t*1e6:7.1f} us host time per image")
