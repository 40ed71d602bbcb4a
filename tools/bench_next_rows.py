"""Measurement of the rows 8f kernels (secondary workloads; the bench line stays bench.py):
other channel functions at 1080p, gather_samples and Model.predict on samples.  HIP events."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import waldboost_amd as wb
from waldboost_amd import _native as nat
from waldboost_amd.chanfunc import SPECS
from waldboost_amd.engine import PyramidEngine
from waldboost_amd.samples import gather_samples_device
from waldboost_amd.synth import synth_image


def t(fn, it=20):
    fn(); torch.cuda.synchronize()
    best = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(it): fn()
        b.record(); torch.cuda.synchronize()
        best.append(a.elapsed_time(b) / it)
    return min(best)


B = 8
imgs = np.stack([synth_image(1080, 1920, s) for s in range(B)])
for key in ("grad_hist", "grad_hist_4_u1", "grad_mag_u1", "grad_mag"):
    e = PyramidEngine(1080, 1920, np.uint8, 2, 8, 1, batch=B, channels=SPECS[key])
    e.load_images(imgs)
    e.run_channels(); torch.cuda.synchronize()
    ms = t(e.launch_channels)
    ab = e.plan.algorithmic_bytes(1)["channels_kernel"] * B
    print(f"channels {key:15s} {ms / B * 1e3:7.1f} us/image  {ab / ms / 1e6:7.1f} GB/s algorithmic ({ab / B / 1e6:.1f} MB/image)")

# float32 images (secondary input type of SURVEY 8d): fp64 resample / gradient passes / projection
imgs_f = np.stack([synth_image(1080, 1920, s, np.float32) for s in range(B)])
for key in ("grad_hist", "grad_mag"):
    e = PyramidEngine(1080, 1920, np.float32, 2, 8, 1, batch=B, channels=SPECS[key])
    e.load_images(imgs_f)
    e.run_channels(); torch.cuda.synchronize()
    ms = t(e.launch_channels)
    ab = e.plan.algorithmic_bytes(4)["channels_kernel"] * B
    oc = t(e.launch_octaves)
    print(f"float32 image: channels {key:10s} {ms / B * 1e3:7.1f} us/image  {ab / ms / 1e6:7.1f} GB/s algorithmic; octaves {oc / B * 1e3:.1f} us/image")

M = wb.load(os.path.join(ROOT, "tests/golden/models/cfg2_d2_T128.pb"))
e = PyramidEngine(1080, 1920, np.uint8, 2, 8, 1, batch=1)
e.load_images(imgs[:1]); e.run_channels()
chns = e.level_tensor(0, 0)
u, v, C = chns.shape
rng = np.random.default_rng(0)
for N in (10_000, 1_000_000):
    rs, cs = rng.integers(0, u - 12, N), rng.integers(0, v - 12, N)
    rd = torch.from_numpy(rs.astype(np.int32)).cuda(); cd = torch.from_numpy(cs.astype(np.int32)).cuda()
    out = torch.empty((N, 12, 12, C), dtype=torch.float32, device="cuda")
    lib = nat.load()
    ms = t(lambda: nat.check(lib.wb_gather_samples_launch(nat.stream_ptr(), nat.ptr(chns), nat.WB_DTYPE_F32, u, v, C, nat.ptr(rd), nat.ptr(cd), N, 12, 12, nat.ptr(out))), 10)
    byt = 2 * N * 12 * 12 * C * 4
    print(f"gather_samples N={N:8d}: {ms:8.3f} ms  {byt / ms / 1e6:8.1f} GB/s (read+write, algorithmic)")
    dm = M.device_cascade()
    H = torch.empty(N, dtype=torch.float32, device="cuda"); mask = torch.empty(N, dtype=torch.uint8, device="cuda")
    ms = t(lambda: nat.check(lib.wb_samples_predict_launch(nat.stream_ptr(), dm.handle, nat.ptr(out), nat.WB_DTYPE_F32, N, nat.ptr(H), nat.ptr(mask))), 10)
    print(f"Model.predict  N={N:8d}: {ms:8.3f} ms  {N / ms / 1e3:8.1f} M samples/s  ({N * 2304 / ms / 1e6:.1f} GB/s if every sample byte were read)")

# waldboost.detect(image, M1, M2) (reference __init__.py:75-130): two 128-stage cascades over ONE pyramid, host ndarray in,
# Boxes out.  Rank path: one byte pyramid of the union of both models' thresholds; float path: the float32 pyramid.
import time
from waldboost_amd import engine as E
M1 = wb.load(os.path.join(ROOT, "tests/golden/models/cfg2_d2_T128.pb"))
M2 = wb.load(os.path.join(ROOT, "tests/golden/models/cfg2_d2_T128.pb"))
rng2 = np.random.default_rng(5)
for w, _ in M2:
    w.threshold[w.left >= 0] += np.float32(0.37)
    w.feature[w.left >= 0, 0] = rng2.integers(0, 12, int((w.left >= 0).sum()))
img1 = synth_image(1080, 1920, 99)
for label, no_ranks in (("float32 pyramid", True), ("rank pyramid (union table)", False)):
    E._NO_RANKS = no_ranks
    for _ in range(4):
        out = wb.detect(img1, M1, M2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 30
    for _ in range(n):
        out = wb.detect(img1, M1, M2)
    dt = (time.perf_counter() - t0) / n
    eng = E.get_engine(1080, 1920, np.uint8, 2, 8, 1, 1)
    one = any(st["fails"] == 0 and st["calls"] > 0 for st in eng._multi.values())
    eng._multi.clear()
    print(f"waldboost.detect, 2 models x 128 stages, 1080p, {label:28s}: {dt * 1e3:7.3f} ms per call, {len(out)} boxes "
          f"({'one graph replay, one wait' if one else 'model by model: more than ' + str(eng._FETCH_ROWS) + ' detections of one model -- one read-back does not hold them'}"
          f"; WB_FETCH_ROWS={eng._FETCH_ROWS})")
E._NO_RANKS = False
