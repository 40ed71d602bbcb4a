"""Diagnostic: Model.detect_stream on random image sequences (shapes, dtypes, runs of equal shapes, lanes, batch) against
Model.detect per image -- bit-identical Boxes in the same order, same n_loc / n_weak.  usage: fuzz_stream.py [first last]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import waldboost_amd as wb
from waldboost_amd.synth import synth_image
lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (0, 40)
MODELS = ["cfg2_d2_T128.pb", "cfg2_gh4u1_d2_T128.pb"]
DTYPES = [np.uint8, np.uint8, np.float32, np.uint16, np.float64]
bad = 0
for seed in range(lo, hi):
    rng = np.random.default_rng(9000 + seed)
    name = MODELS[seed % 2]
    A, B = (wb.load(os.path.join(ROOT, "tests/golden/models", name)) for _ in range(2))
    shapes = [(int(rng.integers(30, 300)), int(rng.integers(30, 400))) for _ in range(3)] + [(9, 9)]
    images = []
    while len(images) < int(rng.integers(5, 30)):
        H, W = shapes[int(rng.integers(0, len(shapes)))]
        dt = np.uint8 if name.endswith("gh4u1_d2_T128.pb") else DTYPES[int(rng.integers(0, len(DTYPES)))]
        for _ in range(int(rng.integers(1, 7))):
            im = synth_image(H, W, int(rng.integers(0, 1 << 30)))
            images.append(im if dt == np.uint8 else (im.astype(dt) * (257 if dt == np.uint16 else 1)).astype(dt))
    lanes, batch = int(rng.integers(1, 5)), int(rng.choice([1, 1, 2, 3, 5, 8]))
    ref = [A.detect(im) for im in images]
    got = list(B.detect_stream(iter(images), lanes=lanes, batch=batch))
    ok = len(got) == len(ref) and (A.n_loc, A.n_weak) == (B.n_loc, B.n_weak)
    for g, r in zip(got, ref):
        ok = ok and np.array_equal(g.get().view(np.uint32), r.get().view(np.uint32)) and \
            np.array_equal(g.get_field("scores").view(np.uint32), r.get_field("scores").view(np.uint32))
    if not ok:
        bad += 1
        print("FAIL seed", seed, "lanes", lanes, "batch", batch, [(im.shape, str(im.dtype)) for im in images][:8], flush=True)
    if seed % 10 == 0:
        print("seed", seed, "images", len(images), "lanes", lanes, "batch", batch, "detections", sum(len(r) for r in ref), "failures so far", bad, flush=True)
print("done; failures:", bad)
