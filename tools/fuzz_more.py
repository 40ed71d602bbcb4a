"""Diagnostic: the randomised end-to-end parity test (tests/test_gpu_fuzz.py) on 320 further seeds."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_fuzz as F
lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (64, 384)
bad = []
for seed in range(lo, hi):
    try:
        F.test_random_configuration(seed)
    except Exception as e:                                   # (keep going: report every failing seed)
        bad.append((seed, repr(e)[:200]))
        print("FAIL", seed, repr(e)[:200], flush=True)
    if seed % 40 == 0:
        print("seed", seed, "failures so far:", len(bad), flush=True)
print("done; failures:", bad)
