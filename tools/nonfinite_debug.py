"""Diagnostic: first mismatch between the GPU channel pyramid and the oracle on an image with non-finite pixels."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import waldboost_amd as wb
from oracle import wb_oracle as orc
from test_oracle import nonfinite_image
from waldboost_amd.channels import channel_pyramid
kind = sys.argv[1] if len(sys.argv) > 1 else "inf"
dtype = np.dtype(sys.argv[2] if len(sys.argv) > 2 else "float32").type
img = nonfinite_image((136, 200), dtype, kind)
opts = dict(wb.default_channel_opts)
with np.errstate(all="ignore"):
    ref = list(orc.channel_pyramid(img, dict(opts, channels=orc.grad_hist)))
got = list(channel_pyramid(img, opts))
for l, ((c, s), (rc, rs)) in enumerate(zip(got, ref)):
    nan = np.isnan(rc)
    bad = (np.isnan(c) != nan) | (~nan & (c.view(np.uint32) != rc.view(np.uint32)))
    if bad.any():
        idx = np.argwhere(bad)
        print(f"level {l} shape {c.shape}: {bad.sum()} mismatches; first {idx[:6].tolist()}")
        for i, j, k in idx[:6]:
            print("   ", (i, j, k), "gpu", c[i, j, k], "oracle", rc[i, j, k])
        if l > 10: break
print("done")
