"""One seed of tests/test_gpu_fuzz.py through the SPECIALISED cascade kernel, RUNS times, record by record against the oracle:
missing / extra / duplicated detections, wrong scores, and where the wrong scores came from (round 4: how the faulty
builds of one hiprtc were characterised -- DESIGN.md section 5).  SEED=558 RUNS=6 python tools/jit_seed_check.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import waldboost_amd as wb
from util import oracle_detect
import test_gpu_fuzz as F
seed = int(os.environ.get("SEED", "558"))
M, img = F.random_configuration(seed)
ref = oracle_detect(M, img)
dm = M.device_cascade()
M.detect_raw(img)
if os.environ.get("DUMP_BASE"):
    os.makedirs(os.environ["DUMP_BASE"], exist_ok=True); os.environ["WB_JIT_DUMP_DIR"] = os.environ["DUMP_BASE"]
print("specialize", dm.specialize(), dm.specialized(), "tile rows", dm.tile_rows)
R = {(l, r, c): s for l, r, c, s in zip(ref["level"].tolist(), ref["r"].tolist(), ref["c"].tolist(), ref["scores"].view(np.uint32).tolist())}
for run in range(int(os.environ.get("RUNS", "6"))):
    res = M.detect_raw(img)
    G = list(zip(res["level"].tolist(), res["r"].tolist(), res["c"].tolist(), res["scores"].view(np.uint32).tolist()))
    gs = {}
    for l, r, c, s in G:
        gs.setdefault((l, r, c), []).append(s)
    missing = sorted(k for k in R if k not in gs)
    extra = sorted(k for k in gs if k not in R)
    dup = sorted(k for k, v in gs.items() if len(v) > 1)
    wrong = sorted(k for k, v in gs.items() if k in R and any(x != R[k] for x in v))
    print("run", run, "n", len(G), len(R), "alive ok", np.array_equal(res["alive"], ref["alive"]), "missing", len(missing), "extra", len(extra), "dup", len(dup), "wrong score", len(wrong))
    byscore = {}
    for k in missing: byscore.setdefault(R[k], []).append(k)
    for k in missing[:10]: print("   missing", k, "tile", (k[1] // dm.tile_rows, k[2] // 64), "row in tile", k[1] % dm.tile_rows, "lane", k[2] % 64, "score %08x" % R[k])
    for k in extra[:10]: print("   extra  ", k, "tile", (k[1] // dm.tile_rows, k[2] // 64), "row in tile", k[1] % dm.tile_rows, "lane", k[2] % 64, "scores", ["%08x" % x for x in gs[k]], "same score as missing:", [byscore.get(x) for x in gs[k]])
    for k in dup[:6]: print("   dup    ", k, ["%08x" % x for x in gs[k]], "ref %08x" % R[k])
    for k in wrong[:6]: print("   wrong  ", k, ["%08x" % x for x in gs[k]], "ref %08x" % R[k], "same as missing:", [byscore.get(x) for x in gs[k]])
    if wrong:
        inv = {}
        for k, s_ in R.items(): inv.setdefault(s_, []).append(k)
        rows = {}
        for k in wrong: rows.setdefault((k[0], k[1]), []).append(k[2])
        for (l, r), cs in sorted(rows.items())[:12]:
            print("   wrong-score row", (l, r), "cols", min(cs), "..", max(cs), "n", len(cs), "grid of level:", ref["alive"][l, 0])
        for k in wrong[:8]:
            print("   wrong", k, "got %08x" % gs[k][0], "= ref score of", inv.get(gs[k][0], "nobody")[:3] if isinstance(inv.get(gs[k][0]), list) else "nobody")
        # partial sums: is the wrong score the reference's running sum after some stage, or missing some stages?
        from util import oracle_model
        from oracle import wb_oracle as orc
        shp, o2, trees, thetas = oracle_model(M)
        k = wrong[0]
        chns = [ch for ch, _ in orc.channel_pyramid(img, o2)][k[0]]
        parts = [float(orc.tree_predict_on_image(t, chns, np.array([k[1]]), np.array([k[2]]))[0]) for t in trees]
        acc = np.cumsum(np.array(parts, np.float32), dtype=np.float32)
        print("   ", k, "leaf values", [round(p, 4) for p in parts]); print("    running sums", [float(x) for x in acc], "got", float(np.array([gs[k][0]], np.uint32).view(np.float32)[0]))
        # where did the wrong windows' scores come from?  implied score on entering stage 8 = got - (own leaves 8..T-1)
        lv = wrong[0][0]
        chns = [ch for ch, _ in orc.channel_pyramid(img, o2)][lv]
        gh, gw = chns.shape[0] - M.shape[0] + 1, chns.shape[1] - M.shape[1] + 1
        rr, cc = np.meshgrid(np.arange(gh), np.arange(gw), indexing="ij"); rr = rr.ravel(); cc = cc.ravel()
        L = np.stack([orc.tree_predict_on_image(t, chns, rr, cc) for t in trees], 0).astype(np.float64)   # [T, n]
        pre8 = L[:8].sum(0); post8 = L[8:].sum(0)
        idx = {(int(r_), int(c_)): i for i, (r_, c_) in enumerate(zip(rr, cc))}
        for k in wrong[:12]:
            i = idx[(k[1], k[2])]
            got = float(np.array([gs[k][0]], np.uint32).view(np.float32)[0])
            h_in = got - post8[i]
            j = int(np.argmin(np.abs(pre8 - h_in)))
            print("   wrong", k, "implied entry score %.4f" % h_in, "own %.4f" % pre8[i], "closest window's:", (int(rr[j]), int(cc[j])), "%.4f" % pre8[j],
                  "| got - own total: %.4f" % (got - pre8[i] - post8[i]))
