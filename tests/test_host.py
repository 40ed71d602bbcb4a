"""CPU tests of the host logic: level plan, tile tables, wire format, Boxes, scalar handling,
and that the C-ABI library loads and exports every symbol include/waldboost_hip.h declares
(no compute calls: there is no GPU here)."""
import os
import re
import ctypes as C
import zlib

import numpy as np
import pytest

import waldboost_amd as wb
from oracle import wb_oracle as orc
from waldboost_amd import _native as nat
from waldboost_amd.engine import theta_as_f32, orientation_constants, nat_f32_key
from waldboost_amd.plan import PyramidPlan, xcd_order, octave_shapes
from util import GOLDEN, golden_meta

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------------------ C ABI
def test_library_loads_and_exports_every_declared_symbol():
    lib = nat.load()
    header = open(os.path.join(ROOT, "include", "waldboost_hip.h")).read()
    declared = set(re.findall(r"^(?:int|const char \*)\s*(wb_\w+)\(", header, re.M))
    assert declared == set(nat.SYMBOLS), (declared ^ set(nat.SYMBOLS))
    for name in declared:
        assert hasattr(lib, name)
    assert lib.wb_abi_version() == nat.WB_ABI_VERSION == 8
    assert lib.wb_last_error() is not None


@pytest.mark.parametrize("depth,stages", [(1, 32), (2, 128), (3, 20), (2, 5)])
def test_specialised_cascade_kernel_source_compiles_for_gfx950_without_a_gpu(depth, stages):
    """wb_jit.hip hands the tile kernel's own source to hiprtc with a model's stage records as constants; the build
    check runs the same generator and compiler on a synthetic cascade (no device needed)."""
    lib = nat.load()
    n = C.c_int64()
    nat.check(lib.wb_jit_compile_check(depth, stages, b"gfx950", C.byref(n)), "wb_jit_compile_check")
    assert n.value > 4096


@pytest.mark.parametrize("depth,stages,eb", [(2, 128, 2), (3, 40, 2), (2, 1024, 1), (2, 1024, 2), (3, 200, 1)])
def test_specialised_kernel_source_compiles_for_16_bit_tiles_and_long_cascades(depth, stages, eb):
    """The same generator for tiles of two-byte elements (WB_DTYPE_RANK16) and for cascades whose stage table is not
    mirrored in LDS (1024 stages: only the first segments are unrolled, the rest runs the kernel's generic loop).
    The check fails for a build that asks for scratch memory (wb_jit.hip build_checked): depth-3 trees and
    1024 stages are the shapes whose builds did, in round 4."""
    lib = nat.load()
    n = C.c_int64()
    nat.check(lib.wb_jit_compile_check2(depth, stages, eb, b"gfx950", C.byref(n)), "wb_jit_compile_check2")
    assert n.value > 4096


def _kernel_scratch_sizes(blob):
    """{kernel name: .private_segment_fixed_size} from the msgpack metadata notes of the code objects in `blob`."""
    def uint_at(v):
        return (v[0] if v[0] <= 0x7F else int.from_bytes(v[1:2], "big") if v[0] == 0xCC else
                int.from_bytes(v[1:3], "big") if v[0] == 0xCD else int.from_bytes(v[1:5], "big") if v[0] == 0xCE else -1)

    def str_at(k):
        h = blob[k]
        if h == 0xD9:
            return blob[k + 2:k + 2 + blob[k + 1]]
        if h == 0xDA:
            return blob[k + 3:k + 3 + int.from_bytes(blob[k + 1:k + 3], "big")]
        return blob[k + 1:k + 1 + (h & 0x1F)]

    key, out = b".private_segment_fixed_size", {}
    at = blob.find(key)
    while at >= 0:
        name = str_at(blob.rfind(b".name", 0, at) + 5).decode()
        out[name] = uint_at(blob[at + len(key):at + len(key) + 5])
        at = blob.find(key, at + 1)
    return out


def test_no_cascade_kernel_of_the_library_uses_scratch_memory():
    """The cascade kernels keep their state in registers and LDS: `.private_segment_fixed_size` is 0 in the metadata of
    every one of them (so is a specialised build expected to be: wb_model_specialize prefers a scratch-free build, and
    wb_jit_compile_check fails on one with scratch).  The only kernels of the library
    with scratch are four instances of the channel kernel that spill two or three registers outside their loops."""
    nat.load()
    sizes = _kernel_scratch_sizes(open(nat.LIB_PATH, "rb").read())
    casc = {k: v for k, v in sizes.items() if "cascade" in k}
    assert len(casc) > 50, len(casc)               # (every template instance of the tile kernel)
    assert set(casc.values()) == {0}, {k: v for k, v in casc.items() if v}
    with_scratch = {k: v for k, v in sizes.items() if v != 0}
    assert all("channels_kernelI" in k and 0 < v <= 16 for k, v in with_scratch.items()), with_scratch
    assert len(with_scratch) <= 4, with_scratch


def test_abi_struct_sizes_match_header():
    assert nat.LEVEL_DTYPE.itemsize == 64 and nat.TILE_DTYPE.itemsize == 8 and nat.DET_DTYPE.itemsize == 16
    header = open(os.path.join(ROOT, "include", "waldboost_hip.h")).read()
    assert int(re.search(r"#define WB_DET_SHARDS (\d+)", header).group(1)) == nat.WB_DET_SHARDS


def test_channels_tile_query_and_argument_errors_without_gpu():
    import ctypes as C
    lib = nat.load()
    tu, tv = C.c_int(), C.c_int()
    assert lib.wb_channels_tile(nat.WB_CHN_GRAD_HIST, 2, C.byref(tu), C.byref(tv)) == 0 and (tu.value, tv.value) == (16, 64)
    assert lib.wb_channels_tile(nat.WB_CHN_GRAD_HIST_4_U1, 2, C.byref(tu), C.byref(tv)) == 0 and (tu.value, tv.value) == (16, 64)
    assert lib.wb_channels_tile(nat.WB_CHN_GRAD_HIST, 1, C.byref(tu), C.byref(tv)) == 0 and (tu.value, tv.value) == (16, 64)
    assert lib.wb_channels_tile(nat.WB_CHN_GRAD_HIST, 4, C.byref(tu), C.byref(tv)) == 0 and (tu.value, tv.value) == (8, 30)
    assert lib.wb_channels_tile(nat.WB_CHN_GRAD_HIST_4_U1, 4, C.byref(tu), C.byref(tv)) == 0 and (tu.value, tv.value) == (8, 32)
    assert lib.wb_channels_tile(nat.WB_CHN_GRAD_HIST, 3, C.byref(tu), C.byref(tv)) == nat.WB_ERR_UNSUPPORTED
    assert b"shrink=3" in lib.wb_last_error()
    # null pointers are rejected before any HIP call
    assert lib.wb_octaves_launch(None, None, 0, 1, 16, 16, 256, None, 0, None, 1, None) == nat.WB_ERR_INVALID
    with pytest.raises(ValueError):
        nat.check(lib.wb_model_info(None, None), "wb_model_info")


def test_compute_entry_points_fail_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(nat.NativeError):
        wb.channels.grad_hist(np.zeros((16, 16), np.uint8))
    M = wb.load(os.path.join(GOLDEN, "cfg1_d1_T32.pb"))
    with pytest.raises(nat.NativeError):
        M.detect(np.zeros((64, 64), np.uint8))


# ------------------------------------------------------------------------------ plan
@pytest.mark.parametrize("H,W,shrink,npo", [(480, 640, 2, 8), (1080, 1920, 2, 8), (2160, 3840, 4, 12), (97, 131, 1, 3), (7, 100, 2, 8)])
def test_plan_matches_oracle_level_plan(H, W, shrink, npo):
    p = PyramidPlan(H, W, shrink, npo)
    ref = orc.level_plan(H, W, shrink, npo)
    assert p.n_levels == len(ref) and p.octaves == orc.octave_shapes(H, W) == octave_shapes(H, W)
    for lv, r in zip(p.levels, ref):
        assert (lv["oct"], lv["nh"], lv["nw"], lv["scale"]) == (r["oct"], r["nh"], r["nw"], r["scale"])


def test_plan_geometry_of_the_survey():
    p = PyramidPlan(1080, 1920, 2, 8)
    assert p.n_levels == 64 and p.n_loc(12, 12) == 3045278
    ab = p.algorithmic_bytes(1)
    assert ab["chn_write"] == 52029776 and abs(ab["total"] - 128.94e6) < 0.05e6
    assert PyramidPlan(480, 640, 2, 8).n_loc(12, 12) == 407350
    assert PyramidPlan(2160, 3840, 4, 12).n_loc(12, 12) == 4435665


def test_tile_tables_cover_every_window_once_and_csr_groups_by_level():
    p = PyramidPlan(300, 500, 2, 4)
    tiles = p.casc_tiles(12, 12, 32, 64)
    grid = p.window_grid(12, 12)
    cover = [np.zeros((max(r, 1), max(c, 1)), np.int32) for r, c in grid]
    for t in tiles:
        r, c = grid[t["level"]]
        cover[t["level"]][t["ty"] * 32:min(t["ty"] * 32 + 32, r), t["tx"] * 64:min(t["tx"] * 64 + 64, c)] += 1
    for (r, c), cv in zip(grid, cover):
        if r and c:
            assert (cv[:r, :c] == 1).all()
    csr = PyramidPlan.tile_csr(tiles, p.n_levels)
    start, order = csr[:p.n_levels + 1], csr[p.n_levels + 1:]
    assert sorted(order.tolist()) == list(range(tiles.size))
    for l in range(p.n_levels):
        assert (tiles["level"][order[start[l]:start[l + 1]]] == l).all()


@pytest.mark.parametrize("n", [1, 7, 8, 9, 63, 64, 1000, 1671])
def test_xcd_order_is_a_permutation(n):
    o = xcd_order(n)
    assert sorted(o.tolist()) == list(range(n))
    # workgroups b and b+8 (same XCD under round-robin dispatch) get neighbouring tiles
    if n >= 16:
        assert o[8] == o[0] + 1


# ------------------------------------------------------------------------------ wire format
def test_pb_roundtrip_is_byte_identical_to_the_reference_file(tmp_path):
    src = os.path.join(GOLDEN, "mixed_d2_T24.pb")
    M = wb.load(src)
    out = tmp_path / "m.pb"
    M.save(str(out))
    assert out.read_bytes() == open(src, "rb").read()
    meta = golden_meta()["pb_mixed"]
    assert list(M.shape) == meta["shape"] and M.channel_opts["shrink"] == meta["shrink"]
    assert M.channel_opts["n_per_oct"] == meta["n_per_oct"] and M.channel_opts["smooth"] == meta["smooth"]
    assert wb.model.symbol_name(M.channel_opts["channels"]) == meta["func"] == "waldboost.channels.grad_hist"
    th = [t if np.isfinite(t) else "-inf" for t in M.theta]
    assert th == meta["theta"]
    z = np.load(os.path.join(GOLDEN, "pb_mixed_fields.npz"))
    for i, w in enumerate(M.classifier):
        for k in ("feature", "threshold", "left", "right", "prediction"):
            assert np.array_equal(getattr(w, k), z[f"t{i}_{k}"]), (i, k)
        assert w.feature.dtype == np.uint8 and w.left.dtype == np.int8 and w.threshold.dtype == np.float32


def test_pb_errors_and_allow_list(tmp_path):
    bad = tmp_path / "bad.pb"
    bad.write_bytes(b"not a model")
    with pytest.raises(ValueError, match="Cannot read model"):
        wb.load(str(bad))
    # a model naming an unknown channel function is refused instead of eval()'d (reference model.py:27-29)
    from waldboost_amd import model_pb2
    p = model_pb2.Model()
    p.shape.extend([12, 12, 4])
    p.channel_opts.shrink, p.channel_opts.n_per_oct, p.channel_opts.smooth = 2, 8, 1
    p.channel_opts.func = "os.system"
    evil = tmp_path / "evil.pb"
    evil.write_bytes(zlib.compress(p.SerializeToString(), 9))
    with pytest.raises(ValueError, match="unknown channel function"):
        wb.load(str(evil))


def test_model_container_semantics():
    M = wb.Model((12, 12, 4), dict(wb.default_channel_opts))
    assert len(M) == 0 and not M and M.eval_cost == 0
    t = wb.DTree([(1, 2, 3), None, None], [0.5, 0, 0], [1, -1, -1], [2, -1, -1], [0, -1, 1])
    M.append(t, -0.25)
    assert len(M) == 1 and bool(M) and M[0] == (t, -0.25) and list(M) == [(t, -0.25)]
    assert t.feature.tolist() == [[1, 2, 3], [0, 0, 0], [0, 0, 0]] and t.node_idx.tolist() == [0] and t.depth() == 1
    b = M.get_boxes(np.array([2, 3]), np.array([5, 7]), 0.25)
    assert np.array_equal(b.get(), np.array([[20, 8, 68, 56], [28, 12, 76, 60]], np.float32))
    assert len(M.get_boxes(np.array([]), np.array([]), 0.5)) == 0


def test_device_cascade_cache_follows_the_content_of_the_stage_list(monkeypatch):
    """The reference's classifier / theta lists and the trees' arrays are public and mutable (model.py:62-67,
    training.py:24-31): any edit -- in place, by rebinding an array, by swapping stages -- must rebuild the GPU copy,
    no edit must not, and nothing of the caller's may be frozen."""
    from waldboost_amd import engine

    class FakeCascade:
        built = 0

        def __init__(self, shape, classifier, theta):
            FakeCascade.built += 1
            self.snapshot = [w.threshold.copy() for w in classifier]

    monkeypatch.setattr(engine, "DeviceCascade", FakeCascade)
    M = wb.load(os.path.join(GOLDEN, "mixed_d2_T24.pb"))
    d0 = M.device_cascade()
    assert M.device_cascade() is d0 and FakeCascade.built == 1
    w = M.classifier[3]
    assert w.threshold.flags.writeable and w.left.flags.writeable
    w.threshold[0] += 1                                       # in place
    d1 = M.device_cascade()
    assert d1 is not d0 and d1.snapshot[3][0] == w.threshold[0] and M.device_cascade() is d1
    w.prediction = w.prediction.copy()                        # rebound array, same values: harmless either way
    w.prediction[-1] = 7                                      # ... and then edited in place
    d2 = M.device_cascade()
    assert d2 is not d1 and M.device_cascade() is d2
    w.feature[0, 0] ^= 1
    d3 = M.device_cascade()
    assert d3 is not d2
    M.theta[2] = np.float64(M.theta[2])                       # same value, other kind (theta_as_f32 depends on it)
    d4 = M.device_cascade()
    assert d4 is not d3
    M.classifier[0], M.classifier[1] = M.classifier[1], M.classifier[0]
    d5 = M.device_cascade()
    assert d5 is not d4 and M.device_cascade() is d5
    # a tree shared by two models stays editable from both sides
    M2 = wb.Model(M.shape, M.channel_opts)
    M2.append(w, 0.0)
    e0 = M2.device_cascade()
    w.threshold[0] -= 1
    assert M2.device_cascade() is not e0 and M.device_cascade() is not d5


@pytest.mark.parametrize("how", ["deepcopy", "copy", "pickle"])
def test_copied_and_unpickled_models_keep_following_in_place_edits(monkeypatch, how):
    """copy.deepcopy / pickle copy every NumPy view on its own: a restored DTree must get a block of its own that its five
    arrays are views of again, or an in-place edit of the copy would never reach content() and the copy would go on
    scanning with the stale GPU cascade (reference Model / DTree are plain objects: model.py:62-67, training.py:24-31)."""
    import copy
    import pickle
    from waldboost_amd import engine

    class FakeCascade:
        def __init__(self, shape, classifier, theta):
            self.snapshot = [w.threshold.copy() for w in classifier]

    monkeypatch.setattr(engine, "DeviceCascade", FakeCascade)
    M = wb.load(os.path.join(GOLDEN, "mixed_d2_T24.pb"))
    M.classifier[2].note = "kept"                             # (whatever else a caller hung on a tree travels too)
    M.device_cascade()
    N = {"deepcopy": copy.deepcopy, "copy": lambda m: copy.copy(m),
         "pickle": lambda m: pickle.loads(pickle.dumps(m))}[how](M)
    if how == "copy":
        N.classifier = [copy.copy(w) for w in N.classifier]   # (a shallow model copy shares its trees: copy those)
    for w, w0 in zip(N.classifier, M.classifier):
        assert w is not w0 and w._blob is not None
        for a in ("threshold", "prediction", "feature", "left", "right"):
            assert np.shares_memory(getattr(w, a), w._blob) and not np.shares_memory(getattr(w, a), getattr(w0, a))
            assert np.array_equal(getattr(w, a), getattr(w0, a)) and getattr(w, a).dtype == getattr(w0, a).dtype
        assert np.array_equal(w.node, w0.node) and np.array_equal(w.node_idx, w0.node_idx)
    assert N.classifier[2].note == "kept"
    d0 = N.device_cascade()
    assert N.device_cascade() is d0
    before = bytes(N.classifier[5].content())
    N.classifier[5].threshold[0] = 9                          # in place, on the copy
    assert bytes(N.classifier[5].content()) != before
    d1 = N.device_cascade()
    assert d1 is not d0 and d1.snapshot[5][0] == 9
    assert M.classifier[5].threshold[0] != 9                  # the original is untouched


@pytest.mark.parametrize("shape,shrink,n_per_oct,func", [((1080, 1920), 2, 8, "WB_CHN_GRAD_HIST"), ((480, 640), 1, 8, "WB_CHN_GRAD_HIST"),
                                                        ((540, 960), 4, 12, "WB_CHN_GRAD_HIST"), ((200, 264), 2, 8, "WB_CHN_GRAD_HIST_4_U1")])
def test_tile_patch_table_covers_every_tap_of_its_tile(shape, shrink, n_per_oct, func):
    """wb_channels_tile_patches (host, no GPU): the source patch of every staged tile must hold every tap (i0, i0 + 1 on both
    axes, scipy's zoom taps: channels.py:132) of every resized pixel the tile computes, and stay inside the kernel's LDS
    budget; identity levels and up-scaling levels stage nothing."""
    lib = nat.load()
    fid = getattr(nat, func)
    p = PyramidPlan(shape[0], shape[1], shrink, n_per_oct, 1, chan_func=fid)
    table, _ = p.level_table()
    tiles = np.ascontiguousarray(p.chan_tiles())
    out = np.zeros(tiles.size, nat.PATCH_DTYPE)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    assert lib.wb_channels_tile_patches(fid, shrink, 1, vp(table), p.n_levels, vp(tiles), tiles.size, vp(out)) == 0
    tu, tv = C.c_int(), C.c_int()
    assert lib.wb_channels_tile(fid, shrink, C.byref(tu), C.byref(tv)) == 0
    tu, tv, S = tu.value, tv.value, shrink
    rw = S * (tv + 2) + 2
    n_staged = 0
    for t, o in zip(tiles[::7], out[::7]):
        lv = p.levels[int(t["level"])]
        ident = lv["h"] == lv["nh"] and lv["w"] == lv["nw"]
        if ident or lv["h"] <= lv["nh"] or lv["w"] <= lv["nw"]:
            assert o["rows"] == 0 and o["bytes"] == 0
            continue
        if o["rows"] == 0:
            continue                                    # (a patch beyond the LDS budget: the kernel's direct path)
        n_staged += 1
        u0, v0 = int(t["ty"]) * tu, int(t["tx"]) * tv
        vrows = min(lv["u"] - u0, tu) if func == "WB_CHN_GRAD_HIST" else tu
        ry0, rx0, rh = S * (u0 - 1) - 1, S * (v0 - 1) - 1, S * (vrows + 2) + 2
        ys = np.clip(np.arange(ry0, ry0 + rh), 0, lv["nh"] - 1)
        xs = np.clip(np.arange(rx0, rx0 + rw), 0, lv["nw"] - 1)
        ty, tx = PyramidPlan.axis_taps(lv["h"], lv["nh"])[ys], PyramidPlan.axis_taps(lv["w"], lv["nw"])[xs]
        assert ty["i0"].min() >= o["r_lo"] and ty["i1"].max() <= o["r_lo"] + o["rows"] - 1
        assert tx["i0"].min() >= o["c_lo"] and tx["i1"].max() <= o["c_lo"] + o["bytes"] - 1
        assert (ty["i1"] == ty["i0"] + 1).all() and (tx["i1"] == tx["i0"] + 1).all()
        assert o["r_lo"] + o["rows"] <= lv["h"] and o["c_lo"] + o["bytes"] <= lv["w"]
    assert n_staged > 0
    assert lib.wb_channels_tile_patches(nat.WB_CHN_GRAD_MAG, shrink, 1, vp(table), p.n_levels, vp(tiles), tiles.size, vp(out)) == nat.WB_ERR_UNSUPPORTED


def test_boxes_container():
    b = wb.Boxes(np.arange(8, dtype=np.float32).reshape(2, 4), scores=np.array([1.0, 2.0], np.float32))
    assert len(b) == 2 and b.has_field("scores") and not b.has_field("label")
    assert np.array_equal(b[1].get(), [[4, 5, 6, 7]]) and b[1].get_field("scores")[0] == 2.0
    c = wb.concatenate([b, b.normalized(2.0)])
    assert len(c) == 4 and np.array_equal(c.get()[2], [0, 2, 4, 6]) and c.get_field("scores").tolist() == [1, 2, 1, 2]
    assert len(wb.concatenate([])) == 0
    with pytest.raises(ValueError):
        b.set_field("x", np.zeros(3))


# ------------------------------------------------------------------------------ scalars / constants
def test_theta_as_f32_follows_numpy2_promotion():
    hs = np.array([0.1, np.nextafter(np.float32(0.1), np.float32(1)), np.nextafter(np.float32(0.1), np.float32(0))], np.float32)
    for theta in (0.1, np.float64(0.1), np.float32(0.1), 1, np.int64(1), np.float64(1e300), float("-inf"), np.float64(-1e-50)):
        with np.errstate(over="ignore"):
            want = hs >= theta
        got = hs >= theta_as_f32(theta)
        assert np.array_equal(want, got), theta
    assert np.isnan(theta_as_f32(float("nan")))


def test_orientation_constants_are_the_reference_ones():
    cs_sn = orientation_constants()
    c, s = orc.orientation_table()
    assert np.array_equal(cs_sn[:4], c) and np.array_equal(cs_sn[4:], s)
    assert [float(x).hex() for x in cs_sn] == ['0x1.0000000000000p+0', '0x1.6a09e667f3bcdp-1', '0x1.1a62633145c07p-54',
                                               '-0x1.6a09e667f3bccp-1', '0x0.0p+0', '0x1.6a09e667f3bccp-1',
                                               '0x1.0000000000000p+0', '0x1.6a09e667f3bcdp-1']


def test_f32_key_is_order_preserving():
    v = np.array([-np.inf, -3.5, -0.0, 0.0, 1e-30, 2.0, np.inf], np.float32)
    k = [int(nat_f32_key(x)) for x in v]
    assert k == sorted(k) and len(set(k)) >= 6


def test_uint8_projection_identity_holds_on_the_cpu_too():
    """The claim behind csrc/wb_channels.hip:project_int, checked with NumPy over all gradient pairs."""
    cs_sn = orientation_constants()
    g = np.arange(-1020, 1021, dtype=np.float64)
    GX, GY = np.meshgrid(g, g, indexing="ij")
    chi = np.float32(cs_sn[5])
    clo = np.float32(cs_sn[5] - np.float64(chi))
    for k in range(4):
        ref = np.abs((GX * cs_sn[k] - GY * cs_sn[4 + k]).astype(np.float32))
        if k == 0:
            fast = np.abs(GX).astype(np.float32)
        elif k == 2:
            fast = np.abs(GY).astype(np.float32)
        else:
            d = np.abs(GX - GY) if k == 1 else np.abs(GX + GY)
            lo = (d.astype(np.float32) * clo).astype(np.float32)
            fast = (d * np.float64(chi) + lo.astype(np.float64)).astype(np.float32)      # == fmaf(d, chi, d*clo)
        slow = (GX != 0) & ((GY == 0) | (GX == GY) | (GX == -GY))
        assert np.array_equal(fast[~slow].view(np.uint32), ref[~slow].view(np.uint32)), k


# ------------------------------------------------------------------------------ rows 8f: names, labels, pool (no GPU)
@pytest.mark.parametrize("fn,name", [("grad_hist_4_u1", "waldboost.fpga.channels.grad_hist_4_u1"),
                                     ("grad_mag_u1", "waldboost.fpga.channels.grad_mag_u1"),
                                     ("grad_mag", "waldboost.channels.grad_mag")])
def test_pb_of_the_other_channel_functions_round_trips_byte_for_byte(tmp_path, fn, name):
    src = os.path.join(GOLDEN, f"{fn}_d2_T24.pb")               # written by the reference's own Model.save
    M = wb.load(src)
    spec = wb.channels.channel_spec(M.channel_opts["channels"])
    assert spec.key == fn and spec.reference_name == name and wb.model.symbol_name(M.channel_opts["channels"]) == name
    assert M.shape[2] == spec.n_channels
    out = tmp_path / "m.pb"
    M.save(str(out))
    assert out.read_bytes() == open(src, "rb").read()
    # our own function objects and the short aliases resolve to the same specs
    assert wb.channels.CHANNEL_FUNCS["waldboost_amd.channels." + fn] is spec.func
    assert wb.fpga.grad_hist_4_u1 is wb.channels.SPECS["grad_hist_4_u1"].func


def test_iou_and_label_boxes_host_logic():
    from waldboost_amd.boxes import iou
    from waldboost_amd.samples import SampleLabel, label_boxes, select_candidates
    dt = wb.Boxes(np.array([[0, 0, 10, 10], [5, 5, 15, 15], [100, 100, 110, 110], [0, 0, 9, 10]], "f"))
    gt = wb.Boxes(np.array([[0, 0, 10, 10], [200, 200, 210, 210]], "f"), ignore=np.array([0, 1]))
    t = iou(dt, gt)
    assert t.shape == (4, 2) and t[0, 0] == 1.0 and abs(t[1, 0] - 25 / 175) < 1e-12 and t[2, 0] == 0 and abs(t[3, 0] - 0.9) < 1e-12
    label_boxes(dt, gt, min_tp_iou=0.7, max_fp_iou=0.3)
    assert list(dt.get_field("tp_label")) == [SampleLabel.TRUE_POSITIVE, SampleLabel.FALSE_POSITIVE,
                                              SampleLabel.FALSE_POSITIVE, SampleLabel.TRUE_POSITIVE]
    assert list(dt.get_field("instance_id")) == [0, 0, 0, 0]
    # a detection on an ignored ground-truth box is neither TP nor FP
    dt2 = wb.Boxes(np.array([[200, 200, 210, 210]], "f"))
    label_boxes(dt2, gt)
    assert list(dt2.get_field("tp_label")) == [SampleLabel.IGNORE] and list(dt2.get_field("instance_id")) == [1]
    # candidate caps
    np.random.seed(1)
    many = wb.Boxes(np.tile(np.array([[300, 300, 310, 310]], "f"), (50, 1)))
    label_boxes(many, gt, max_fp_candidates=7)
    assert 1 <= (many.get_field("tp_label") == SampleLabel.FALSE_POSITIVE).sum() <= 7       # np.random.choice draws with replacement
    assert select_candidates(np.array([0, 1, 1, 0, 1], bool), 10).tolist() == [1, 2, 4]
    with pytest.raises(ValueError):
        label_boxes(dt, wb.Boxes(np.zeros((1, 4), "f"), ignore=np.zeros((1, 1))))
    label_boxes(None, gt)                                        # no-op, like the reference


def test_sample_pool_bookkeeping_without_a_gpu():
    from waldboost_amd.samples import SampleLabel, SamplePool
    pool = SamplePool(min_tp=2, min_fp=3)
    assert pool.pool_stats() == dict(num_tp=0, num_fp=0)
    bx = wb.Boxes(np.zeros((4, 4), "f"), scores=np.array([1.0, -np.inf, 2.0, 0.5], "f"),
                  tp_label=np.array([1, -1, -1, 1], np.int32), samples=np.arange(4 * 2 * 2 * 1, dtype=np.float32).reshape(4, 2, 2, 1))
    pool.samples = bx
    assert pool.pool_stats() == dict(num_tp=2, num_fp=2)
    pool.remove_low_scoring()
    assert pool.pool_stats() == dict(num_tp=2, num_fp=1)
    X, H = pool.get_false_positives()
    assert X.shape == (1, 2, 2, 1) and H.tolist() == [2.0]
    X[...] = -1                                                  # copies: the pool is not touched
    assert pool.get_samples(SampleLabel.FALSE_POSITIVE)[0].min() >= 0


def test_get_regression_target_follows_the_reference():
    """reference samples.py:152-157: dt - gt[instance_id], and the error when boxes are unlabelled."""
    from waldboost_amd.boxes import Boxes
    from waldboost_amd.samples import get_regression_target
    dt = Boxes(np.array([[10, 10, 30, 30], [50, 52, 70, 75], [0, 0, 5, 5]], np.float32))
    gt = Boxes(np.array([[12, 9, 33, 31], [48, 50, 72, 74]], np.float32))
    with pytest.raises(ValueError):
        get_regression_target(dt, gt)
    dt.set_field("instance_id", np.array([0, 1, -1], np.int32))      # -1 indexes the last box, as NumPy does upstream
    get_regression_target(dt, gt)
    want = dt.get() - gt.get()[[0, 1, -1]]
    assert np.array_equal(dt.get_field("regression_target"), want)


def test_model_copies_and_pickles_without_its_device_state():
    """The reference's Model is a plain Python object (model.py:36-67): copy.deepcopy and pickle work on it."""
    import copy
    import pickle
    M = wb.load(os.path.join(GOLDEN, "models", "cfg2_d2_T128.pb"))
    M._lanes = {"not": "picklable state would live here"}
    for N in (copy.deepcopy(M), pickle.loads(pickle.dumps(M))):
        assert len(N) == len(M) and tuple(N.shape) == tuple(M.shape) and "_lanes" not in N.__dict__ and N._device is None
        for (w, t), (v, u) in zip(M, N):
            assert t == u and all(np.array_equal(getattr(w, k), getattr(v, k)) for k in ("feature", "threshold", "left", "right", "prediction"))


def test_tile_lists_dispatch_the_short_workgroups_last_on_every_xcd():
    """plan._tiles: the same tiles as the natural order; on every XCD (dispatch slots x, x + 8, ...) the tiles known to be
    short -- the channel kernel's identity levels and bottom-cut tiles, the cascade's edge-cut tiles -- come behind all
    others, costliest first; a plan with short_last off (large batches, WB_TILE_ORDER=natural) keeps the plain order."""
    for args in [(1080, 1920, 2, 8, 1), (300, 500, 2, 4, 1), (64, 64, 2, 8, 1)]:
        p, q = PyramidPlan(*args), PyramidPlan(*args)
        q.batch_hint = 64                                     # (a large batch keeps the plain order when the tile counts rotate the XCDs)
        assert p._short_last(3421) and not q._short_last(3421) and q._short_last(19224) and q._short_last(12)
        q._short_last = lambda n: False
        tc, tc0 = p.chan_tiles(), q.chan_tiles()
        tk, tk0 = p.casc_tiles(12, 12, 32, 64), q.casc_tiles(12, 12, 32, 64)
        assert sorted(map(tuple, tc.tolist())) == sorted(map(tuple, tc0.tolist()))
        assert sorted(map(tuple, tk.tolist())) == sorted(map(tuple, tk0.tolist()))
        grid = p.window_grid(12, 12)
        full = np.array([min(32, grid[t["level"]][0] - t["ty"] * 32) == 32 and min(64, grid[t["level"]][1] - t["tx"] * 64) == 64 for t in tk])
        ident = np.array([p.levels[t["level"]]["h"] == p.levels[t["level"]]["nh"] and p.levels[t["level"]]["w"] == p.levels[t["level"]]["nw"]
                          for t in tc])
        for x in range(8):
            f = full[x::8]
            if f.any() and not f.all():
                assert f[:int(f.sum())].all()                 # every full cascade tile of this XCD in front of its cut ones
            i = ident[x::8]
            if i.any():
                assert not i[:int(np.argmax(i))].any() and i.sum() <= i.size - np.argmax(i)   # no identity tile before the first short one
