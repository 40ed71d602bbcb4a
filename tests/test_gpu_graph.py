"""Parity of the path bench.py times: hipGraph replay of octaves -> channels -> cascade over resident
batches, several graphs in flight on separate streams, images replaced between replays.

Every image of every replay is compared with the oracle (reference model.py:149-179 per image):
detections (level, r, c, score bits) and alive[level, stage], bit-exact.  Also BASELINE configs[2]
at its full size (64 x 1080p in one launch per kernel)."""
import os

import numpy as np
import pytest

import waldboost_amd as wb
from waldboost_amd import _native as nat
from waldboost_amd.synth import synth_image
from util import GOLDEN, oracle_detect

pytestmark = pytest.mark.gpu

MODELS = {"grad_hist": "cfg2_d2_T128.pb", "grad_hist_4_u1": "cfg2_gh4u1_d2_T128.pb"}


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def engine_results(e, stt, T):
    """Per image (level, r, c, score) in reference order + alive[level, stage] of the engine's last scan."""
    d = e.sorted_detections().cpu().numpy().view(nat.DET_DTYPE).reshape(-1)
    alive = stt["alive"][:, :, :T].cpu().numpy().astype(np.int64)
    assert int(e.detb.counts.max().item()) <= e.detb.cap
    return [(d[d["image"] == b], alive[b]) for b in range(e.batch)]


def assert_image_matches(got, ref):
    d, alive = got
    assert np.array_equal(alive, ref["alive"])
    assert np.array_equal(d["level"], ref["level"]) and np.array_equal(d["r"], ref["r"]) and np.array_equal(d["c"], ref["c"])
    assert np.array_equal(bits(d["score"]), bits(ref["scores"]))


@pytest.mark.parametrize("channels", sorted(MODELS))
@pytest.mark.parametrize("shape,B", [((1080, 1920), 1), ((300, 420), 3)])
def test_graph_replay_on_two_streams_with_new_images_vs_oracle(channels, shape, B):
    import torch
    from waldboost_amd.engine import PyramidEngine
    M = wb.load(os.path.join(GOLDEN, "models", MODELS[channels]))
    spec = wb.channels.channel_spec(M.channel_opts["channels"])
    dm = M.device_cascade()
    H, W = shape
    P = 2
    engines = [PyramidEngine(H, W, np.uint8, 2, 8, 1, batch=B, det_capacity=16384 * B, channels=spec) for _ in range(P)]
    seed = lambda rnd, i, b: 5000 + 100 * rnd + 10 * i + b
    for i, e in enumerate(engines):
        e.load_images(np.stack([synth_image(H, W, seed(0, i, b)) for b in range(B)]))
    graphs = [e.capture(dm) for e in engines]
    lanes = [torch.cuda.Stream() for _ in range(P)]
    oracle = {}
    for rnd in range(2):
        if rnd:                                     # new images into the resident buffers, then replay the same graphs
            for i, e in enumerate(engines):
                e.load_images(np.stack([synth_image(H, W, seed(rnd, i, b)) for b in range(B)]))
        torch.cuda.synchronize()
        for _ in range(3):                          # replays overlap across the two streams, as in bench.py
            for i in range(P):
                with torch.cuda.stream(lanes[i]):
                    graphs[i].replay()
        torch.cuda.synchronize()
        for i, e in enumerate(engines):
            res = engine_results(e, e._casc_state(dm), len(M))
            for b in range(B):
                s = seed(rnd, i, b)
                if s not in oracle:
                    oracle[s] = oracle_detect(M, synth_image(H, W, s))
                assert_image_matches(res[b], oracle[s])
    assert sum(o["scores"].size for o in oracle.values()) > 0


def test_config3_batch_of_64_1080p_images():
    """BASELINE configs[2] at full size: 64 x 1920x1080 in one launch per kernel.  Images 0, 31 and 63
    against the oracle, the rest through the batch properties (sorted output, statistics add up)."""
    M = wb.load(os.path.join(GOLDEN, "models", "cfg2_d2_T128.pb"))
    B = 64
    imgs = np.stack([synth_image(1080, 1920, 2000 + b) for b in range(B)])
    res = M.detect_batch_raw(imgs)
    assert M.n_loc == B * 3045278
    key = np.stack([res["image"], res["level"], res["r"], res["c"]], 1).astype(np.int64)
    assert np.array_equal(np.lexsort((key[:, 3], key[:, 2], key[:, 1], key[:, 0])), np.arange(key.shape[0]))
    alive = res["alive"]
    assert alive[:, :, 0].sum() == M.n_loc and alive.sum() == M.n_weak and (np.diff(alive, axis=2) <= 0).all()
    assert np.bincount(res["image"], minlength=B).min() > 0
    for b in (0, 31, 63):
        ref = oracle_detect(M, imgs[b])
        sel = res["image"] == b
        assert np.array_equal(alive[b], ref["alive"])
        assert np.array_equal(res["level"][sel], ref["level"]) and np.array_equal(res["r"][sel], ref["r"])
        assert np.array_equal(res["c"][sel], ref["c"]) and np.array_equal(bits(res["scores"][sel]), bits(ref["scores"]))
        assert np.array_equal(bits(res["boxes"][sel]), bits(ref["boxes"]))


@pytest.mark.parametrize("channels", sorted(MODELS))
def test_model_detect_replays_one_graph_per_cascade_vs_oracle(channels):
    """Model.detect runs eagerly once and replays ONE captured graph (memset .. read-back copies) from the second call
    on: every call -- eager, replayed, replayed again after the detection buffer had to grow, and with a second
    cascade taking over the engine in between -- must give the oracle's boxes, scores, order and statistics."""
    from waldboost_amd import engine as E
    M = wb.load(os.path.join(GOLDEN, "models", MODELS[channels]))
    H, W = 300, 420
    E._ENGINES.clear()
    imgs = [synth_image(H, W, 7100 + i) for i in range(4)]
    refs = [oracle_detect(M, im) for im in imgs]

    def check(model, im, ref):
        before = (model.n_loc, model.n_weak)
        res = model.detect_raw(im)
        assert np.array_equal(res["alive"], ref["alive"])
        assert np.array_equal(res["level"], ref["level"]) and np.array_equal(res["r"], ref["r"]) and np.array_equal(res["c"], ref["c"])
        assert np.array_equal(bits(res["scores"]), bits(ref["scores"])) and np.array_equal(bits(res["boxes"]), bits(ref["boxes"]))
        assert model.n_weak - before[1] == int(ref["alive"].sum()) and model.n_loc > before[0]
        b = model.detect(im)
        assert np.array_equal(bits(b.get()), bits(ref["boxes"])) and np.array_equal(bits(b.get_field("scores")), bits(ref["scores"]))

    for im, ref in zip(imgs, refs):              # call 1 eager, the rest replays (each check() makes two calls)
        check(M, im, ref)
    eng = next(iter(E._ENGINES.values()))
    stt = eng._casc_state(M.device_cascade())
    assert stt.get("graph") is not None and stt["detect_calls"] >= 8
    # a detection buffer too small for the image: the replay reports the overflow, the buffer grows, the graph is dropped
    # and captured again against the new buffer
    assert refs[0]["scores"].size > 64
    eng.det_capacity = 64
    eng._alloc_det()
    assert "graph" not in stt
    for im, ref in zip(imgs, refs):
        check(M, im, ref)
    assert eng.detb.cap * nat.WB_DET_SHARDS > 64 and stt.get("graph") is not None
    # another cascade on the same engine (one cascade resident per engine): the first one's graph goes with its state
    M2 = wb.load(os.path.join(GOLDEN, "models", MODELS[channels]))
    M2.theta = [t - 0.25 if np.isfinite(t) else t for t in M2.theta]
    ref2 = oracle_detect(M2, imgs[1])
    check(M2, imgs[1], ref2)
    check(M, imgs[2], refs[2])
    check(M2, imgs[3], oracle_detect(M2, imgs[3]))


def test_model_detect_with_more_detections_than_the_read_back_prefix():
    """Model.detect's one-copy read-back holds _FETCH_ROWS records; a scan with more detections (here: a prefix shrunk to
    64 rows) falls back to the packed-record read-back -- eager and from the replayed graph alike -- with the same result."""
    from waldboost_amd import engine as E
    M = wb.load(os.path.join(GOLDEN, "models", MODELS["grad_hist"]))
    img = synth_image(300, 420, 7301)
    ref = oracle_detect(M, img)
    assert ref["scores"].size > 64
    E._ENGINES.clear()
    old = E.PyramidEngine._FETCH_ROWS
    E.PyramidEngine._FETCH_ROWS = 64
    try:
        for _ in range(3):                               # eager, capture + replay, replay
            res = M.detect_raw(img)
            assert np.array_equal(res["level"], ref["level"]) and np.array_equal(res["r"], ref["r"]) and np.array_equal(res["c"], ref["c"])
            assert np.array_equal(bits(res["scores"]), bits(ref["scores"])) and np.array_equal(bits(res["boxes"]), bits(ref["boxes"]))
            assert np.array_equal(res["alive"], ref["alive"])
    finally:
        E.PyramidEngine._FETCH_ROWS = old
        E._ENGINES.clear()


def test_captured_step_refuses_to_replay_after_the_engine_reallocated():
    """A captured graph addresses the control block and the detection buffer; once either has been re-allocated
    (a grown detection buffer here) the replay must raise instead of adding into freed memory."""
    import torch
    from waldboost_amd.engine import PyramidEngine
    M = wb.load(os.path.join(GOLDEN, "mixed_d2_T24.pb"))
    dm = M.device_cascade()
    H, W = 240, 320
    e = PyramidEngine(H, W, np.uint8, 2, 8, 1, batch=1, det_capacity=64)       # one record per shard: overflows
    img = synth_image(H, W, 3)
    e.load_images(img)
    g = e.capture(dm)
    g.replay()
    torch.cuda.synchronize()
    ref = oracle_detect(M, img)
    assert ref["scores"].size > 64
    n = e.ensure_capacity(dm)                                                   # grows the buffer, scans again
    assert n == ref["scores"].size
    with pytest.raises(RuntimeError, match="stale"):
        g.replay()
    g2 = e.capture(dm)
    g2.replay()
    torch.cuda.synchronize()
    assert_image_matches(engine_results(e, e._casc_state(dm), len(M))[0], ref)


def test_memset_free_steps_survive_other_users_of_the_engine():
    """A step holds no memset: the octave kernel resets what the cascade accumulates into, the cascade resets the
    octaves' (min, max) keys for the next step.  Anything else that runs the octaves on the same engine in between -- a
    pyramid for a caller, Model.detect's cached engine handed to channel_pyramid -- leaves dirty keys behind; the next
    replay must notice (a stale maximum would widen the resize's clip range silently)."""
    import torch
    from waldboost_amd.engine import PyramidEngine
    M = wb.load(os.path.join(GOLDEN, "mixed_d2_T24.pb"))
    dm = M.device_cascade()
    H, W = 200, 264
    bright, dark = synth_image(H, W, 31), (synth_image(H, W, 32) // 3).astype(np.uint8)     # different (min, max)
    assert bright.max() > dark.max() + 50
    e = PyramidEngine(H, W, np.uint8, 2, 8, 1, batch=1)
    e.load_images(bright)
    g = e.capture(dm)
    g.replay()
    torch.cuda.synchronize()
    assert_image_matches(engine_results(e, e._casc_state(dm), len(M))[0], oracle_detect(M, bright))
    e.run_channels()                               # someone else's pyramid of the BRIGHT image: keys left dirty
    e.load_images(dark)
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    assert_image_matches(engine_results(e, e._casc_state(dm), len(M))[0], oracle_detect(M, dark))
    # the same through the API: Model.detect's replayed graph and channel_pyramid share the cached engine
    for _ in range(3):
        M.detect(bright)
    lv = list(wb.channels.channel_pyramid(bright, M.channel_opts))
    ref = oracle_detect(M, dark)
    res = M.detect_raw(dark)
    assert np.array_equal(res["alive"], ref["alive"]) and np.array_equal(bits(res["scores"]), bits(ref["scores"]))
    assert len(lv) == ref["alive"].shape[0]


def test_capture_is_not_disturbed_by_the_cyclic_collector():
    """A CUDAGraph that Python's cyclic collector frees in the middle of a stream capture aborts it ("operation not
    permitted when stream is capturing", thrown from the destructor: tools/gc_capture_probe.py) -- and an engine that
    keeps a captured step used to be such a cycle.  Every capture of the package runs inside engine.capturing: collect
    first, collector paused until the capture has ended.  Here: a cycle holding a graph becomes garbage INSIDE a capture
    while the collector is set to fire on every allocation."""
    import gc
    import torch
    from waldboost_amd.engine import PyramidEngine, capturing
    M = wb.load(os.path.join(GOLDEN, "models", MODELS["grad_hist"]))
    dm = M.device_cascade()
    H, W = 200, 260
    e = PyramidEngine(H, W, np.uint8, 2, 8, 1, batch=1, det_capacity=16384)
    img = synth_image(H, W, 4242)
    e.load_images(img[None])
    e.run(dm)

    class Node:
        pass

    a, b = Node(), Node()
    a.other, b.other = b, a                              # a cycle only the collector frees ...
    b.graph = torch.cuda.CUDAGraph()                     # ... holding a captured graph
    with capturing(b.graph):
        e.run(dm)
    torch.cuda.synchronize()
    old = gc.get_threshold()
    g = torch.cuda.CUDAGraph()
    try:
        gc.set_threshold(1, 1, 1)
        with capturing(g):
            e.run(dm)
            del a, b                                     # garbage now; the collector would take it at the next allocation
            junk = [[i] for i in range(2000)]
            e.run(dm)
        del junk
    finally:
        gc.set_threshold(*old)
    gc.collect()
    g.replay()
    torch.cuda.synchronize()
    assert_image_matches(engine_results(e, e._casc_state(dm), len(M))[0], oracle_detect(M, img))
    # and an engine with a captured step of its own is freed by reference counting alone
    import weakref
    e2 = PyramidEngine(H, W, np.uint8, 2, 8, 1, batch=1, det_capacity=16384)
    e2.load_images(img[None])
    e2.batch_enqueue(dm)
    e2.batch_enqueue(dm)                                 # (the second call captures and keeps the step)
    torch.cuda.synchronize()
    gone = weakref.ref(e2)
    gc.disable()
    try:
        del e2
        assert gone() is None
    finally:
        gc.enable()


def test_every_user_of_an_engines_octaves_keeps_the_clean_keys_promise(monkeypatch):
    """The fused step holds no memset: the octaves' (min, max) keys must be zero when it starts, and a host-side flag says
    whether they are.  With WB_CHECK_KEYS the engine reads the keys back before every fused step and raises if the flag
    lies.  One engine, every path that touches its octaves interleaved: Model.detect (eager, then replaying its graph),
    channel_pyramid for a caller, the pyramid around a caller's own channel function (bare octave launch + resize_level),
    the sample miner's scan, and a two-model detect -- the detections in between must stay the oracle's."""
    from waldboost_amd import engine as E, samples
    monkeypatch.setattr(E, "_CHECK_KEYS", True)
    M = wb.load(os.path.join(GOLDEN, "models", MODELS["grad_hist"]))
    M2 = wb.load(os.path.join(GOLDEN, "models", MODELS["grad_hist"]))
    M2.theta = [t - 0.25 if np.isfinite(t) else t for t in M2.theta]
    H, W = 260, 340
    E._ENGINES.clear()
    imgs = [synth_image(H, W, 7300 + i) for i in range(3)]
    refs = [oracle_detect(M, im) for im in imgs]

    def check(k):
        res = M.detect_raw(imgs[k])
        assert np.array_equal(res["alive"], refs[k]["alive"]) and np.array_equal(bits(res["scores"]), bits(refs[k]["scores"]))
        assert np.array_equal(res["r"], refs[k]["r"]) and np.array_equal(res["c"], refs[k]["c"])

    check(0); check(1)                                        # eager, then the captured graph
    opts = dict(M.channel_opts)
    levels = [c for c, _ in wb.channels.channel_pyramid(imgs[2], opts)]          # the same engine, for a caller
    assert len(levels) > 4
    check(2); check(0)

    def my_channels(im):                                      # a channel function this build has no kernel for
        return np.stack([im, 255 - im], -1).astype(np.uint8)
    n = sum(1 for _ in wb.channels.channel_pyramid(imgs[1], dict(opts, channels=my_channels)))
    assert n == len(levels)
    check(1)
    gt = wb.Boxes(np.array([[40, 40, 120, 120]], "f"))
    mined = list(samples.get_samples_from_image(M, imgs[0], gt, max_tp_candidates=10 ** 6, max_fp_candidates=10 ** 6))   # the miner's scan
    assert isinstance(mined, list)
    check(2)
    both = wb.detect(imgs[0], M, M2)                          # two cascades on one pyramid (its own captured step)
    assert both.get().shape[1] == 4 and set(np.unique(both.get_field("label"))) <= {0, 1}
    wb.detect(imgs[1], M, M2)                                 # (the multi-model step replays its own graph from the second call on)
    check(0); check(1)
    assert len(E._ENGINES) >= 1
