"""Model.detect_stream: the reference's detection loop (scripts/waldboost-detect.py:64-67 -- model.detect per image)
pipelined over several engines and streams.  Every image's Boxes must be what Model.detect returns for it (which
test_gpu_parity pins to the oracle), bit for bit and in the iterable's order; n_loc / n_weak advance the same way."""
import os

import numpy as np
import pytest

import waldboost_amd as wb
from waldboost_amd.synth import synth_image
from util import GOLDEN, oracle_detect

pytestmark = pytest.mark.gpu


def load(name="cfg2_d2_T128.pb"):
    return wb.load(os.path.join(GOLDEN, "models", name))


def mixed_images():
    """Shapes and dtypes change along the sequence; one image is smaller than the window (no levels)."""
    out = [synth_image(240, 320, 100 + i) for i in range(7)]
    out += [synth_image(200, 300, 200 + i).astype(np.float32) for i in range(2)]
    out += [np.zeros((8, 8), np.uint8)]
    out += [synth_image(240, 320, 300 + i) for i in range(4)]
    out += [synth_image(150, 420, 400), synth_image(240, 320, 401), synth_image(150, 420, 402)]
    return out


def same_boxes(a, b):
    return (np.array_equal(a.get().view(np.uint32), b.get().view(np.uint32))
            and np.array_equal(a.get_field("scores").view(np.uint32), b.get_field("scores").view(np.uint32)))


@pytest.mark.parametrize("lanes", [1, 2, 3, 5])
def test_stream_equals_detect_per_image(lanes):
    images = mixed_images() * 2            # (the second pass finds every lane's graph captured)
    A, B = load(), load()
    ref = [A.detect(im) for im in images]
    got = list(B.detect_stream(iter(images), lanes=lanes))
    assert len(got) == len(ref)
    assert sum(len(r) for r in ref) > 0
    for i, (g, r) in enumerate(zip(got, ref)):
        assert same_boxes(g, r), f"image {i}"
    assert (A.n_loc, A.n_weak) == (B.n_loc, B.n_weak)


def test_stream_first_image_against_oracle():
    M = load()
    im = synth_image(240, 320, 77)
    ref = oracle_detect(M, im)
    got = list(M.detect_stream([im, im, im, im, im]))
    for g in got:                      # eager call, captured call, replays
        assert np.array_equal(g.get_field("scores").view(np.uint32), ref["scores"].view(np.uint32))
        assert np.array_equal(g.get(), ref["boxes"])


def test_stream_is_lazy_and_survives_an_early_stop():
    M = load()
    taken = []

    def source():
        for i in range(20):
            taken.append(i)
            yield synth_image(240, 320, 500 + i)

    gen = M.detect_stream(source(), lanes=3)
    first = next(gen)
    assert len(taken) <= 3                       # at most lanes - 1 images ahead of the one handed out
    second = next(gen)
    gen.close()
    assert len(taken) <= 5
    N = load()
    assert same_boxes(first, N.detect(synth_image(240, 320, 500))) and same_boxes(second, N.detect(synth_image(240, 320, 501)))
    # the lanes are reusable afterwards
    again = list(M.detect_stream([synth_image(240, 320, 500)]))
    assert same_boxes(again[0], first)


def test_stream_with_a_model_that_grows_between_images():
    """The cascade is snapshotted per image: stages appended while earlier images are still in flight do not touch them."""
    M, N = load(), load()
    ims = [synth_image(240, 320, 600 + i) for i in range(6)]
    full = list(M.classifier), list(M.theta)
    k = len(M) // 2

    def cut(model):
        model.classifier, model.theta = list(full[0][:k]), list(full[1][:k])

    def grow(model):
        model.classifier, model.theta = list(full[0]), list(full[1])

    cut(M)
    cut(N)

    def source():
        for i, im in enumerate(ims):
            if i == 3:
                grow(M)
            yield im

    got = list(M.detect_stream(source(), lanes=3))
    ref = []
    for i, im in enumerate(ims):
        if i == 3:
            grow(N)
        ref.append(N.detect(im))
    for i, (g, r) in enumerate(zip(got, ref)):
        assert same_boxes(g, r), f"image {i}"


def test_stream_hands_an_upload_error_to_the_consumer():
    M = load()
    good = [synth_image(240, 320, 700 + i) for i in range(8)]
    bad = np.zeros((240, 320), np.complex64)
    gen = M.detect_stream(good + [bad] + good[:2], lanes=3)
    with pytest.raises((TypeError, NotImplementedError, ValueError)):
        list(gen)
    # ... and a later stream on the same lanes is unharmed
    N = load()
    assert same_boxes(list(M.detect_stream(good[:1]))[0], N.detect(good[0]))


def test_stream_grows_a_lane_whose_detection_buffer_overflows():
    from waldboost_amd import _native as nat
    M, N = load(), load()
    ims = [synth_image(240, 320, 800 + i) for i in range(9)]
    ref = [N.detect(im) for im in ims]
    assert max(len(r) for r in ref) > 16
    list(M.detect_stream(ims, lanes=3))                     # lanes built, graphs captured
    for group in M._lanes.values():
        for eng, _ in group:
            eng.det_capacity = 16 * nat.WB_DET_SHARDS       # far too small: every shard overflows
            eng._alloc_det()
    got = list(M.detect_stream(ims, lanes=3))
    for i, (g, r) in enumerate(zip(got, ref)):
        assert same_boxes(g, r), f"image {i}"


@pytest.mark.parametrize("lanes,batch", [(3, 4), (2, 3), (1, 2), (3, 16)])
@pytest.mark.parametrize("post", ["ordered", "host", "device"])
def test_stream_in_batches_equals_detect_per_image(lanes, batch, post, monkeypatch):
    """Batches fill image by image; a shape change, the tiny image and the end of the sequence send partly filled ones
    (whose stale slots must not show).  post: where a batch's results are put in order -- "ordered": split by image and
    ordered by wb_det_order_batch_launch (the default); the forms it falls back to when an image holds more than 4096
    detections: "host" (few detections: from the packed read-back) and "device" (torch sort + wb_boxes_launch)."""
    import waldboost_amd.model as wm
    if post != "ordered":
        monkeypatch.setattr(wm, "_ORDER_BATCH", False)
    if post == "device":
        monkeypatch.setattr(wm, "_HOST_POST_BATCH", 0)
    images = mixed_images() * 2
    A, B = load(), load()
    ref = [A.detect(im) for im in images]
    got = list(B.detect_stream(iter(images), lanes=lanes, batch=batch))
    assert len(got) == len(ref)
    for i, (g, r) in enumerate(zip(got, ref)):
        assert same_boxes(g, r), f"image {i}"
    assert (A.n_loc, A.n_weak) == (B.n_loc, B.n_weak)


def test_stream_in_batches_grows_an_overflowing_detection_buffer():
    from waldboost_amd import _native as nat
    M, N = load(), load()
    ims = [synth_image(240, 320, 900 + i) for i in range(10)]
    ref = [N.detect(im) for im in ims]
    list(M.detect_stream(ims, lanes=2, batch=4))
    for group in M._lanes.values():
        for eng, _ in group:
            eng.det_capacity = 16 * nat.WB_DET_SHARDS
            eng._alloc_det()
    got = list(M.detect_stream(ims, lanes=2, batch=4))
    for i, (g, r) in enumerate(zip(got, ref)):
        assert same_boxes(g, r), f"image {i}"


def test_detect_on_images_through_the_stream():
    """testing.detect_on_images (reference testing.py:127-132) with lanes / batch: the same tuples as its plain loop."""
    from waldboost_amd.testing import detect_on_images
    from waldboost_amd.boxes import Boxes
    M, N = load(), load()
    ims = mixed_images()
    dicts = [{"image": im, "groundtruth_boxes": Boxes(np.array([[1.0, 2.0, 30.0 + i, 40.0]]))} if i % 2 else {"image": im}
             for i, im in enumerate(ims)]
    ref = list(detect_on_images(dicts, N))
    got = list(detect_on_images(iter(dicts), M, lanes=3, batch=4))
    assert len(ref) == len(got) == len(ims)
    for (g0, d0, s0), (g1, d1, s1) in zip(ref, got):
        assert s0 == s1 and np.array_equal(g0.get(), g1.get())
        assert same_boxes(d0, d1) and np.array_equal(d0.get_field("label"), d1.get_field("label"))


@pytest.mark.parametrize("batch", [1, 4])
def test_stream_with_more_detections_than_one_read_back_holds(batch, monkeypatch):
    """Model.detect's one-copy read-back holds _FETCH_ROWS detections; beyond that the records are fetched with further
    copies (and, in batches, ordered on the device) -- shrunk to 64 rows here, so that ordinary images exceed it."""
    from waldboost_amd import engine as E
    E._ENGINES.clear()                                  # (cached engines hold read-back buffers of the usual size)
    monkeypatch.setattr(E.PyramidEngine, "_FETCH_ROWS", 64)
    monkeypatch.setattr(E.PyramidEngine, "_ORDER_ROWS", 64)   # (a batch's per-image blocks likewise: the batch goes the older way)
    try:
        M, N = load(), load()
        ims = [synth_image(240, 320, 950 + i) for i in range(7)]
        ref = [N.detect(im) for im in ims]
        assert max(len(r) for r in ref) > 64
        got = list(M.detect_stream(ims, lanes=3, batch=batch))
        for i, (g, r) in enumerate(zip(got, ref)):
            assert same_boxes(g, r), f"image {i}"
        o = oracle_detect(N, ims[0])
        assert np.array_equal(ref[0].get(), o["boxes"]) and np.array_equal(ref[0].get_field("scores").view(np.uint32), o["scores"].view(np.uint32))
    finally:
        E._ENGINES.clear()


def test_detect_takes_page_locked_arrays_and_torch_tensors():
    """An extension of the reference's host-ndarray input (model.py:149): page-locked arrays are uploaded asynchronously,
    torch tensors (host, page-locked, device) are taken as they are -- same Boxes as from a plain ndarray."""
    import torch
    M = wb.load(os.path.join(GOLDEN, "mixed_d2_T24.pb"))
    img = synth_image(240, 320, 5)
    ref = M.detect(img)
    pinned_t = torch.from_numpy(img).pin_memory()
    for src in (pinned_t.numpy(), pinned_t, torch.from_numpy(img), torch.from_numpy(img).cuda()):
        for _ in range(3):                                   # (eager, captured, replayed)
            got = M.detect(src)
            assert np.array_equal(got.get(), ref.get()) and np.array_equal(got.get_field("scores"), ref.get_field("scores"))
    with pytest.raises(TypeError):
        M.detect([[0, 1], [2, 3]])
    with pytest.raises(ValueError):
        M.detect(torch.zeros((2, 3, 4), dtype=torch.uint8))
    f32 = torch.from_numpy(synth_image(240, 320, 6, np.float32)).cuda()
    a, b = M.detect(f32), M.detect(f32.cpu().numpy())
    assert np.array_equal(a.get(), b.get()) and len(a) == len(b)


@pytest.mark.parametrize("batch", [1, 4])
def test_detect_stream_from_one_reused_page_locked_buffer(batch):
    """A producer that decodes every frame into the SAME page-locked buffer: the asynchronous upload must have left the
    buffer before the stream asks for the next frame."""
    import torch
    M = wb.load(os.path.join(GOLDEN, "mixed_d2_T24.pb"))
    frames = [synth_image(200, 264, 40 + i) for i in range(11)]
    ref = [M.detect(f) for f in frames]
    buf = torch.empty((200, 264), dtype=torch.uint8).pin_memory()
    view = buf.numpy()

    def producer():
        for f in frames:
            view[...] = f                                    # overwrites what the previous frame was uploaded from
            yield view

    out = list(M.detect_stream(producer(), lanes=3, batch=batch))
    assert len(out) == len(ref)
    for a, b in zip(out, ref):
        assert np.array_equal(a.get(), b.get()) and np.array_equal(a.get_field("scores"), b.get_field("scores"))
