"""Detection model -- drop-in for ``waldboost.model.Model`` on the detection path
(reference model.py:32-344): same constructor, ``append``/iteration, ``detect``,
``predict_on_image``, ``channels``, ``scan_channels``, ``get_boxes``, ``eval_cost``/``reset``
and the zlib+protobuf ``.pb`` format.  The dense cascade scan runs in csrc/wb_cascade.hip.
"""
import os
import zlib

import numpy as np
from google.protobuf.message import DecodeError

from . import _native as nat
from . import channels as _channels
from . import engine as _engine
from . import model_pb2
from .boxes import Boxes, concatenate
from .channels import channel_pyramid
from .compare import channel_tensor
from .training import DTree, _REBINDS

# above this many detections per call the compaction, ordering and boxes stay on the GPU
_HOST_POST_MAX = 1 << 16
_HOST_POST_BATCH = 2048          # detect_stream's batches: more detections than this are ordered on the device
# detect_stream's batches: split by image and ordered by wb_det_order_batch_launch (tests and WB_NO_ORDER_BATCH switch it off)
_ORDER_BATCH = os.environ.get("WB_NO_ORDER_BATCH") is None


def symbol_name(s):
    """Name written into .pb files: the reference's own module.qualname for the channel functions
    this build implements, so that the reference can load our files too (model.py:23-24, 302)."""
    spec = _channels.channel_spec(s)
    if spec is not None:
        return spec.reference_name
    return s.__module__ + "." + s.__qualname__


def symbol_from_name(name: str):
    """Allow-list replacement of the reference's eval-based lookup (model.py:27-29)."""
    try:
        return _channels.CHANNEL_FUNCS[name]
    except KeyError:
        raise ValueError(f"unknown channel function {name!r} in model file "
                         f"(known: {sorted(_channels.CHANNEL_FUNCS)})") from None


class Model:
    def __init__(self, shape, channel_opts):
        self.shape = shape
        self.channel_opts = channel_opts
        self.classifier = []
        self.theta = []
        self._device = None
        self._content = None
        self.reset()

    def __getstate__(self):
        """copy / pickle carry the model (the reference's Model is a plain object), not its device-side state."""
        d = dict(self.__dict__)
        d["_device"] = d["_content"] = None
        d.pop("_lanes", None)
        return d

    # ---- statistics (reference model.py:69-89)
    @property
    def eval_cost(self):
        return self.n_weak / self.n_loc if self.n_loc > 0 else 0

    def reset(self):
        self.n_loc = 0
        self.n_weak = 0

    # ---- container (reference model.py:91-92, 261-283)
    def __getitem__(self, i):
        return self.classifier[i], self.theta[i]

    def __len__(self):
        return len(self.classifier)

    def __bool__(self):
        return bool(self.classifier)

    def __iter__(self):
        yield from zip(self.classifier, self.theta)

    def append(self, weak, theta):
        self.classifier.append(weak)
        self.theta.append(theta)
        self._device = None

    def device_cascade(self):
        """The device-side cascade for the current stage list.  `classifier` and `theta` are plain public lists in the
        reference and callers edit them -- and the trees' arrays -- in place, so the cached handle is keyed by the lists'
        current CONTENT: the identity and the bytes of every tree (DTree.content: one private block per tree, joined in
        one call) and the value and kind of every theta.  The GPU holds a private copy; nothing of the caller's is frozen."""
        # (theta by value AND kind: the float kind matters to theta_as_f32; a NaN theta still matches itself -- tuple
        # comparison takes identical objects as equal)
        ids = tuple(map(id, self.classifier))
        cc = self._content
        if cc is None or cc[0] != ids or cc[1] != _REBINDS[0]:
            # (the trees' blocks as buffers, looked up once per stage list; a tree with a rebound array has no block and
            # is read array by array on every call)
            trees = list(self.classifier)                        # (keeps the ids alive and unique)
            views = [memoryview(w._blob) for w in trees] if all(w._blob is not None for w in trees) else None
            cc = self._content = (ids, _REBINDS[0], views, trees)
        content = b"".join(cc[2]) if cc[2] is not None else b"".join([bytes(w.content()) for w in cc[3]])
        sig = (tuple(self.shape), ids, tuple(self.theta), tuple(map(type, self.theta)), content)
        if self._device is None or self._device[0] != sig:
            dev = _engine.DeviceCascade(self.shape, self.classifier, self.theta)
            self._device = (sig, dev, list(self.classifier))     # (the list keeps the ids alive and unique)
        return self._device[1]

    # ---- pyramid (reference model.py:95-134)
    def channels(self, image):
        yield from channel_pyramid(image, self.channel_opts)

    def scan_channels(self, image):
        yield from ((chns, scale, self.predict_on_image(chns)) for chns, scale in self.channels(image))

    def get_boxes(self, r, c, scale) -> Boxes:
        """XYXY boxes of window origins (r, c) at pyramid scale `scale` (reference model.py:136-147)."""
        r = np.asarray(r)
        c = np.asarray(c)
        if r.size == 0:
            return Boxes(np.empty((0, 4), "f"))
        m, n = self.shape[:2]
        x1 = c.reshape(-1, 1)
        y1 = r.reshape(-1, 1)
        rects = np.concatenate([x1, y1, x1 + n, y1 + m], axis=1).astype(np.float32)
        return Boxes(rects).normalized(scale=1.0 / scale)

    # ---- the cascade on one channel image (reference model.py:216-259)
    def predict_on_image(self, X):
        """All windows of X[u,v,C] through the cascade -> (rs, cs, hs) of the survivors in
        row-major order; updates n_loc / n_weak."""
        rs, cs, hs, alive = self.predict_on_image_stats(X)
        return rs, cs, hs

    def predict_on_image_stats(self, X):
        """As predict_on_image, plus alive[t] = windows entering stage t."""
        import torch
        u, v, ch_image = X.shape
        m, n, ch_cls = self.shape
        assert ch_image == ch_cls, f"Invalid number of channels. Expected {ch_cls} given {ch_image}."
        dm = self.device_cascade()
        Xd, wdt = channel_tensor(X, nat.require_gpu())       # any dtype, compared as NumPy would (compare.py)
        eng = _SingleLevel.get(u, v, ch_image, np.uint8 if wdt == nat.WB_DTYPE_U8 else np.float32)
        eng.load(Xd)
        n_det, alive = eng.scan(dm)
        self.n_loc += max(u - m, 0) * max(v - n, 0)
        self.n_weak += int(alive.sum())
        det = eng.sorted(n_det)
        if n_det == 0:
            return np.empty(0, np.int64), np.empty(0, np.int64), np.empty(0, np.float32), alive
        d = det.cpu().numpy().view(nat.DET_DTYPE).reshape(-1)
        return d["r"].astype(np.int64), d["c"].astype(np.int64), d["score"].copy(), alive

    # ---- whole-image detection (reference model.py:149-179)
    def detect(self, image) -> Boxes:
        """Detect objects in a 2-D image; returns Boxes with a 'scores' field, levels in pyramid
        order and windows in row-major order within a level, like the reference."""
        res = self.detect_raw(image, _full=False)
        out = Boxes(res["boxes"])
        out.set_field("scores", res["scores"])
        return out

    def detect_raw(self, image, _full=True):
        """detect() with everything the parity tests compare: boxes, scores, (level, r, c),
        alive[level, stage]; updates n_loc / n_weak.  (_full=False: boxes and scores only -- what detect() returns.)"""
        _channels._validate_image(image, allow_tensor=True)
        shrink, n_per_oct, smooth, spec = _channels.read_opts(self.channel_opts, allow_callable=True)
        if spec is None:
            if not isinstance(image, np.ndarray):
                image = image.cpu().numpy()                 # (a caller's own channel function takes host arrays)
            return self._detect_raw_levelwise(image)
        m, n, Cc = self.shape
        assert Cc == spec.n_channels, f"Invalid number of channels. Expected {Cc} given {spec.n_channels}."
        H, W = (int(x) for x in image.shape)
        dm = self.device_cascade()
        eng = _engine.get_engine(H, W, _engine.array_dtype(image), shrink, n_per_oct, smooth, 1, channels=spec)
        T = len(self)
        if eng.plan.n_levels == 0:
            return dict(boxes=np.empty((0, 4), "f"), scores=np.empty(0, "f"), level=np.empty(0, np.int32),
                        r=np.empty(0, np.int64), c=np.empty(0, np.int64), alive=np.zeros((0, T), np.int64),
                        scales=[])
        eng.load_images(image)
        # one memset + octaves + channels (straight to this cascade's threshold ranks when it has rank tables) + cascade
        # + boxes and sort keys + the read-back: one hipGraph replay from the second call on
        fin = eng.detect_run(dm)
        return self._collect(eng, dm, eng._casc_state(dm), _full, fin)

    def _detect_raw_levelwise(self, image):
        """detect_raw for a channel function without a kernel: the reference's own loop (model.py:171-177) -- one level at
        a time from the generator (the caller's function runs between the GPU steps), each scanned on the GPU."""
        T = len(self)
        parts, alive, scales = [], [], []
        for lv, (chns, scale) in enumerate(self.channels(image)):
            rs, cs, hs, al = self.predict_on_image_stats(chns)
            alive.append(al)
            scales.append(scale)
            if rs.size:
                parts.append((np.full(rs.size, lv, np.int32), rs, cs, hs, self.get_boxes(rs, cs, scale).get()))
        cat = lambda k, dt: np.concatenate([p[k] for p in parts]).astype(dt, copy=False) if parts else np.empty(0, dt)
        return dict(boxes=np.concatenate([p[4] for p in parts]) if parts else np.empty((0, 4), "f"), scores=cat(3, np.float32),
                    level=cat(0, np.int32), r=cat(1, np.int64), c=cat(2, np.int64),
                    alive=np.stack(alive) if alive else np.zeros((0, T), np.int64), scales=scales)

    def scan_engine(self, eng, view=None):
        """Run this cascade over the channel pyramid already resident in `eng` (channels computed
        by the caller: several models can share one pyramid, reference __init__.py:120-124).
        view: this model's member view of a RankGroup whose threshold ranks `eng` holds (scan those instead of the
        float32 channels).  Returns the same dict as detect_raw and updates n_loc / n_weak."""
        m, n, Cc = self.shape
        assert Cc == eng.spec.n_channels, f"Invalid number of channels. Expected {Cc} given {eng.spec.n_channels}."
        dm = view if view is not None else self.device_cascade()
        return self._collect(eng, dm, eng.run_cascade(dm, ranks=view is not None))

    def _collect(self, eng, dm, stt, full=True, fin=False):
        """Results of the scan `stt` of image 0 of `eng`: the dict detect_raw returns; updates n_loc / n_weak.
        fin: what eng.detect_run returned for this scan (False: fetch it here)."""
        m, n, Cc = self.shape
        T = dm.n_stages                                   # (the cascade that was scanned: detect_stream collects late)
        if fin is False:
            fin = eng.fetch_final(dm, stt)                # ONE host synchronisation: sort keys, boxes, scores, statistics
        if fin is not None:
            # get_boxes and the (level, r, c) keys were formed on the device (wb_det_finish_sorted_launch) and -- up to 4096
            # detections -- put in the reference's order there: the host copies slices out of the read-back buffer.
            # Otherwise it sorts the keys -- unique, so any sort kind gives the reference order -- and gathers
            keys, boxes_d, scores_d, alive, ordered = fin
            alive = alive[0].reshape(eng.plan.n_levels, T)
            if "n_loc" not in stt:
                stt["n_loc"] = eng.plan.n_loc(m, n)
            self.n_loc += stt["n_loc"]
            self.n_weak += int(alive.sum())
            if ordered:
                ks = keys                                  # (the fields below are taken out as new arrays)
                res = dict(boxes=boxes_d[:keys.size].copy(), scores=scores_d[:keys.size].copy(), alive=alive, scales=list(eng.plan.scales))
            else:
                ks = np.sort(keys)
                at = (ks & np.uint64((1 << 26) - 1)).astype(np.intp)
                res = dict(boxes=boxes_d[at], scores=scores_d[at], alive=alive, scales=list(eng.plan.scales))
            if full:
                res.update(level=(ks >> np.uint64(54)).astype(np.int32),
                           r=((ks >> np.uint64(40)) & np.uint64(0x3fff)).astype(np.int64),
                           c=((ks >> np.uint64(26)) & np.uint64(0x3fff)).astype(np.int64))
            return res
        recs, alive = eng.fetch(dm, stt)                  # ONE host synchronisation: packed records + statistics
        alive = alive[0].reshape(eng.plan.n_levels, T)
        n_det = recs.shape[0]
        self.n_loc += eng.plan.n_loc(m, n)
        self.n_weak += int(alive.sum())
        if n_det <= _HOST_POST_MAX:
            # few detections (the usual case): order and boxes on the host -- the same float32 arithmetic as
            # boxes_kernel, without three launches and four copies
            d = recs.view(nat.DET_DTYPE).reshape(-1)
            level, r, c = d["level"].astype(np.int64), d["r"].astype(np.int64), d["c"].astype(np.int64)
            order = np.argsort((level << 32) | (r << 16) | c)        # (level, r, c): unique keys, any sort kind
            level, r, c, score = level[order], r[order], c[order], d["score"][order]
            inv = eng.inv_scales()[level] if n_det else np.zeros(0, "f")
            boxes = np.empty((n_det, 4), np.float32)
            np.multiply(c.astype(np.float32), inv, out=boxes[:, 0])
            np.multiply(r.astype(np.float32), inv, out=boxes[:, 1])
            np.multiply((c + n).astype(np.float32), inv, out=boxes[:, 2])
            np.multiply((r + m).astype(np.float32), inv, out=boxes[:, 3])
            return dict(boxes=boxes, scores=score, level=level.astype(np.int32), r=r, c=c, alive=alive,
                        scales=list(eng.plan.scales))
        det = eng.sorted_detections()
        boxes, scores = eng.boxes(det, dm)
        d = det.cpu().numpy().view(nat.DET_DTYPE).reshape(-1)
        return dict(boxes=boxes.cpu().numpy(), scores=scores.cpu().numpy(), level=d["level"].copy(),
                    r=d["r"].astype(np.int64), c=d["c"].astype(np.int64), alive=alive, scales=list(eng.plan.scales))

    def detect_stream(self, images, lanes=3, batch=1):
        """detect() over an iterable of 2-D images, as a generator of Boxes in the iterable's order -- the loop the
        reference's detection script runs (scripts/waldboost-detect.py:64-67), pipelined: `lanes` engines, each on its
        own stream, hold consecutive images, so image i + 1 is uploaded and image i - 1's detections are read back and
        ordered on the host while image i is scanned (a single detect() call is three quarters upload, waits and host
        work: DESIGN section 5).  Same results, same n_loc / n_weak updates as one detect() call per image.
        batch: consecutive images of one shape and dtype are gathered `batch` at a time and go through every kernel in
        one launch (a shape change or the end of the iterable sends a partly filled batch); ordering and boxes of a
        batch are computed on the device.  Up to lanes * batch - 1 images are taken from the iterable ahead of the one
        whose Boxes are being yielded.
        (Measured and left out: the blocking upload on a helper thread -- 0.13 to 0.16 ms per 1080p image against 0.14 to
        0.15 without: what the copy frees, the two threads lose again handing the interpreter lock back and forth.)"""
        import torch
        lanes, K = max(int(lanes), 1), max(int(batch), 1)
        shrink, n_per_oct, smooth, spec = _channels.read_opts(self.channel_opts, allow_callable=True)
        if spec is None:                      # (a caller's own channel function runs on the host between the GPU steps)
            for image in images:
                yield self.detect(image)
            return
        m, n, Cc = self.shape
        assert Cc == spec.n_channels, f"Invalid number of channels. Expected {Cc} given {spec.n_channels}."
        pool = self.__dict__.setdefault("_lanes", {})        # (H, W, dtype, channel_opts, batch) -> [(engine, stream)]
        pending = []                                          # [(engine, stream, dm, token, images in it)] oldest first
        fill = None                                           # the batch being gathered: [key, engine, stream, dm, images in it]

        def finish(item):
            eng, stream, dm, token, count = item
            if K == 1:
                fin = eng.detect_collect(dm, token, stream)
                if fin is None:                               # (more detections than the one read-back holds: further copies)
                    with torch.cuda.stream(stream):
                        res = self._collect(eng, dm, eng._casc_state(dm), False, None)
                else:
                    res = self._collect(eng, dm, eng._casc_state(dm), False, fin)
                out = Boxes(res["boxes"])
                out.set_field("scores", res["scores"])
                return [out]
            with torch.cuda.stream(stream):
                return self._collect_batch(eng, dm, token[0], count, enqueued=token[1])

        def send(f):
            _, eng, stream, dm, count = f
            with torch.cuda.stream(stream):
                stt = eng.batch_enqueue(dm)
                # the batch's results split by image, ordered and on their way to the host right behind the scan
                ordered = _ORDER_BATCH and eng.order_batch_enqueue(dm, stt)
                pending.append((eng, stream, dm, (stt, ordered), count))

        try:
            for image in images:
                _channels._validate_image(image, allow_tensor=True)
                H, W = (int(x) for x in image.shape)
                idt = _engine.array_dtype(image)
                key = (H, W, idt.str, shrink, n_per_oct, smooth, spec.key, K)
                dm = self.device_cascade()
                if fill is not None and (fill[0] != key or fill[3] is not dm):
                    send(fill)                                # (another shape, or the model changed: the batch goes as it is)
                    fill = None
                while len(pending) >= lanes:
                    yield from finish(pending.pop(0))
                if fill is None:
                    group = pool.get(key)
                    if group is None:
                        if len(pool) >= 2:
                            pool.pop(next(iter(pool)))
                        group = pool[key] = []
                    # the lane this image takes: one no pending image holds (fewer than `lanes` are pending here)
                    busy = {id(it[0]) for it in pending}
                    lane = next((ln for ln in group if id(ln[0]) not in busy), None)
                    if lane is None:
                        lane = (_engine.PyramidEngine(H, W, idt, shrink, n_per_oct, smooth, K, channels=spec),
                                torch.cuda.Stream())
                        group.append(lane)
                    eng, stream = lane
                    if eng.plan.n_levels == 0:
                        while pending:
                            yield from finish(pending.pop(0))
                        yield self.detect(image)
                        continue
                    fill = [key, eng, stream, dm, 0]
                _, eng, stream, _, count = fill
                with torch.cuda.stream(stream):
                    if K == 1:
                        eng.load_images(image)
                        pending.append((eng, stream, dm, eng.detect_enqueue(dm), 1))
                        fill = None
                    else:
                        eng.load_slot(count, image)
                        fill[4] = count + 1
                if fill is not None and fill[4] == K:
                    send(fill)
                    fill = None
                if len(pending) >= lanes:
                    yield from finish(pending.pop(0))
                # an image uploaded asynchronously from the caller's page-locked buffer: the copy has left that buffer
                # before the iterable is asked for its next image (a decoder may write into the same buffer again); by
                # now the scan is enqueued and an older lane has been collected, so this wait is over before it starts
                eng.wait_upload()
            if fill is not None:
                send(fill)
                fill = None
            while pending:
                yield from finish(pending.pop(0))
        finally:
            for item in pending:                              # (the consumer stopped early: let the lanes drain)
                item[1].synchronize()
            if fill is not None:
                fill[2].synchronize()

    def _collect_batch(self, eng, dm, stt, count, enqueued=False):
        """Boxes of images [0, count) of the batch `eng` has just scanned (detect_stream; the slots behind `count` hold
        earlier images, whose results are dropped).  Split by image, ordered and finished on the device
        (wb_det_order_batch_launch) with one read-back; an image with more than 4096 detections sends the batch the
        older way: few detections ordered on the host from the one read-back, otherwise ordered, and their boxes formed,
        on the device (wb_boxes_launch).  Updates n_loc / n_weak."""
        m, n, Cc = self.shape
        # split by image, ordered and finished on the device (wb_det_order_batch_launch: up to 4096 detections per
        # image): per image a copy of its slices of the one read-back
        # (enqueued: detect_stream has put the launch and the copies behind the scan already -- only the wait is left)
        res = eng.fetch_ordered_batch(dm, stt, enqueued) if _ORDER_BATCH else None   # ONE host synchronisation (overflow: grows and scans again)
        if res is not None:
            per_image, alive = res
            self.n_loc += count * eng.plan.n_loc(m, n)
            self.n_weak += int(alive[:count].sum())
            out = []
            for keys, boxes, scores in per_image[:count]:
                bx = Boxes(boxes.copy())
                bx.set_field("scores", scores.copy())
                out.append(bx)
            return out
        got, alive = eng.fetch(dm, stt, limit=_HOST_POST_BATCH)   # ONE host synchronisation (overflow: grows and scans again)
        self.n_loc += count * eng.plan.n_loc(m, n)
        self.n_weak += int(alive[:count].sum())
        if isinstance(got, np.ndarray):
            d = got.view(nat.DET_DTYPE).reshape(-1)
            d = d[d["image"] < count]
            image, level, r, c = (d[k].astype(np.int64) for k in ("image", "level", "r", "c"))
            order = np.argsort((image << 48) | (level << 32) | (r << 16) | c)          # unique keys, any sort kind
            image, level, r, c, scores = image[order], level[order], r[order], c[order], d["score"][order]
            inv = eng.inv_scales()[level] if d.size else np.zeros(0, "f")
            boxes = np.empty((d.size, 4), np.float32)
            np.multiply(c.astype(np.float32), inv, out=boxes[:, 0])
            np.multiply(r.astype(np.float32), inv, out=boxes[:, 1])
            np.multiply((c + n).astype(np.float32), inv, out=boxes[:, 2])
            np.multiply((r + m).astype(np.float32), inv, out=boxes[:, 3])
        else:
            det = _engine.sort_records(eng.packed[1:1 + got])
            boxes, scores = eng.boxes(det, dm)
            image = det[:, 0].cpu().numpy()
            boxes, scores = boxes.cpu().numpy(), scores.cpu().numpy()
        cuts = np.searchsorted(image, np.arange(count + 1))
        out = []
        for b in range(count):
            bx = Boxes(boxes[cuts[b]:cuts[b + 1]])
            bx.set_field("scores", scores[cuts[b]:cuts[b + 1]])
            out.append(bx)
        return out

    def detect_batch(self, images):
        """detect() on a batch: `images` is [B,H,W] (ndarray or device tensor) of one shape and dtype;
        the whole batch goes through each kernel in one launch.  Returns a list of B Boxes (same
        content and order as B calls of detect) and updates n_loc / n_weak."""
        res = self.detect_batch_raw(images)
        out = []
        for b in range(res["batch"]):
            sel = res["image"] == b
            bx = Boxes(res["boxes"][sel])
            bx.set_field("scores", res["scores"][sel])
            out.append(bx)
        return out

    def detect_batch_raw(self, images):
        """All detections of a batch as flat arrays (image, level, r, c, boxes, scores, alive[B,L,T])."""
        if getattr(images, "ndim", None) != 3 and (not hasattr(images, "dim") or images.dim() != 3):
            raise ValueError("images must have 3 dimensions [B,H,W]")
        B, H, W = (int(x) for x in images.shape)
        dtype = _engine.array_dtype(images)
        shrink, n_per_oct, smooth, spec = _channels.read_opts(self.channel_opts)
        m, n, Cc = self.shape
        assert Cc == spec.n_channels, f"Invalid number of channels. Expected {Cc} given {spec.n_channels}."
        dm = self.device_cascade()
        T = len(self)
        eng = _engine.get_engine(H, W, dtype, shrink, n_per_oct, smooth, B, channels=spec)
        L = eng.plan.n_levels
        if L == 0:
            return dict(batch=B, image=np.empty(0, np.int32), level=np.empty(0, np.int32), r=np.empty(0, np.int64),
                        c=np.empty(0, np.int64), boxes=np.empty((0, 4), "f"), scores=np.empty(0, "f"),
                        alive=np.zeros((B, 0, T), np.int64), scales=[])
        eng.load_images(images)
        stt = eng.run(dm)
        # split by image, ordered and finished on the device, one read-back (up to 4096 detections per image and 256
        # images: the read-back block is fixed-size); otherwise a device sort of all records and wb_boxes_launch
        res = eng.fetch_ordered_batch(dm, stt) if (_ORDER_BATCH and B <= 256) else None
        if res is not None:
            per_image, alive = res
            alive = alive.reshape(B, L, T)
            self.n_loc += B * eng.plan.n_loc(m, n)
            self.n_weak += int(alive.sum())
            keys = np.concatenate([k for k, _, _ in per_image]) if B else np.empty(0, np.uint64)
            return dict(batch=B, image=np.repeat(np.arange(B, dtype=np.int32), [k.size for k, _, _ in per_image]),
                        level=(keys >> np.uint64(54)).astype(np.int32), r=((keys >> np.uint64(40)) & np.uint64(0x3fff)).astype(np.int64),
                        c=((keys >> np.uint64(26)) & np.uint64(0x3fff)).astype(np.int64),
                        boxes=np.concatenate([b_ for _, b_, _ in per_image]), scores=np.concatenate([s_ for _, _, s_ in per_image]),
                        alive=alive, scales=list(eng.plan.scales))
        eng.ensure_capacity(dm)
        det = eng.sorted_detections()
        boxes, scores = eng.boxes(det, dm)
        alive = stt["alive"][:, :, :T].cpu().numpy().astype(np.int64).reshape(B, L, T)
        self.n_loc += B * eng.plan.n_loc(m, n)
        self.n_weak += int(alive.sum())
        d = det.cpu().numpy().view(nat.DET_DTYPE).reshape(-1)
        return dict(batch=B, image=d["image"].copy(), level=d["level"].copy(), r=d["r"].astype(np.int64),
                    c=d["c"].astype(np.int64), boxes=boxes.cpu().numpy(), scores=scores.cpu().numpy(), alive=alive,
                    scales=list(eng.plan.scales))

    def predict(self, X):
        """The cascade on samples X[N, m, n, C] -> (H, mask): H[i] is the response accumulated in stage
        order while sample i is alive and -inf once a stage rejected it, mask[i] whether it passed
        every stage (reference model.py:181-214; the training pool re-scores its samples with it)."""
        import torch
        n, *shape = X.shape
        assert tuple(shape) == tuple(self.shape), f"Invalid shape of X. Expected {self.shape}, given {shape}"
        if n == 0:
            return np.zeros(0, np.float32), np.ones(0, bool)
        dm = self.device_cascade()
        dev = nat.require_gpu()
        Xd, wdt = channel_tensor(X, dev)
        H = torch.empty(n, dtype=torch.float32, device=dev)
        mask = torch.empty(n, dtype=torch.uint8, device=dev)
        nat.check(nat.load().wb_samples_predict_launch(nat.stream_ptr(), dm.handle, nat.ptr(Xd), wdt, n, nat.ptr(H),
                                                       nat.ptr(mask)), "wb_samples_predict_launch")
        return H.cpu().numpy(), mask.cpu().numpy().astype(bool)

    # ---- wire format (reference model.py:285-344)
    def as_proto(self, proto):
        proto.Clear()
        proto.shape.extend(int(x) for x in self.shape)
        proto.channel_opts.shrink = self.channel_opts["shrink"]
        proto.channel_opts.n_per_oct = self.channel_opts["n_per_oct"]
        proto.channel_opts.smooth = self.channel_opts["smooth"]
        proto.channel_opts.func = symbol_name(self.channel_opts["channels"])
        for weak, theta in self:
            w_pb = proto.classifier.add()
            weak.as_proto(w_pb)
            proto.theta.append(float(theta))

    @staticmethod
    def from_proto(proto):
        shape = tuple(proto.shape)
        channel_opts = {
            "shrink": proto.channel_opts.shrink,
            "n_per_oct": proto.channel_opts.n_per_oct,
            "smooth": proto.channel_opts.smooth,
            "channels": symbol_from_name(proto.channel_opts.func),
        }
        M = Model(shape, channel_opts)
        for weak_proto, theta_proto in zip(proto.classifier, proto.theta):
            M.append(DTree.from_proto(weak_proto), theta_proto)
        return M

    def save(self, filename):
        proto = model_pb2.Model()
        self.as_proto(proto)
        data = zlib.compress(proto.SerializeToString(), 9)
        with open(filename, "wb") as f:
            f.write(data)

    @staticmethod
    def load(filename):
        with open(filename, "rb") as f:
            data = f.read()
        proto = model_pb2.Model()
        try:
            data = zlib.decompress(data)
            proto.ParseFromString(data)
        except (DecodeError, zlib.error):
            raise ValueError(f"Cannot read model from {filename}")
        return Model.from_proto(proto)


class _SingleLevel:
    """Device state for Model.predict_on_image on a caller-supplied HWC channel image."""
    _cache = {}

    @classmethod
    def get(cls, u, v, C, dtype=np.float32):
        import torch
        dtype = np.dtype(dtype)
        key = (u, v, C, dtype.str, torch.cuda.current_device() if torch.cuda.is_available() else -1)
        e = cls._cache.get(key)
        if e is None:
            if len(cls._cache) >= 8:
                cls._cache.pop(next(iter(cls._cache)))
            e = cls._cache[key] = cls(u, v, C, dtype)
        return e

    def __init__(self, u, v, C, dtype=np.float32):
        import torch
        self.lib = nat.load()
        self.dev = nat.require_gpu()
        if u >= 65536 or v >= 65536:
            raise ValueError("channel image larger than 65535 pixels per side")
        self.u, self.v, self.C = u, v, C
        t = np.zeros(1, nat.LEVEL_DTYPE)
        t[0]["u"], t[0]["v"], t[0]["chn_off"] = u, v, 0
        self.levels = torch.from_numpy(t.view(np.uint8).copy()).to(self.dev)
        self.dtype = np.dtype(dtype)
        self.tdtype = torch.uint8 if self.dtype == np.uint8 else torch.float32
        self.wb_dtype = nat.WB_DTYPE_U8 if self.dtype == np.uint8 else nat.WB_DTYPE_F32
        self.X = torch.empty((max(u * v * C, 1) + 16,), dtype=self.tdtype, device=self.dev)   # + spare bytes (see wb_cascade_launch)
        self.detb = _engine.DetBuffer(1 << 10, self.dev)
        self._tiles = {}

    def load(self, X):
        import torch
        if isinstance(X, torch.Tensor):
            self.X[:self.u * self.v * self.C].copy_(X.to(self.tdtype).reshape(-1))
        else:
            self.X[:self.u * self.v * self.C].copy_(torch.from_numpy(np.ascontiguousarray(X, self.dtype).reshape(-1)))

    def scan(self, dm):
        import torch
        from .plan import PyramidPlan
        key = (dm.m, dm.n, dm.tile_rows, dm.tile_cols, dm.n_stages)
        T = dm.n_stages
        if key not in self._tiles:
            tl = PyramidPlan._tiles([(max(self.u - dm.m, 0), max(self.v - dm.n, 0))], dm.tile_rows, dm.tile_cols)
            self._tiles = {key: (int(tl.size), torch.from_numpy(tl.view(np.uint8).copy()).to(self.dev) if tl.size else None)}
        n_tiles, tiles = self._tiles[key]
        alive = torch.empty((1, 1, max(T, 1)), dtype=torch.int32, device=self.dev)
        while True:
            self.detb.zero()
            alive.zero_()
            if n_tiles:
                nat.check(self.lib.wb_cascade_launch(nat.stream_ptr(), dm.handle, nat.ptr(self.X), self.wb_dtype, 0,
                                                     1, nat.ptr(self.levels), 1, nat.ptr(tiles), n_tiles,
                                                     nat.ptr(self.detb.recs), nat.ptr(self.detb.counts), self.detb.cap,
                                                     nat.ptr(alive)), "wb_cascade_launch")
            need = self.detb.max_count()
            if need <= self.detb.cap:
                break
            self.detb = _engine.DetBuffer(int(need * 1.5) + 16, self.dev)
        n = int(self.detb.counts.sum().item())
        return n, alive[0, 0, :T].cpu().numpy().astype(np.int64)

    def sorted(self, n):
        return _engine.sort_records(self.detb.compact())
