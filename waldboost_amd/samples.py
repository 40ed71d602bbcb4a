"""Samples from images -- drop-in for the detection-side part of ``waldboost.samples``: the training
pipeline's hard-negative mining runs the hot path over every training image, labels the surviving
windows against the ground truth and crops them out of the channel pyramid.

What runs where:

* ``gather_samples`` (reference samples.py:14-43) is a HIP copy kernel (``wb_gather_samples_launch``);
* ``get_samples_from_image`` (reference samples.py:160-216) keeps the channel pyramid on the GPU between
  the cascade scan and the crop -- only detections and the selected crops travel to the host;
* ``label_boxes`` / ``select_candidates`` / ``SampleLabel`` (reference samples.py:46-157) are small NumPy
  host steps with the reference's semantics (TP: best IoU above ``min_tp_iou`` on a non-ignored box; FP:
  best IoU below ``max_fp_iou``; at most ``max_*_candidates`` of each per call, drawn with
  ``np.random.choice``), ``bbx.iou`` replaced by ``boxes.iou``;
* ``SamplePool`` (reference samples.py:219-338) re-scores its samples with ``Model.predict`` on the GPU.
"""
import logging

import numpy as np

from . import _native as nat
from .boxes import Boxes, concatenate, iou


# --------------------------------------------------------------------------------------- crops
def _np_dtype(chns):
    return np.dtype(str(chns.dtype).replace("torch.", ""))


def gather_samples_device(chns, rs, cs, shape):
    """Crops of `shape` = (m, n, C) at the origins (rs[i], cs[i]) of the channel image `chns`
    (ndarray or device tensor, float32 / uint8) as a device tensor (N, m, n, C)."""
    import torch
    lib = nat.load()
    dev = nat.require_gpu()
    dt = _np_dtype(chns)
    if dt == np.uint8:
        tdt, wdt = torch.uint8, nat.WB_DTYPE_U8
    elif dt == np.float32:
        tdt, wdt = torch.float32, nat.WB_DTYPE_F32
    else:
        raise TypeError(f"channel image must be float32 or uint8 (as produced by channel_pyramid), got {dt}")
    u, v, C = (int(x) for x in chns.shape)
    m, n = int(shape[0]), int(shape[1])
    rows = np.asarray(rs).reshape(-1).astype(np.int64)
    cols = np.asarray(cs).reshape(-1).astype(np.int64)
    if rows.size and (rows.min() < 0 or cols.min() < 0 or rows.max() + m > u or cols.max() + n > v):
        # the reference slices without range checks and would build a ragged object array here
        raise IndexError("sample window outside the channel image")
    src = chns if isinstance(chns, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(chns))
    src = src.to(dev, tdt).contiguous()
    pos = torch.from_numpy(np.stack([rows, cols]).astype(np.int32)).to(dev)
    out = torch.empty((rows.size, m, n, C), dtype=tdt, device=dev)
    nat.check(lib.wb_gather_samples_launch(nat.stream_ptr(), nat.ptr(src), wdt, u, v, C, nat.ptr(pos[0]), nat.ptr(pos[1]),
                                           rows.size, m, n, nat.ptr(out)), "wb_gather_samples_launch")
    return out


def gather_samples(chns, rs, cs, shape) -> np.ndarray:
    """X[i] = chns[rs[i]:rs[i]+m, cs[i]:cs[i]+n, :] -> (N, m, n, C) in the dtype of `chns`."""
    n_r, n_c = np.asarray(rs).size, np.asarray(cs).size
    if n_r != n_c:
        raise ValueError("Sizes of 'rs' and 'cs' must match")
    if n_r == 0:
        return np.empty((0,) + tuple(shape), dtype=_np_dtype(chns))
    return gather_samples_device(chns, rs, cs, shape).cpu().numpy()


# --------------------------------------------------------------------------------------- labels
class SampleLabel:
    TRUE_POSITIVE = 1
    FALSE_POSITIVE = -1
    IGNORE = 0


def select_candidates(condition, max_candidates: int) -> np.ndarray:
    """Indices of True entries, thinned to `max_candidates` by ``np.random.choice`` when there are more."""
    hits = np.flatnonzero(condition)
    return hits if hits.size <= max_candidates else np.random.choice(hits, max_candidates)


def _match_ground_truth(dt_boxes, gt_boxes):
    """(best IoU, index of the best ground-truth box, its ignore flag) for every detection."""
    flags = gt_boxes.get_field("ignore") if gt_boxes.has_field("ignore") else np.zeros(len(gt_boxes))
    if flags.ndim != 1:
        raise ValueError("'ignore' field must be single dimension")
    table = iou(dt_boxes, gt_boxes)
    best = table.argmax(axis=1)
    return table[np.arange(table.shape[0]), best], best, flags[best]


def label_boxes(dt_boxes, gt_boxes, min_tp_iou: float = 0.7, max_fp_iou: float = 0.3,
                max_tp_candidates: int = 100, max_fp_candidates: int = 100):
    """Adds the fields 'tp_label' (SampleLabel per box) and 'instance_id' (best ground-truth box, -1
    without ground truth) to `dt_boxes`, in place."""
    if dt_boxes is None:
        return
    n = len(dt_boxes)
    labels = np.full(n, SampleLabel.IGNORE, np.int32)
    if gt_boxes is None or len(gt_boxes) == 0:
        owner = np.full(n, -1, np.int32)
        labels[select_candidates(np.ones(n, bool), max_fp_candidates)] = SampleLabel.FALSE_POSITIVE
    else:
        overlap, owner, ignored = _match_ground_truth(dt_boxes, gt_boxes)
        negatives = select_candidates(overlap < max_fp_iou, max_fp_candidates)
        positives = select_candidates((overlap > min_tp_iou) & (ignored == 0), max_tp_candidates)
        labels[positives] = SampleLabel.TRUE_POSITIVE
        labels[negatives] = SampleLabel.FALSE_POSITIVE          # written last, as in the reference
    dt_boxes.set_field("instance_id", owner)
    dt_boxes.set_field("tp_label", labels)


def get_regression_target(dt_boxes, gt_boxes):
    """Box offsets of labelled detections to the ground truth they belong to, as field
    'regression_target' (reference samples.py:152-157; host arithmetic on a handful of boxes)."""
    if not dt_boxes.has_field("instance_id"):
        raise ValueError("'instance_id' field is missing")
    owner = dt_boxes.get_field("instance_id")
    dt_boxes.add_field("regression_target", dt_boxes.get() - gt_boxes[owner].get())


# --------------------------------------------------------------------------------------- mining
def get_samples_from_image(model, image, gt_boxes, tp=True, fp=True, **kwargs):
    """Generator over the pyramid levels that hold selected detections of `model` on `image`: Boxes
    with 'scores', 'row', 'col', 'instance_id', 'tp_label' and 'samples' (their (m, n, C) crops)."""
    from . import channels as _channels
    from . import engine as _engine
    _channels._validate_image(image)
    shrink, n_per_oct, smooth, spec = _channels.read_opts(model.channel_opts)
    if spec.dtype == np.uint8:
        _channels._require_u8(image, spec.key)
    eng = _engine.get_engine(image.shape[0], image.shape[1], image.dtype, shrink, n_per_oct, smooth, 1, channels=spec)
    if eng.plan.n_levels == 0:
        return
    eng.load_images(image)
    eng.run_channels()
    found = model.scan_engine(eng)                       # every level in one launch group
    wanted = []
    if tp:
        wanted.append(SampleLabel.TRUE_POSITIVE)
    if fp:
        wanted.append(SampleLabel.FALSE_POSITIVE)
    level_of = found["level"]
    for lv in np.unique(level_of):
        idx = np.flatnonzero(level_of == lv)
        boxes = Boxes(found["boxes"][idx], scores=found["scores"][idx], row=found["r"][idx], col=found["c"][idx])
        label_boxes(boxes, gt_boxes, **kwargs)
        keep = np.flatnonzero(np.isin(boxes.get_field("tp_label"), wanted))
        if keep.size == 0:
            continue
        boxes = boxes[keep]
        crops = gather_samples_device(eng.level_tensor(0, int(lv)), boxes.get_field("row"), boxes.get_field("col"),
                                      model.shape)
        boxes.set_field("samples", crops.cpu().numpy())
        yield boxes


class SamplePool(object):
    """Pool of labelled training samples that is topped up from images and re-scored as the model grows."""

    def __init__(self, min_tp=1000, min_fp=1000, logger=None, **kwargs):
        self.samples = None
        self.min_tp, self.min_fp = min_tp, min_fp
        self.label_boxes_args = kwargs
        self.logger = logger or logging.getLogger("SamplePool")

    def _count(self, label):
        return 0 if self.samples is None else int((self.samples.get_field("tp_label") == label).sum())

    def pool_stats(self):
        return dict(num_tp=self._count(SampleLabel.TRUE_POSITIVE), num_fp=self._count(SampleLabel.FALSE_POSITIVE))

    def update_scores(self, model):
        """Re-evaluate the model on every pooled sample ('scores' becomes -inf where it rejects)."""
        if self.samples is not None:
            self.samples.set_field("scores", model.predict(self.samples.get_field("samples"))[0])

    def remove_low_scoring(self, min_score=-np.inf):
        if self.samples is None:
            return
        alive = self.samples.get_field("scores") > min_score
        self.logger.log(15, f"Removed {int((~alive).sum())}/{alive.size} samples (min_score={min_score:.2f})")
        self.samples = self.samples[np.flatnonzero(alive)]

    def update(self, model, iterable):
        """Drop what the current model rejects, then scan images from `iterable` (dicts with 'image' and
        'groundtruth_boxes') until the pool holds min_tp / min_fp samples again."""
        self.update_scores(model)
        self.remove_low_scoring()
        missing = {SampleLabel.TRUE_POSITIVE: max(self.min_tp - self._count(SampleLabel.TRUE_POSITIVE), 0),
                   SampleLabel.FALSE_POSITIVE: max(self.min_fp - self._count(SampleLabel.FALSE_POSITIVE), 0)}
        self.logger.log(15, f"Pool needs tp: {missing[SampleLabel.TRUE_POSITIVE]}, fp: {missing[SampleLabel.FALSE_POSITIVE]}")
        if not any(missing.values()):
            return
        fresh = []
        for item in iterable:
            for boxes in get_samples_from_image(model, item["image"], item["groundtruth_boxes"],
                                                tp=missing[SampleLabel.TRUE_POSITIVE] > 0,
                                                fp=missing[SampleLabel.FALSE_POSITIVE] > 0, **self.label_boxes_args):
                got = boxes.get_field("tp_label")
                for lab in missing:
                    missing[lab] -= int((got == lab).sum())
                fresh.append(boxes)
            if all(v <= 0 for v in missing.values()):
                break
        if fresh:
            self.samples = concatenate(([self.samples] if self.samples is not None else []) + fresh)

    def get_samples(self, label):
        """(X, H): feature maps (N, m, n, C) and scores (N,) of the pooled samples with `label` (copies)."""
        chosen = self.samples[np.flatnonzero(self.samples.get_field("tp_label") == label)]
        return chosen.get_field("samples").copy(), chosen.get_field("scores").flatten().copy()

    def get_true_positives(self):
        return self.get_samples(SampleLabel.TRUE_POSITIVE)

    def get_false_positives(self):
        return self.get_samples(SampleLabel.FALSE_POSITIVE)
