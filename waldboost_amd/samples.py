"""Samples from images -- drop-in for the detection-side part of ``waldboost.samples``
(reference samples.py): the training pipeline's hard-negative mining runs the hot path over every
training image and then crops the surviving windows out of the channel pyramid.

``gather_samples`` (reference samples.py:14-43) is a HIP copy kernel; ``get_samples_from_image``
(reference samples.py:160-216) keeps the channel pyramid on the GPU between the cascade scan and
the crop, so only the selected samples travel to the host.  ``label_boxes`` /
``select_candidates`` are the reference's NumPy host logic (samples.py:46-157) with the
third-party ``bbx.iou`` replaced by ``boxes.iou``.  ``SamplePool`` re-scores its samples with
``Model.predict`` on the GPU.
"""
import logging

import numpy as np

from . import _native as nat
from .boxes import Boxes, concatenate, iou


def _dev_pos(rs, cs, u, v, m, n):
    import torch
    rs = np.asarray(rs).reshape(-1)
    cs = np.asarray(cs).reshape(-1)
    if rs.size and (rs.min() < 0 or cs.min() < 0 or rs.max() + m > u or cs.max() + n > v):
        # the reference slices without checks and would build a ragged object array here
        raise IndexError("sample window outside the channel image")
    dev = nat.require_gpu()
    return torch.from_numpy(rs.astype(np.int32)).to(dev), torch.from_numpy(cs.astype(np.int32)).to(dev)


def gather_samples(chns, rs, cs, shape) -> np.ndarray:
    """Crop feature maps: X[i] = chns[rs[i]:rs[i]+m, cs[i]:cs[i]+n, :] -> (N, m, n, C), dtype of chns
    (reference samples.py:14-43).  `chns` may be a NumPy array or a device tensor (float32/uint8)."""
    import torch
    rs = np.asarray(rs)
    cs = np.asarray(cs)
    if rs.size != cs.size:
        raise ValueError("Sizes of 'rs' and 'cs' must match")
    m, n, _ = shape
    if rs.size == 0:
        return np.empty((0,) + tuple(shape), dtype=_np_dtype(chns))
    return gather_samples_device(chns, rs, cs, shape).cpu().numpy()


def _np_dtype(chns):
    return np.dtype(str(chns.dtype).replace("torch.", ""))


def gather_samples_device(chns, rs, cs, shape):
    """gather_samples returning the device tensor (N, m, n, C)."""
    import torch
    lib = nat.load()
    dev = nat.require_gpu()
    dt = _np_dtype(chns)
    if dt not in (np.dtype(np.float32), np.dtype(np.uint8)):
        raise TypeError(f"channel image must be float32 or uint8 (as produced by channel_pyramid), got {dt}")
    tdt, wdt = (torch.uint8, nat.WB_DTYPE_U8) if dt == np.uint8 else (torch.float32, nat.WB_DTYPE_F32)
    u, v, C = chns.shape
    m, n = int(shape[0]), int(shape[1])
    Xd = chns if isinstance(chns, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(chns))
    Xd = Xd.to(dev, tdt).contiguous()
    rd, cd = _dev_pos(rs, cs, u, v, m, n)
    N = int(rd.numel())
    out = torch.empty((N, m, n, C), dtype=tdt, device=dev)
    nat.check(lib.wb_gather_samples_launch(nat.stream_ptr(), nat.ptr(Xd), wdt, u, v, C, nat.ptr(rd), nat.ptr(cd), N,
                                           m, n, nat.ptr(out)), "wb_gather_samples_launch")
    return out


def select_candidates(condition, max_candidates: int) -> np.ndarray:
    """At most max_candidates indices where condition is True (reference samples.py:46-78)."""
    idx = np.flatnonzero(condition)
    if idx.size > max_candidates:
        idx = np.random.choice(idx, max_candidates)
    return idx


class SampleLabel:
    """Constants for labeling samples as true/false positives (reference samples.py:81-85)."""
    TRUE_POSITIVE = 1
    FALSE_POSITIVE = -1
    IGNORE = 0


def label_boxes(dt_boxes, gt_boxes, min_tp_iou: float = 0.7, max_fp_iou: float = 0.3,
                max_tp_candidates: int = 100, max_fp_candidates: int = 100):
    """Label boxes as TP / FP / ignore and assign the ground-truth instance (reference
    samples.py:88-149); mutates dt_boxes by adding 'instance_id' and 'tp_label'."""
    if dt_boxes is None:
        return
    if gt_boxes is not None and len(gt_boxes) > 0:
        ignore_flag = gt_boxes.get_field("ignore") if gt_boxes.has_field("ignore") else np.zeros(len(gt_boxes))
        if ignore_flag.ndim != 1:
            raise ValueError("'ignore' field must be single dimension")
        overlap = iou(dt_boxes, gt_boxes)
        dt_iou = np.max(overlap, axis=1)
        dt_instance_id = np.argmax(overlap, axis=1)
        dt_ignore_flag = ignore_flag[dt_instance_id]
        fp = select_candidates(dt_iou < max_fp_iou, max_fp_candidates)
        tp = select_candidates(np.logical_and(dt_iou > min_tp_iou, dt_ignore_flag == 0), max_tp_candidates)
        box_label = np.full(len(dt_boxes), SampleLabel.IGNORE, np.int32)
        box_label[tp] = SampleLabel.TRUE_POSITIVE
        box_label[fp] = SampleLabel.FALSE_POSITIVE
    else:
        dt_instance_id = np.full(len(dt_boxes), -1, np.int32)
        box_label = np.full(len(dt_boxes), SampleLabel.IGNORE, np.int32)
        fp = select_candidates(np.ones(len(dt_boxes), bool), max_fp_candidates)     # reference: np.bool (removed in NumPy 1.24)
        box_label[fp] = SampleLabel.FALSE_POSITIVE
    dt_boxes.set_field("instance_id", dt_instance_id)
    dt_boxes.set_field("tp_label", box_label)


def get_samples_from_image(model, image, gt_boxes, tp=True, fp=True, **kwargs):
    """Scan `image` with `model` and yield, level by level, Boxes of the selected detections with
    fields 'scores', 'row', 'col', 'instance_id', 'tp_label' and 'samples' (the (m,n,C) crops of the
    channel pyramid) -- reference samples.py:160-216.  The pyramid and the scan stay on the GPU;
    only detections and the selected crops are copied to the host."""
    from . import channels as _channels
    from . import engine as _engine
    _channels._validate_image(image)
    shrink, n_per_oct, smooth, spec = _channels.read_opts(model.channel_opts)
    if spec.dtype == np.uint8:
        _channels._require_u8(image, spec.key)
    H, W = image.shape
    eng = _engine.get_engine(H, W, image.dtype, shrink, n_per_oct, smooth, 1, channels=spec)
    if eng.plan.n_levels == 0:
        return
    eng.load_images(image)
    eng.run_channels()
    res = model.scan_engine(eng)
    for lv in range(eng.plan.n_levels):
        sel = np.flatnonzero(res["level"] == lv)
        if sel.size == 0:
            continue
        r, c, h = res["r"][sel], res["c"][sel], res["scores"][sel]
        dt_boxes = Boxes(res["boxes"][sel])               # dt_boxes in the original image space
        dt_boxes.set_field("scores", h)
        dt_boxes.set_field("row", r)
        dt_boxes.set_field("col", c)
        label_boxes(dt_boxes, gt_boxes, **kwargs)
        tp_label = dt_boxes.get_field("tp_label")
        sample_selector = np.logical_or(np.logical_and(tp_label == SampleLabel.TRUE_POSITIVE, tp),
                                        np.logical_and(tp_label == SampleLabel.FALSE_POSITIVE, fp))
        sample_indices = np.flatnonzero(sample_selector)
        dt_boxes = dt_boxes[sample_indices]
        if len(dt_boxes) == 0:
            continue
        samples = gather_samples_device(eng.level_tensor(0, lv), dt_boxes.get_field("row").flatten(),
                                        dt_boxes.get_field("col").flatten(), model.shape)
        dt_boxes.set_field("samples", samples.cpu().numpy())
        yield dt_boxes


class SamplePool(object):
    """Container for training samples (reference samples.py:219-338)."""

    def __init__(self, min_tp=1000, min_fp=1000, logger=None, **kwargs):
        self.samples = None
        self.min_tp = min_tp
        self.min_fp = min_fp
        self.label_boxes_args = kwargs
        self.logger = logger or logging.getLogger("SamplePool")

    def update(self, model, iterable):
        """Add new samples by scanning the images provided by iterable."""
        self.update_scores(model)
        self.remove_low_scoring()
        stats = self.pool_stats()
        sample_tp = max(self.min_tp - stats["num_tp"], 0)
        sample_fp = max(self.min_fp - stats["num_fp"], 0)
        self.logger.log(15, f"Pool size: tp: {stats['num_tp']}/{self.min_tp}, fp: {stats['num_fp']}/{self.min_fp}")
        if sample_tp or sample_fp:
            new_samples = []
            for gt_dict in iterable:
                image = gt_dict["image"]
                gt_boxes = gt_dict["groundtruth_boxes"]
                for dt_boxes in get_samples_from_image(model, image, gt_boxes, tp=sample_tp > 0, fp=sample_fp > 0,
                                                       **self.label_boxes_args):
                    sample_label = dt_boxes.get_field("tp_label")
                    sample_tp -= (sample_label == SampleLabel.TRUE_POSITIVE).sum()
                    sample_fp -= (sample_label == SampleLabel.FALSE_POSITIVE).sum()
                    new_samples.append(dt_boxes)
                if sample_fp <= 0 and sample_tp <= 0:
                    break
            if new_samples:
                self.samples = concatenate(([self.samples] if self.samples is not None else []) + new_samples)

    def pool_stats(self):
        if self.samples is None:
            return dict(num_tp=0, num_fp=0)
        labels = self.samples.get_field("tp_label")
        return dict(num_tp=(labels == SampleLabel.TRUE_POSITIVE).sum(), num_fp=(labels == SampleLabel.FALSE_POSITIVE).sum())

    def update_scores(self, model):
        if self.samples is not None:
            new_scores, _ = model.predict(self.samples.get_field("samples"))
            self.samples.set_field("scores", new_scores)

    def remove_low_scoring(self, min_score=-np.inf):
        """Remove samples rejected by the model from the pool."""
        if self.samples is not None:
            mask = self.samples.get_field("scores") > min_score
            self.samples = self.samples[np.flatnonzero(mask)]
            self.logger.log(15, f"Removed {(mask == 0).sum()}/{len(mask)} samples (min_score={min_score:.2f})")

    def get_samples(self, label):
        labels = self.samples.get_field("tp_label")
        boxes = self.samples[np.flatnonzero(labels == label)]
        return boxes.get_field("samples").copy(), boxes.get_field("scores").flatten().copy()

    def get_true_positives(self):
        return self.get_samples(label=SampleLabel.TRUE_POSITIVE)

    def get_false_positives(self):
        return self.get_samples(label=SampleLabel.FALSE_POSITIVE)
