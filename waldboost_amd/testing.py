"""``detect_on_images`` -- the evaluation loop's caller of the hot path (reference testing.py:127-132).
The evaluation itself (``Evaluator``, ``evaluate_model``) is outside this build."""
import collections

import numpy as np

from .boxes import Boxes


def detect_on_images(images, *model, gt_key="groundtruth_boxes", lanes=1, batch=1):
    """Yield (gt_boxes, dt_boxes, image.shape[:2]) for every dict of `images` (keys 'image' and
    `gt_key`), detections from ``waldboost_amd.detect(image, *model)``.
    lanes / batch (an extension; one model only): more than one of either runs the images through
    ``Model.detect_stream`` -- same tuples in the same order, but up to lanes * batch - 1 dicts are taken from `images`
    ahead of the tuple being yielded."""
    from . import detect
    empty_boxes = Boxes(np.empty((0, 4)), ignore=np.empty(0))
    if (lanes > 1 or batch > 1) and len(model) == 1:
        meta = collections.deque()

        def source():
            for data_dict in images:
                image = data_dict.get("image")
                meta.append((data_dict.get(gt_key, empty_boxes), image.shape[:2]))
                yield image

        for dt_boxes in model[0].detect_stream(source(), lanes=lanes, batch=batch):
            gt_boxes, shape = meta.popleft()
            dt_boxes.set_field("label", np.zeros(len(dt_boxes), np.int64))      # (what detect() adds for its one model)
            yield gt_boxes, dt_boxes, shape
        return
    for data_dict in images:
        image = data_dict.get("image")
        gt_boxes = data_dict.get(gt_key, empty_boxes)
        dt_boxes = detect(image, *model)
        yield gt_boxes, dt_boxes, image.shape[:2]
