"""``detect_on_images`` -- the evaluation loop's caller of the hot path (reference testing.py:127-132).
The evaluation itself (``Evaluator``, ``evaluate_model``) is outside this build."""
import numpy as np

from .boxes import Boxes


def detect_on_images(images, *model, gt_key="groundtruth_boxes"):
    """Yield (gt_boxes, dt_boxes, image.shape[:2]) for every dict of `images` (keys 'image' and
    `gt_key`), detections from ``waldboost_amd.detect(image, *model)``."""
    from . import detect
    empty_boxes = Boxes(np.empty((0, 4)), ignore=np.empty(0))
    for data_dict in images:
        image = data_dict.get("image")
        gt_boxes = data_dict.get(gt_key, empty_boxes)
        dt_boxes = detect(image, *model)
        yield gt_boxes, dt_boxes, image.shape[:2]
