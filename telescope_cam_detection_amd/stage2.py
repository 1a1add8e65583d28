"""Stage-2 crop batcher.

The reference classifies every detection separately, bs=1: crop geometry in
`TwoStageDetectionPipeline.classify_detection` (/root/reference/src/two_stage_pipeline_yolox.py:244-289) then
`SpeciesClassifier.preprocess` (src/species_classifier.py:298-352) per crop.  Here all crops of a frame (or of
several frames) become ONE `[N, 3, S, S]` classifier batch with one HIP launch (`rtd_crop_resize_batch`).
The classifier network itself (timm EVA02, fetched by name) is out of scope - this module stops at its input.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _capi

IMAGENET_MEAN = (0.485, 0.456, 0.406)      # defaults of the reference classifier's data config
IMAGENET_STD = (0.229, 0.224, 0.225)


def crop_rect(bbox: Dict[str, float], frame_hw: Tuple[int, int], min_crop_size: int = 32,
              crop_padding_percent: float = 20) -> Optional[Tuple[int, int, int, int]]:
    """Crop rectangle (x1, y1, x2, y2) of one detection, or None when Stage 2 skips it.

    Same arithmetic as src/two_stage_pipeline_yolox.py:244-283: truncate the box to ints, skip boxes narrower /
    lower than `min_crop_size` BEFORE padding, pad each side by int(extent * pct / 100), clamp to the frame,
    reject empty rectangles."""
    x1, y1, x2, y2 = int(bbox["x1"]), int(bbox["y1"]), int(bbox["x2"]), int(bbox["y2"])
    crop_w, crop_h = x2 - x1, y2 - y1
    if crop_w < min_crop_size or crop_h < min_crop_size:
        return None
    pad_x = int(crop_w * crop_padding_percent / 100)
    pad_y = int(crop_h * crop_padding_percent / 100)
    h, w = frame_hw
    xa = max(0, min(x1 - pad_x, w - 1))
    ya = max(0, min(y1 - pad_y, h - 1))
    xb = max(0, min(x2 + pad_x, w))
    yb = max(0, min(y2 + pad_y, h))
    if xb <= xa or yb <= ya:
        return None
    return xa, ya, xb, yb


class CropBatcher:
    """Builds the classifier input batch on the GPU.  Frames must be HWC uint8 BGR torch tensors on the device."""

    def __init__(self, input_size: int = 336, mean: Sequence[float] = IMAGENET_MEAN, std: Sequence[float] = IMAGENET_STD,
                 min_crop_size: int = 32, crop_padding_percent: float = 20):
        self.input_size = int(input_size)
        self.mean = (C.c_float * 3)(*mean)
        self.std = (C.c_float * 3)(*std)
        self.min_crop_size = min_crop_size
        self.crop_padding_percent = crop_padding_percent

    def rects_for(self, detections: List[Dict], frame_hw) -> Tuple[List[int], List[Tuple[int, int, int, int]]]:
        kept, rects = [], []
        for i, det in enumerate(detections):
            r = crop_rect(det["bbox"], frame_hw, self.min_crop_size, self.crop_padding_percent)
            if r is not None:
                kept.append(i)
                rects.append(r)
        return kept, rects

    def preprocess_batch(self, frames, rects_per_frame):
        """frames: list of device uint8 HWC tensors; rects_per_frame: list (per frame) of (x1,y1,x2,y2) lists.
        Returns a [N, 3, S, S] fp32 torch tensor on the frames' device (N = total number of crops)."""
        import torch

        flat = [(f, r) for f, rects in zip(frames, rects_per_frame) for r in rects]
        n = len(flat)
        dev = frames[0].device if frames else torch.device("cuda", 0)
        S = self.input_size
        out = torch.empty((n, 3, S, S), dtype=torch.float32, device=dev)
        if n == 0:
            return out
        ptrs = (C.c_void_p * n)()
        hw = (C.c_int32 * (2 * n))()
        rc = (C.c_int32 * (4 * n))()
        for i, (f, r) in enumerate(flat):
            assert f.is_cuda and f.dtype == torch.uint8 and f.is_contiguous() and f.dim() == 3 and f.shape[2] == 3
            ptrs[i] = f.data_ptr()
            hw[2 * i], hw[2 * i + 1] = int(f.shape[0]), int(f.shape[1])
            rc[4 * i], rc[4 * i + 1], rc[4 * i + 2], rc[4 * i + 3] = (int(v) for v in r)
        stream = torch.cuda.current_stream(dev).cuda_stream
        code = _capi.lib().rtd_crop_resize_batch(n, ptrs, hw, rc, S, self.mean, self.std, out.data_ptr(), C.c_void_p(stream))
        if code != _capi.RTD_OK:
            _capi._raise(code, None)
        return out
