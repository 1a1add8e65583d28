"""Stage-2 crop batcher.

The reference classifies every detection separately, bs=1: crop geometry in
`TwoStageDetectionPipeline.classify_detection` (/root/reference/src/two_stage_pipeline_yolox.py:244-289) then
`SpeciesClassifier.preprocess` (src/species_classifier.py:298-352) per crop.  Here all crops of a frame (or of
several frames) become ONE `[N, 3, S, S]` classifier batch with one HIP launch (`rtd_crop_resize_batch`).
`BatchedStage2` is the consumer of that batch: it replaces the per-detection loop of
`TwoStageDetectionPipeline.process_detections` (src/two_stage_pipeline_yolox.py:453-481 -> :203-451) with crop geometry for
every detection -> ONE crop launch -> ONE classifier forward per taxonomic category -> the same per-detection result rules
(`SpeciesClassifier.classify`'s formatting, src/species_classifier.py:383-413, the time-of-day re-ranking and the rejected
taxonomic levels of :393-445), detections returned in their input order.  The classifier network itself (timm EVA02-L/14@336,
fetched by name, src/species_classifier.py:252-262) is out of scope; the tests drive the glue with a declared stand-in that has the
reference classifier's attributes (tests/standins.py), `bench.py --workload two_stage` times the crop batch alone.
"""
from __future__ import annotations

import logging
import time

import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _capi

IMAGENET_MEAN = (0.485, 0.456, 0.406)      # defaults of the reference classifier's data config
IMAGENET_STD = (0.229, 0.224, 0.225)

logger = logging.getLogger(__name__)


def crop_rect(bbox: Dict[str, float], frame_hw: Tuple[int, int], min_crop_size: int = 64,
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
                 min_crop_size: int = 64, crop_padding_percent: float = 20):
        # defaults of the reference pipeline: min_crop_size 64, 20 % padding (src/two_stage_pipeline_yolox.py:42-43, main.py:1067)
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

    MAX_CROPS_PER_LAUNCH = 64        # rtd_crop_resize_batch's limit (include/rtdetr_mi355.h)

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
        stream = torch.cuda.current_stream(dev).cuda_stream
        for i0 in range(0, n, self.MAX_CROPS_PER_LAUNCH):               # the kernel takes up to 64 crops per launch: a busy frame batch is chunked
            part = flat[i0:i0 + self.MAX_CROPS_PER_LAUNCH]
            k = len(part)
            ptrs = (C.c_void_p * k)()
            hw = (C.c_int32 * (2 * k))()
            rc = (C.c_int32 * (4 * k))()
            for i, (f, r) in enumerate(part):
                assert f.is_cuda and f.dtype == torch.uint8 and f.is_contiguous() and f.dim() == 3 and f.shape[2] == 3
                ptrs[i] = f.data_ptr()
                hw[2 * i], hw[2 * i + 1] = int(f.shape[0]), int(f.shape[1])
                rc[4 * i], rc[4 * i + 1], rc[4 * i + 2], rc[4 * i + 3] = (int(v) for v in r)
            code = _capi.lib().rtd_crop_resize_batch(k, ptrs, hw, rc, S, self.mean, self.std, out[i0:i0 + k].data_ptr(), C.c_void_p(stream))
            if code != _capi.RTD_OK:
                _capi._raise(code, None)
        return out


def normalised_bbox(bbox: Dict[str, float], min_size: int = 1) -> Dict[str, float]:
    """What the pipeline does to a bbox before cropping (src/two_stage_pipeline_yolox.py:232-238 via src/bbox_utils.py:12-59):
    inverted corners swapped, extents of at least `min_size`, width / height / area recomputed (un-truncated floats).  The
    reference's own function is used when its package is importable."""
    try:
        from src.bbox_utils import ensure_valid_bbox  # type: ignore
        return ensure_valid_bbox(bbox, min_size)
    except Exception:
        pass
    xa, xb = sorted((bbox["x1"], bbox["x2"]))
    ya, yb = sorted((bbox["y1"], bbox["y2"]))
    xb = xb if xb - xa >= min_size else xa + min_size
    yb = yb if yb - ya >= min_size else ya + min_size
    return {"x1": xa, "y1": ya, "x2": xb, "y2": yb, "width": xb - xa, "height": yb - ya, "area": (xb - xa) * (yb - ya)}


def format_predictions(classifier, probs_row, top_k: int) -> List[Dict]:
    """Top-k of one softmax row -> the result dicts of `SpeciesClassifier.classify` (src/species_classifier.py:383-413): threshold
    (0.1 in hierarchical mode, else `confidence_threshold`), hierarchical label for the confidence, geographic whitelist."""
    import torch

    top_probs, top_idx = torch.topk(probs_row, min(top_k, probs_row.numel()))
    out = []
    floor = 0.1 if getattr(classifier, "use_hierarchical", False) else classifier.confidence_threshold
    for prob, idx in zip(top_probs.tolist(), top_idx.tolist()):
        if prob < floor:
            continue
        label, level = classifier.get_hierarchical_label(idx, prob)
        if label is None:
            continue
        if getattr(classifier, "enable_geographic_filter", False) and getattr(classifier, "allowed_species", None):
            if label not in classifier.allowed_species:
                continue
        out.append({"species": label, "confidence": prob, "class_id": idx, "taxonomic_level": level})
    return out


class BatchedStage2:
    """Batched stand-in for `TwoStageDetectionPipeline.process_detections`.  `pipeline` is the reference pipeline object (or
    anything with its attributes: enable_species_classification, class_id_to_category, species_classifiers, min_crop_size,
    crop_padding_percent, rejected_taxonomic_levels, time_of_day_top_k, time_of_day_penalty, enhancer); its classifiers need
    `.model` (callable on a [N,3,S,S] batch) plus the attributes `format_predictions` reads.  Pipelines with an image enhancer
    (Real-ESRGAN, ~1 s per crop) keep the reference's own per-detection path."""

    def __init__(self, pipeline, batcher: Optional[CropBatcher] = None, activity_fn=None):
        self.pipeline = pipeline
        self.batcher = batcher or CropBatcher(min_crop_size=pipeline.min_crop_size, crop_padding_percent=pipeline.crop_padding_percent)
        if activity_fn is None:
            try:
                from src.species_activity_patterns import is_species_likely_active as activity_fn  # type: ignore
            except Exception:
                activity_fn = None
        self.activity_fn = activity_fn
        self._warned_no_activity = False

    @staticmethod
    def _set(det, species, confidence, category, level):            # src/two_stage_pipeline_yolox.py:180-201
        det["species"] = species
        det["species_confidence"] = float(confidence)
        det["stage2_category"] = category
        det["taxonomic_level"] = level

    def _conclude(self, det, category, results):
        """From the classifier's result list to the detection's species fields (src/two_stage_pipeline_yolox.py:393-445)."""
        p = self.pipeline
        if not results:
            self._set(det, None, 0.0, category, None)
            return
        tod = det.get("time_of_day")
        if tod and self.activity_fn is None and not self._warned_no_activity:
            # the reference always re-ranks when det['time_of_day'] is set (src/two_stage_pipeline_yolox.py:393-425); without its
            # src.species_activity_patterns module (or an activity_fn argument) that step cannot run here - say so once, loudly
            self._warned_no_activity = True
            logger.warning("Stage 2: detections carry time_of_day but no species-activity function is available "
                           "(src.species_activity_patterns not importable, no activity_fn given): time-of-day re-ranking is skipped")
        if tod and self.activity_fn is not None:
            for r in results:
                r["confidence_original"] = r["confidence"]
                r["activity_boosted"] = bool(self.activity_fn(r["species"], tod))
                if not r["activity_boosted"]:
                    r["confidence"] = r["confidence"] * p.time_of_day_penalty
            results.sort(key=lambda r: r["confidence"], reverse=True)
        top = results[0]
        level = top.get("taxonomic_level", "species")
        if level in p.rejected_taxonomic_levels:
            self._set(det, None, 0.0, category, None)
        else:
            self._set(det, top["species"], top["confidence"], category, level)

    def process_batch(self, frames, detections_per_frame):
        """frames: device-resident HWC uint8 BGR tensors; detections_per_frame: Stage-1 dict lists (mutated in place and returned,
        order kept).  One crop launch for every eligible detection of every frame, one forward per category."""
        import torch

        p = self.pipeline
        if not p.enable_species_classification:
            for dets in detections_per_frame:
                for d in dets:
                    d["species"] = None
                    d["species_confidence"] = 0.0
            return detections_per_frame
        if getattr(p, "enhancer", None) is not None:
            return [p.process_detections(f, dets) for f, dets in zip(frames, detections_per_frame)]
        jobs = []                                                  # (frame index, detection, category, rect)
        for fi, (f, dets) in enumerate(zip(frames, detections_per_frame)):
            hw = (int(f.shape[0]), int(f.shape[1]))
            for d in dets:
                bbox = normalised_bbox(d.get("bbox", {}))
                d["bbox"] = bbox
                category = p.class_id_to_category.get(d.get("class_id"))
                if category not in p.species_classifiers:          # :227-231
                    d["species"] = None
                    d["species_confidence"] = 0.0
                    continue
                rect = crop_rect(bbox, hw, p.min_crop_size, p.crop_padding_percent)
                if rect is None:                                   # too small / empty crop (:256-259, :281-285)
                    self._set(d, None, 0.0, category, None)
                    continue
                jobs.append((fi, d, category, rect))
        if not jobs:
            return detections_per_frame
        rects_per_frame = [[] for _ in frames]
        order = [[] for _ in frames]
        for j, (fi, _, _, rect) in enumerate(jobs):
            rects_per_frame[fi].append(rect)
            order[fi].append(j)
        batch = self.batcher.preprocess_batch(frames, rects_per_frame)          # [N,3,S,S], crops in frame-major order
        row_of = {j: r for r, j in enumerate(j for per in order for j in per)}
        by_cat: Dict[str, List[int]] = {}
        for j, (_, _, category, _) in enumerate(jobs):
            by_cat.setdefault(category, []).append(j)
        for category, js in by_cat.items():
            clf = p.species_classifiers[category]
            t_fwd = time.perf_counter()
            try:
                with torch.no_grad():
                    x = batch[torch.tensor([row_of[j] for j in js], device=batch.device)]
                    probs = torch.softmax(clf.model(x), dim=1).float().cpu()
            except Exception:                                      # :447-451: a failing classifier leaves its detections unlabelled
                for j in js:
                    d = jobs[j][1]
                    d["species"], d["species_confidence"], d["taxonomic_level"] = None, 0.0, None
                continue
            for row, j in enumerate(js):
                d = jobs[j][1]
                top_k = p.time_of_day_top_k if d.get("time_of_day") else 1
                self._conclude(d, category, format_predictions(clf, probs[row], top_k))
            # the pipeline's get_stats() averages `classification_times` (milliseconds, one entry per classified detection in the
            # reference, src/two_stage_pipeline_yolox.py:380-386, :509-511): every crop of the batched forward books its share of it
            times = getattr(p, "classification_times", None)
            if times is not None:
                share = (time.perf_counter() - t_fwd) * 1000.0 / len(js)
                for _ in js:
                    times.append(share)
        return detections_per_frame

    def process_detections(self, frame, detections):
        """Drop-in for `TwoStageDetectionPipeline.process_detections(frame, detections)` (one frame)."""
        import torch

        if isinstance(frame, np.ndarray):
            frame = torch.from_numpy(np.ascontiguousarray(frame)).cuda()
        return self.process_batch([frame.contiguous()], [detections])[0]
