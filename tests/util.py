"""Shared helpers for the parity tests (fixtures -> inputs, order-tolerant comparators)."""
import os

import numpy as np
import torch

from telescope_cam_detection_amd.arch import ARCHS
from telescope_cam_detection_amd.synth import make_frame
from telescope_cam_detection_amd.weights import synth_weights

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# fixture name -> arch (the .npz carries seeds / sizes / frame list itself)
CASE_ARCH = {
    "c1_r18_640_bs1": "r18", "c1_r18_640_scene": "r18", "c1_r18_640_resize": "r18",
    "c2_r50_640_bs8": "r50", "c2_r50_640_scene_bs2": "r50", "c3_r101_1280_bs1": "r101", "c3_r101_1280_bs4": "r101", "c4_r18_1920_bs1": "r18",
    "t_tiny_160": "tiny", "t_tiny_160x224": "tiny", "t_tinyb_192x128": "tinyb", "t_tinyc_160x224": "tinyc",
}


def load_case(name):
    g = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    arch = ARCHS[CASE_ARCH[name]]
    wseed, ih, iw, n = (int(v) for v in g["meta"])
    kind = str(g["kind"])
    frames = [make_frame(kind, int(s), int(h), int(w)) for s, h, w in g["frames"]]
    return arch, wseed, (ih, iw), frames, g


_WCACHE = {}


def weights_for(arch, seed):
    key = (arch.name, seed)
    if key not in _WCACHE:
        _WCACHE[key] = synth_weights(arch, seed)
    return _WCACHE[key]


def sample(t, n=2048):
    f = torch.as_tensor(t).detach().float().flatten()
    step = max(1, f.numel() // n)
    return f[::step][:n].numpy()


def match_detections(ref_labels, ref_boxes, ref_scores, labels, boxes, scores, score_tol, box_tol, return_unmatched=False):
    """Order-tolerant comparison of two (labels, boxes, scores) triples for ONE frame.

    torch.topk's tie order is unspecified and equal / near-equal scores exist even in fp32
    (SURVEY.md §7 "hard parts"), so rows are matched greedily: same label, |dscore| <= score_tol,
    max|dbox| <= box_tol.  Returns (n_matched, n_ref, worst_score_err, worst_box_err over matches)
    [+ the indices of the reference rows left without a partner when return_unmatched].
    """
    ref_labels, labels = np.asarray(ref_labels), np.asarray(labels)
    ref_boxes, boxes = np.asarray(ref_boxes, np.float64), np.asarray(boxes, np.float64)
    ref_scores, scores = np.asarray(ref_scores, np.float64), np.asarray(scores, np.float64)
    used = np.zeros(len(labels), bool)
    matched, ws, wb = 0, 0.0, 0.0
    unmatched = []
    for i in range(len(ref_labels)):
        cand = np.where((labels == ref_labels[i]) & ~used & (np.abs(scores - ref_scores[i]) <= score_tol))[0]
        if len(cand) == 0:
            unmatched.append(i)
            continue
        d = np.abs(boxes[cand] - ref_boxes[i]).max(axis=1)
        j = int(np.argmin(d))
        if d[j] <= box_tol:
            used[cand[j]] = True
            matched += 1
            ws = max(ws, abs(scores[cand[j]] - ref_scores[i]))
            wb = max(wb, d[j])
        else:
            unmatched.append(i)
    if return_unmatched:
        return matched, len(ref_labels), ws, wb, unmatched
    return matched, len(ref_labels), ws, wb
