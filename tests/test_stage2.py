"""Stage-2 crop batcher: geometry on the CPU, the batched crop+resize kernel on the GPU."""
import numpy as np
import pytest
import torch

from oracle import stage2_oracle as s2o
from telescope_cam_detection_amd.stage2 import CropBatcher, crop_rect
from telescope_cam_detection_amd.synth import scene_frame


def test_crop_geometry_matches_reference_arithmetic():
    rng = np.random.default_rng(0)
    hw = (1080, 1920)
    n_none = 0
    for _ in range(2000):
        x1, y1 = rng.uniform(-50, 1900), rng.uniform(-50, 1070)
        bbox = {"x1": x1, "y1": y1, "x2": x1 + rng.uniform(0, 700), "y2": y1 + rng.uniform(0, 700)}
        for pct in (0, 20, 33.3):
            a, b = crop_rect(bbox, hw, 32, pct), s2o.crop_rect(bbox, hw, 32, pct)
            assert a == b
            n_none += a is None
            if a is not None:
                assert 0 <= a[0] < a[2] <= hw[1] and 0 <= a[1] < a[3] <= hw[0]
    assert 0 < n_none < 6000
    assert crop_rect({"x1": 10.9, "y1": 10.9, "x2": 41.1, "y2": 41.9}, hw, 32, 20) is None       # 31 px after truncation
    assert crop_rect({"x1": 0, "y1": 0, "x2": 100, "y2": 50}, (60, 120), 32, 20) == (0, 0, 120, 60)
    b = CropBatcher(input_size=64)
    kept, rects = b.rects_for([{"bbox": {"x1": 5, "y1": 5, "x2": 10, "y2": 10}}, {"bbox": {"x1": 100, "y1": 100, "x2": 300, "y2": 260}}], hw)
    assert kept == [1] and rects == [(60, 68, 340, 292)]


@pytest.mark.gpu
@pytest.mark.parametrize("S", [336, 64])
def test_crop_batch_matches_torch_preprocess(S):
    rng = np.random.default_rng(5)
    frames = [scene_frame(11, 720, 1280), scene_frame(12, 480, 640)]
    dev = [torch.from_numpy(f).cuda() for f in frames]
    batcher = CropBatcher(input_size=S)
    rects_per_frame = []
    for f in frames:
        h, w = f.shape[:2]
        rects = []
        for _ in range(9):                                               # config 5: sides drawn from [64, 512), mixed sizes
            cw, ch = int(rng.integers(64, min(512, w))), int(rng.integers(64, min(512, h)))
            x, y = int(rng.integers(0, w - cw + 1)), int(rng.integers(0, h - ch + 1))
            rects.append((x, y, x + cw, y + ch))
        rects.append((0, 0, S, S) if S <= min(h, w) else (0, 0, w, h))   # identity-size crop
        rects.append((w - 3, h - 2, w, h))                               # tiny crop: heavy upsampling, edge clamping
        rects_per_frame.append(rects)
    out = batcher.preprocess_batch(dev, rects_per_frame).cpu()
    flat = [(f, r) for f, rects in zip(frames, rects_per_frame) for r in rects]
    assert out.shape == (len(flat), 3, S, S)
    for i, (f, (x1, y1, x2, y2)) in enumerate(flat):
        want = s2o.preprocess(f[y1:y2, x1:x2], S)[0]
        torch.testing.assert_close(out[i], want, atol=2e-5, rtol=1e-5)
    assert batcher.preprocess_batch(dev, [[], []]).shape == (0, 3, S, S)
