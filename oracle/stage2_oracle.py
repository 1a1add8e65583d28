"""CPU ORACLE (test infrastructure only) for the Stage-2 crop path: restates the reference's crop geometry
(/root/reference/src/two_stage_pipeline_yolox.py:244-289) and `SpeciesClassifier.preprocess`
(src/species_classifier.py:298-352) with plain PyTorch on the CPU.  The reference module itself cannot be imported here
(timm / cv2 absent) and holds no test vectors for this path: parity unpinned w.r.t. the reference's own tests."""
import numpy as np
import torch
import torch.nn.functional as F


def crop_rect(bbox, frame_hw, min_crop_size=32, crop_padding_percent=20):
    x1 = int(bbox['x1']); y1 = int(bbox['y1']); x2 = int(bbox['x2']); y2 = int(bbox['y2'])          # :245-248
    crop_w = x2 - x1; crop_h = y2 - y1                                                              # :251-252
    if crop_w < min_crop_size or crop_h < min_crop_size:                                            # :256
        return None
    padding_x = int(crop_w * crop_padding_percent / 100)                                           # :262-263
    padding_y = int(crop_h * crop_padding_percent / 100)
    h, w = frame_hw
    x1p = max(0, min(x1 - padding_x, w - 1)); y1p = max(0, min(y1 - padding_y, h - 1))              # :276-277
    x2p = max(0, min(x2 + padding_x, w)); y2p = max(0, min(y2 + padding_y, h))                      # :278-279
    if x2p <= x1p or y2p <= y1p:                                                                    # :281
        return None
    return x1p, y1p, x2p, y2p


def preprocess(crop_bgr: np.ndarray, input_size=336, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)) -> torch.Tensor:
    image = torch.from_numpy(np.ascontiguousarray(crop_bgr))
    image = image[:, :, [2, 1, 0]]                                                                  # :317
    if image.shape[0] != input_size or image.shape[1] != input_size:
        image = image.permute(2, 0, 1).unsqueeze(0).float()
        image = F.interpolate(image, size=(input_size, input_size), mode='bilinear', align_corners=False)   # :323-328
        image = image.squeeze(0)
    else:
        image = image.permute(2, 0, 1).float()
    image = image / 255.0                                                                           # :335
    image = (image - torch.tensor(mean).view(3, 1, 1)) / torch.tensor(std).view(3, 1, 1)            # :344
    return image.unsqueeze(0)
