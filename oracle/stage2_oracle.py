"""CPU ORACLE (test infrastructure only) for the Stage-2 crop path: restates the reference's crop geometry
(/root/reference/src/two_stage_pipeline_yolox.py:244-289) and `SpeciesClassifier.preprocess`
(src/species_classifier.py:298-352) with plain PyTorch on the CPU.  The reference module itself cannot be imported here
(timm / cv2 absent) and holds no test vectors for this path: parity unpinned w.r.t. the reference's own tests."""
import numpy as np
import torch
import torch.nn.functional as F


def crop_rect(bbox, frame_hw, min_crop_size=64, crop_padding_percent=20):
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


def ensure_valid_bbox(bbox, min_size=1):                                                          # src/bbox_utils.py:12-59
    x1, y1, x2, y2 = bbox['x1'], bbox['y1'], bbox['x2'], bbox['y2']
    if x1 > x2:
        x1, x2 = x2, x1
    if y1 > y2:
        y1, y2 = y2, y1
    if x2 - x1 < min_size:
        x2 = x1 + min_size
    if y2 - y1 < min_size:
        y2 = y1 + min_size
    return {'x1': x1, 'y1': y1, 'x2': x2, 'y2': y2, 'width': x2 - x1, 'height': y2 - y1, 'area': (x2 - x1) * (y2 - y1)}


def classify(classifier, crop_bgr, top_k):
    """`SpeciesClassifier.classify` (src/species_classifier.py:354-419) for ONE crop: preprocess, forward, softmax, top-k, formatting."""
    x = preprocess(crop_bgr, classifier.input_size)
    dev = next(classifier.model.parameters()).device
    with torch.no_grad():
        probs = torch.softmax(classifier.model(x.to(dev)), dim=1).float().cpu()
    top_probs, top_indices = torch.topk(probs[0], top_k)
    results = []
    for prob, idx in zip(top_probs, top_indices):
        prob = prob.item(); idx = idx.item()
        min_threshold = 0.1 if classifier.use_hierarchical else classifier.confidence_threshold      # :388
        if prob < min_threshold:
            continue
        label, tax_level = classifier.get_hierarchical_label(idx, prob)                              # :393
        if label is not None:
            if classifier.enable_geographic_filter and classifier.allowed_species:                   # :397-401
                if label not in classifier.allowed_species:
                    continue
            results.append({'species': label, 'confidence': prob, 'class_id': idx, 'taxonomic_level': tax_level})
    return results


def classify_detection(pipeline, frame_bgr: np.ndarray, detection, is_active=None):
    """`TwoStageDetectionPipeline.classify_detection` (src/two_stage_pipeline_yolox.py:203-451) for ONE detection, without an enhancer."""
    def set_fields(det, species, confidence, category, level):                                       # :180-201
        det['species'] = species; det['species_confidence'] = float(confidence)
        det['stage2_category'] = category; det['taxonomic_level'] = level
    if not pipeline.enable_species_classification:
        return detection
    bbox = ensure_valid_bbox(detection.get('bbox', {}))                                              # :233-238
    detection['bbox'] = bbox
    category = pipeline.class_id_to_category.get(detection.get('class_id'))                          # :241
    if category not in pipeline.species_classifiers:                                                 # :244-248
        detection['species'] = None; detection['species_confidence'] = 0.0
        return detection
    rect = crop_rect(bbox, frame_bgr.shape[:2], pipeline.min_crop_size, pipeline.crop_padding_percent)
    if rect is None:                                                                                 # :256-259 / :281-285
        set_fields(detection, None, 0.0, category, None)
        return detection
    x1, y1, x2, y2 = rect
    crop = frame_bgr[y1:y2, x1:x2]                                                                   # :289
    classifier = pipeline.species_classifiers[category]
    time_of_day = detection.get('time_of_day')
    top_k = pipeline.time_of_day_top_k if time_of_day else 1                                         # :385
    results = classify(classifier, crop, top_k)
    if results:
        if time_of_day and is_active is not None:                                                    # :393-419
            for r in results:
                r['confidence_original'] = r['confidence']
                r['activity_boosted'] = bool(is_active(r['species'], time_of_day))
                if not r['activity_boosted']:
                    r['confidence'] = r['confidence'] * pipeline.time_of_day_penalty
            results.sort(key=lambda x: x['confidence'], reverse=True)
        top = results[0]
        level = top.get('taxonomic_level', 'species')
        if level in pipeline.rejected_taxonomic_levels:                                              # :437-439
            set_fields(detection, None, 0.0, category, None)
        else:
            set_fields(detection, top['species'], top['confidence'], category, level)
    else:
        set_fields(detection, None, 0.0, category, None)
    return detection
