"""STAND-INS for the reference's Stage-2 classifier and pipeline objects (test infrastructure only - nothing in the package imports
this): they give `BatchedStage2` something with the reference objects' attributes to drive; they say nothing about the real
classifier's accuracy or cost."""
import numpy as np


class StandInSpeciesClassifier:
    """STAND-IN for the reference's `SpeciesClassifier` (timm EVA02-L/14@336 + iNat21 taxonomy, both unavailable offline): a small
    seeded conv net over the same [N,3,S,S] input with the attributes `BatchedStage2` / `format_predictions` read.  It exists so that
    the batched glue can be tested end to end and timed; it says nothing about the real classifier's accuracy or cost."""

    def __init__(self, num_classes: int = 40, input_size: int = 336, device: str = "cuda:0", seed: int = 0, use_hierarchical: bool = True,
                 confidence_threshold: float = 0.3):
        import torch
        import torch.nn as nn

        g = torch.Generator().manual_seed(seed)
        net = nn.Sequential(nn.Conv2d(3, 16, 7, 4, 3), nn.ReLU(), nn.Conv2d(16, 32, 3, 2, 1), nn.ReLU(), nn.AdaptiveAvgPool2d(4), nn.Flatten(),
                            nn.Linear(512, num_classes))
        with torch.no_grad():
            for prm in net.parameters():
                prm.copy_(torch.randn(prm.shape, generator=g) * (2.5 if prm.dim() == 2 else 0.15))
        self.model = net.to(device).eval()
        self.input_size = input_size
        self.use_hierarchical = use_hierarchical
        self.confidence_threshold = confidence_threshold
        self.enable_geographic_filter = False
        self.allowed_species = None
        self.hierarchy_thresholds = {"species": 0.6, "genus": 0.4, "family": 0.3, "order": 0.2, "class": 0.1}
        self.taxonomy = {str(i): {"common_name": f"species_{i}", "genus": f"genus_{i // 2}", "family": f"family_{i // 4}",
                                  "order": f"order_{i // 8}", "class": "Aves" if i % 2 else "Mammalia"} for i in range(num_classes)}

    def get_hierarchical_label(self, class_id: int, confidence: float):
        """same contract as src/species_classifier.py:168-233: the most specific rank the confidence supports, or (None, None)"""
        entry = self.taxonomy.get(str(class_id), {})
        if not self.use_hierarchical:
            return entry.get("common_name", f"species_{class_id}"), "species"
        for level, key in (("species", "common_name"), ("genus", "genus"), ("family", "family"), ("order", "order"), ("class", "class")):
            if confidence >= self.hierarchy_thresholds[level]:
                label = entry.get(key)
                return (label, level) if label else (None, None)
        return None, None


class StandInPipeline:
    """The attributes of `TwoStageDetectionPipeline` (src/two_stage_pipeline_yolox.py:63-91) that `BatchedStage2` reads, with
    stand-in classifiers for the reference's categories - for tests and bench.py only."""

    def __init__(self, device: str = "cuda:0", categories=("bird", "mammal"), min_crop_size: int = 64, crop_padding_percent: int = 20):
        from telescope_cam_detection_amd.coco_constants import CLASS_ID_TO_CATEGORY

        self.enable_species_classification = True
        self.class_id_to_category = CLASS_ID_TO_CATEGORY
        self.species_classifiers = {c: StandInSpeciesClassifier(device=device, seed=10 + i) for i, c in enumerate(categories)}
        self.min_crop_size = min_crop_size
        self.crop_padding_percent = crop_padding_percent
        self.rejected_taxonomic_levels = ["order", "class"]
        self.time_of_day_top_k = 5
        self.time_of_day_penalty = 0.3
        self.enhancer = None
