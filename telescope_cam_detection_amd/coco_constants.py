"""COCO class tables with the semantics of the reference's src/coco_constants.py:7-40:
80 contiguous class ids (what RT-DETR's deploy-mode post-processor emits), the wildlife subset
{0,14,15,16,21} used by `wildlife_only`, and the mammal ids used by `get_class_category`."""

# index == class id
COCO_CLASSES = [
    'person', 'bicycle', 'car', 'motorcycle', 'airplane', 'bus', 'train', 'truck',
    'boat', 'traffic light', 'fire hydrant', 'stop sign', 'parking meter', 'bench', 'bird', 'cat',
    'dog', 'horse', 'sheep', 'cow', 'elephant', 'bear', 'zebra', 'giraffe',
    'backpack', 'umbrella', 'handbag', 'tie', 'suitcase', 'frisbee', 'skis', 'snowboard',
    'sports ball', 'kite', 'baseball bat', 'baseball glove', 'skateboard', 'surfboard', 'tennis racket', 'bottle',
    'wine glass', 'cup', 'fork', 'knife', 'spoon', 'bowl', 'banana', 'apple',
    'sandwich', 'orange', 'broccoli', 'carrot', 'hot dog', 'pizza', 'donut', 'cake',
    'chair', 'couch', 'potted plant', 'bed', 'dining table', 'toilet', 'tv', 'laptop',
    'mouse', 'remote', 'keyboard', 'cell phone', 'microwave', 'oven', 'toaster', 'sink',
    'refrigerator', 'book', 'clock', 'vase', 'scissors', 'teddy bear', 'hair drier', 'toothbrush',
]

WILDLIFE_CLASSES = {0: "person", 14: "bird", 15: "cat", 16: "dog", 21: "bear"}
CLASS_ID_TO_CATEGORY = {14: "bird", 15: "mammal", 16: "mammal", 21: "mammal"}
MAMMAL_CLASS_IDS = [15, 16, 21]
