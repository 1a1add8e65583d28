"""Golden vectors for the host-side rows, produced by the REFERENCE's own code (test infrastructure; build container only).

Three reference modules import nothing but the standard library and therefore load here although the detector itself does not
(torchvision / cv2 / upstream RT-DETR are absent):

  /root/reference/src/bbox_utils.py:12-120                  ensure_valid_bbox, validate_bbox_coords, is_valid_bbox
  /root/reference/src/coco_constants.py:7-40                COCO_CLASSES, WILDLIFE_CLASSES, CLASS_ID_TO_CATEGORY, MAMMAL_CLASS_IDS
  /root/reference/src/shared_inference_coordinator.py:27-338  SharedInferenceCoordinator
  /root/reference/src/memory_manager.py:158-248             MemoryManager.reduce_memory_usage / handle_oom_error (imports torch, which
                                                            is present; without a GPU it runs in its "CPU mode" and the recommendations
                                                            do not depend on the device)

They are loaded from where they lie (never copied), driven with seeded inputs / the scripted scenarios of
`tests/host_scenarios.py`, and only DATA is written: `tests/golden/host_bbox.json`, `host_coco.json`, `host_coordinator.json`,
`host_engine_sequence.json`.
`/root/reference` does not exist on the GPU box; the tests read the JSON files only.

    python oracle/make_host_golden.py            # rewrite the four files
    python oracle/make_host_golden.py --check    # regenerate in memory and compare with the committed files
"""
import importlib.util
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SRC = "/root/reference/src"
OUT = os.path.join(ROOT, "tests", "golden")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def ref_module(name: str):
    path = os.path.join(REF_SRC, name + ".py")
    spec = importlib.util.spec_from_file_location("reference_" + name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = mod          # dataclasses resolve their module through sys.modules
    spec.loader.exec_module(mod)
    return mod


def bbox_inputs():
    """>= 200 boxes: the hand-picked corner cases first, then seeded draws mixing ints and floats, inverted corners, extents below
    the minimum, negative and huge coordinates."""
    cases = [
        (10, 20, 30, 40, 1), (30, 40, 10, 20, 1), (10.5, 20.25, 10.5, 20.25, 1), (5, 5, 5, 9, 1), (5, 5, 9, 5, 1),
        (0, 0, 0, 0, 1), (-10, -20, -30, -40, 1), (-3.5, 2, 4, -7.25, 1), (100, 100, 100.5, 100.999, 1), (1, 1, 2, 2, 1),
        (1, 1, 2, 2, 2), (1, 1, 2, 2, 0), (7, 7, 7, 7, 0), (1e6, 2e6, 1e6 + 0.25, 2e6 + 3, 1), (0.1, 0.2, 0.30000000000000004, 0.4, 1),
        (3, 4, 3, 4, 5), (9, 1, 2, 8, 3), (2.5, 2.5, 3.0, 9.5, 2.5), (640, 0, 0, 640, 1), (1919.9999, 1079.5, 1920, 1080, 1),
    ]
    rng = random.Random(20261005)

    def coord(kind):
        if kind == 0:
            return rng.randint(-400, 2000)
        if kind == 1:
            return round(rng.uniform(-400.0, 2000.0), rng.choice((0, 1, 3, 6)))
        return rng.uniform(-150.0, 700.0)

    while len(cases) < 260:
        kind = rng.randint(0, 2)
        x1, y1 = coord(kind), coord(rng.randint(0, 2))
        shape = rng.randint(0, 4)
        if shape == 0:          # ordinary box
            x2, y2 = x1 + abs(coord(kind)) * 0.3 + 2, y1 + abs(coord(kind)) * 0.3 + 2
        elif shape == 1:        # inverted in x, y or both
            x2, y2 = x1 - rng.choice((0, 1, 7.5, 120)), y1 + rng.choice((-40, -0.5, 3, 90.25))
        elif shape == 2:        # thinner than the minimum extent
            x2, y2 = x1 + rng.choice((0, 0.25, 0.999, 1)), y1 + rng.choice((0, 0.5, 1, 1.0000001))
        elif shape == 3:        # mixed int / float corners
            x2, y2 = int(x1) + rng.randint(-5, 60), float(y1) + rng.uniform(-5, 60)
        else:
            x2, y2 = coord(rng.randint(0, 2)), coord(rng.randint(0, 2))
        cases.append((x1, y1, x2, y2, rng.choice((1, 1, 1, 1, 0, 2, 8, 0.5))))
    return cases


def make_bbox(bu):
    rows = []
    for x1, y1, x2, y2, m in bbox_inputs():
        box = {"x1": x1, "y1": y1, "x2": x2, "y2": y2}
        rows.append({
            "bbox": box, "min_size": m,
            "ensure_valid_bbox": bu.ensure_valid_bbox(dict(box), m),
            "validate_bbox_coords": list(bu.validate_bbox_coords(x1, y1, x2, y2, m)),
            "is_valid_bbox": bu.is_valid_bbox(dict(box), m),
        })
    # a detection as `RTDETRDetector.detect` emits it (int `area`, /root/reference/src/rtdetr_detector.py:290-301): the extra key is dropped,
    # the area comes back un-truncated (src/bbox_utils.py:46-58)
    det_box = {"x1": 10.25, "y1": 20.5, "x2": 110.75, "y2": 220.125, "area": int((110.75 - 10.25) * (220.125 - 20.5))}
    rows.append({"bbox": det_box, "min_size": 1, "ensure_valid_bbox": bu.ensure_valid_bbox(dict(det_box)),
                 "validate_bbox_coords": list(bu.validate_bbox_coords(10.25, 20.5, 110.75, 220.125)), "is_valid_bbox": bu.is_valid_bbox(det_box)})
    return {"source": "src/bbox_utils.py", "defaults": {"min_size": 1}, "is_valid_bbox_on_malformed": [bu.is_valid_bbox({}), bu.is_valid_bbox(None)],
            "vectors": rows}


def make_coco(cc):
    return {
        "source": "src/coco_constants.py",
        "COCO_CLASSES": list(cc.COCO_CLASSES),
        "WILDLIFE_CLASSES": {str(k): v for k, v in cc.WILDLIFE_CLASSES.items()},
        "CLASS_ID_TO_CATEGORY": {str(k): v for k, v in cc.CLASS_ID_TO_CATEGORY.items()},
        "MAMMAL_CLASS_IDS": list(cc.MAMMAL_CLASS_IDS),
    }


def make_coordinator(sic):
    from tests import host_scenarios
    out = host_scenarios.run_all(sic.SharedInferenceCoordinator)
    out["source"] = "src/shared_inference_coordinator.py"
    return out


def make_engine_sequence(mm):
    """What the unchanged caller's memory manager tells `InferenceEngine._apply_degradation` (src/inference_engine_yolox.py:706-748) to do:
    the recommendations of four OOM events in a row (:609-611 -> src/memory_manager.py:207-248) and of the three pressure levels
    (:600-604 -> src/memory_manager.py:158-205).  tests/test_engine_sequence.py replays them on the real detector."""
    m = mm.MemoryManager(device="cuda:0")
    ooms = []
    for _ in range(4):
        rec = m.handle_oom_error()
        ooms.append({"recommendations": rec, "oom_events": m.oom_events, "degradation_level": m.degradation_level})
    m.record_recovery()
    m2 = mm.MemoryManager(device="cuda:0")
    levels = {lvl.name: m2.reduce_memory_usage(lvl) for lvl in (mm.MemoryPressure.HIGH, mm.MemoryPressure.CRITICAL, mm.MemoryPressure.EXTREME)}
    return {"source": "src/memory_manager.py", "handle_oom_error": ooms, "recoveries_after_one_record": m.recoveries, "reduce_memory_usage": levels}


def generate():
    return {
        "host_bbox.json": make_bbox(ref_module("bbox_utils")),
        "host_coco.json": make_coco(ref_module("coco_constants")),
        "host_coordinator.json": make_coordinator(ref_module("shared_inference_coordinator")),
        "host_engine_sequence.json": make_engine_sequence(ref_module("memory_manager")),
    }


def main():
    files = generate()
    check = "--check" in sys.argv
    bad = 0
    for name, data in files.items():
        path = os.path.join(OUT, name)
        text = json.dumps(data, indent=1, sort_keys=True) + "\n"
        if check:
            same = os.path.exists(path) and open(path).read() == text
            print(f"{name}: {'identical' if same else 'DIFFERS'}")
            bad += not same
        else:
            with open(path, "w") as f:
                f.write(text)
            print(f"wrote {path} ({len(text)} bytes)")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
