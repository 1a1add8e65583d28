"""The unchanged caller's call sequence on the real detector, and file-format checkpoints through the detector (f4).

`InferenceEngine` itself cannot be imported (yolox / cv2 absent), so `EngineReplay` restates - with the line it follows beside every
step - ONLY the calls the engine makes on its detector object: construct + `load_model` with the CPU-fallback twin
(/root/reference/src/inference_engine_yolox.py:196-272), `detect` (:542-562), OOM -> `handle_oom_error` -> `_apply_degradation`
-> one retry (:607-627), the degrade writes (:706-748).  What the memory manager recommends at each OOM event is NOT restated: it
is read from `tests/golden/host_engine_sequence.json`, which `oracle/make_host_golden.py` recorded from the reference's own
`MemoryManager` (src/memory_manager.py:207-248).
"""
import json
import os

import numpy as np
import pytest
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REF_KW = ("config_path", "model_path", "device", "conf_threshold", "input_size", "wildlife_only")     # :199-206 - nothing else is passed


def recorded():
    with open(os.path.join(GOLDEN, "host_engine_sequence.json")) as f:
        return json.load(f)


class EngineReplay:
    """The detector-facing part of `InferenceEngine`, step by step."""

    def __init__(self, detector_cls, **kw):
        assert set(kw) == set(REF_KW)
        self.detector_cls, self.kw = detector_cls, dict(kw)
        self.device = kw["device"]
        self.input_size = kw["input_size"]
        self.detector = None
        self.oom_events = 0
        self.degradation_active = False
        self.log = []

    def load_model(self) -> bool:                                   # :174-284
        try:
            try:
                self.detector = self.detector_cls(**self.kw)        # :199-206
                if not self.detector.load_model():                  # :209-210
                    raise RuntimeError("Failed to load RT-DETR model")
            except (torch.cuda.OutOfMemoryError, RuntimeError) as e:  # :233
                self.log.append(("gpu_load_failed", str(e)))
                torch.cuda.empty_cache()                            # :239-240
                self.device = "cpu"                                 # :243
                self.detector = self.detector_cls(**{**self.kw, "device": "cpu"})   # :245-253
                if not self.detector.load_model():                  # :266-268
                    self.log.append(("cpu_load_failed",))
                    return False
            return True
        except Exception as e:                                      # :280-284
            self.log.append(("load_raised", repr(e)))
            return False

    def apply_degradation(self, rec):                               # :706-748
        if not rec:
            return
        if rec.get("clear_cache", False):
            torch.cuda.empty_cache()                                # memory_manager.clear_cache, src/memory_manager.py:291-298
        if rec.get("reduce_input_size", False):
            suggested = rec.get("suggested_input_size")
            suggested = tuple(suggested) if suggested else suggested   # JSON has no tuples; the reference hands a tuple (:232)
            if suggested and suggested != self.input_size:
                self.input_size = suggested
                if self.detector:
                    self.detector.input_size = suggested            # :730
                    if hasattr(self.detector, "exp") and self.detector.exp is not None:   # :731-732
                        self.detector.exp.test_size = suggested
                self.degradation_active = True
        if rec.get("cpu_fallback", False) and self.device != "cpu":  # :736-748
            self.device = "cpu"
            if self.detector:
                try:
                    self.detector.device = "cpu"                    # :743
                    self.detector.model.to("cpu")                   # :744
                    self.degradation_active = True
                    self.log.append(("moved_to_cpu",))
                except Exception as e:                              # :747-748
                    self.log.append(("move_to_cpu_failed", str(e)))

    def run_inference(self, frame):                                 # :564-627 without the sparse / motion gates
        try:
            return self.detector.detect(frame)                      # :554
        except torch.cuda.OutOfMemoryError as e:                    # :607
            self.log.append(("oom", str(e)[:60]))
            rec = recorded()["handle_oom_error"][min(self.oom_events, 3)]["recommendations"]   # :610
            self.oom_events += 1
            torch.cuda.empty_cache()                                # handle_oom_error itself: src/memory_manager.py:217-222
            torch.cuda.synchronize()
            self.apply_degradation(rec)                             # :611
            try:
                return self.detector.detect(frame)                  # :620 (the 1 s `wait_time` pause is skipped)
            except Exception as retry_e:                            # :621-623
                self.log.append(("retry_failed", repr(retry_e)))
                return []
        except Exception as e:                                      # :625-627
            self.log.append(("error", repr(e)))
            return []


def hog_device_memory():
    """Fill the GPU through torch's caching allocator and hand the blocks back to the CACHE (not to the driver): the state in
    which the reference's `torch.cuda.empty_cache()` on OOM actually frees something."""
    free, _ = torch.cuda.mem_get_info(0)
    blocks = []
    try:
        blocks.append(torch.empty(max(free - (768 << 20), 1 << 20), dtype=torch.uint8, device="cuda:0"))
        for _ in range(64):
            blocks.append(torch.empty(16 << 20, dtype=torch.uint8, device="cuda:0"))
    except torch.cuda.OutOfMemoryError:
        pass
    left = torch.cuda.mem_get_info(0)[0]
    del blocks
    return left


def test_recorded_recommendations_are_what_the_replay_expects():
    g = recorded()
    recs = [e["recommendations"] for e in g["handle_oom_error"]]
    assert [r["cpu_fallback"] for r in recs] == [False, False, True, True] and [e["oom_events"] for e in g["handle_oom_error"]] == [1, 2, 3, 4]
    assert all(r["clear_cache"] and r["reduce_input_size"] and r["suggested_input_size"] == [640, 640] and r["wait_time"] == 1.0 for r in recs)
    assert g["reduce_memory_usage"]["EXTREME"]["suggested_input_size"] == [640, 640] and not g["reduce_memory_usage"]["HIGH"]["reduce_input_size"]


def test_cpu_twin_of_the_detector_refuses_to_load():
    """Deliberate deviation: this detector has no CPU path, so the engine's CPU-fallback twin (:245-268) gets
    `load_model() == False` and `InferenceEngine.load_model` returns False instead of degrading to a CPU model.  Runs without a GPU."""
    from telescope_cam_detection_amd.rtdetr_detector import RTDETRDetector
    twin = RTDETRDetector(config_path="rtdetrv2_r18vd_120e_coco.yml", model_path="synthetic:r18:0", device="cpu", conf_threshold=0.25,
                          input_size=(640, 640), wildlife_only=True)
    assert twin.load_model() is False and twin.model is None
    assert twin.detect(np.zeros((8, 8, 3), np.uint8)) == []         # src/rtdetr_detector.py:248-250
    assert twin.detect_batch([np.zeros((8, 8, 3), np.uint8)] * 2) == [[], []]   # :317-319


@pytest.mark.gpu
def test_the_engines_call_sequence_on_the_real_detector():
    from oracle import rtdetr_oracle as orc
    from telescope_cam_detection_amd import _capi
    from telescope_cam_detection_amd.arch import ARCHS
    from telescope_cam_detection_amd.rtdetr_detector import RTDETRDetector
    from telescope_cam_detection_amd.synth import scene_frame
    from tests.util import weights_for

    size = (320, 320)
    kw = dict(config_path="RT-DETR/rtdetrv2_pytorch/configs/rtdetrv2/rtdetrv2_r18vd_120e_coco.yml", model_path="synthetic:r18:0",
              device="cuda:0", conf_threshold=0.05, input_size=size, wildlife_only=False)
    eng = EngineReplay(RTDETRDetector, **kw)

    # 1. construct + load_model on the GPU (:199-210)
    assert eng.load_model() is True and eng.log == [] and eng.device == "cuda:0"
    det = eng.detector
    engine0 = det.model.engine
    assert det.model is not None and det.device == "cuda:0" and tuple(det.input_size) == size and det.nms_threshold is None
    assert not hasattr(det, "exp") or det.exp is None               # :731 is skipped for this detector
    st = engine0.stats()
    assert st["failed_calls"] == 0 and st["graphs"] >= 1            # batch 1 and max_batch were planned and graphed by load_model

    # 2. detect on a camera frame (:554): the dict schema, against the oracle
    small = scene_frame(11, 360, 480)
    first = eng.run_inference(small)
    arch = ARCHS["r18"]
    x, hw = orc.preprocess(small, size)
    ol, ob, osc = orc.model_forward(arch, weights_for(arch, 0), x, [hw])
    want = orc.format_detections(ol[0], ob[0], osc[0], 0.05, False)
    assert len(first) == len(want) > 0
    for a, b in zip(first, want):
        assert a["class_id"] == b["class_id"] and a["class_name"] == b["class_name"] and abs(a["confidence"] - b["confidence"]) <= 1e-3
        assert all(abs(a["bbox"][k] - b["bbox"][k]) <= 1e-2 for k in ("x1", "y1", "x2", "y2")) and isinstance(a["bbox"]["area"], int)

    # 2b. hot-reload of the settings while serving (update_settings, :672-685): the engine WRITES detector.conf_threshold / nms_threshold;
    #     the next detect filters at the new threshold (a prefix of the old answer: rows are in descending score order), NMS has no meaning here
    scores = [d["confidence"] for d in first]
    cut = float(np.median(scores))
    det.conf_threshold = cut                                          # :678
    det.nms_threshold = 0.45                                          # :684
    hot = eng.run_inference(small)
    assert hot == [d for d in first if d["confidence"] >= cut] and 0 < len(hot) < len(first) and det.nms_threshold == 0.45
    det.conf_threshold = kw["conf_threshold"]
    assert eng.run_inference(small) == first

    # 3. a 4K frame arrives while the rest of the process (torch's cache) holds the GPU: the frame's staging buffers cannot be
    #    allocated -> RTD_E_OOM -> torch.cuda.OutOfMemoryError out of detect (:607) -> empty_cache, degrade writes, ONE retry (:609-620)
    big = scene_frame(12, 2160, 3840)
    left = hog_device_memory()
    assert left < big.nbytes, f"{left} bytes still free: the hog did not fill the GPU"
    after_oom = eng.run_inference(big)
    assert [e[0] for e in eng.log] == ["oom"] and eng.oom_events == 1
    assert len(after_oom) > 0                                        # the retry, after empty_cache, ran
    # expected detector state after the first OOM event: input_size WRITTEN to the suggestion, engine untouched
    assert det.input_size == (640, 640) and eng.degradation_active and det.device == "cuda:0" and det.model.engine is engine0
    assert det._engine_input_size == size
    st = engine0.stats()
    assert st["failed_calls"] == 1 and st["last_error_code"] == _capi.RTD_E_OOM and st["in_flight"] == 0
    assert eng.run_inference(big) == after_oom and eng.run_inference(small) == first   # same handle, same answers as before the event

    # 4. OOM events two and three: the third recommends the CPU fallback (:736-748).  `model.to("cpu")` raises, the caller logs it
    #    (:747-748) and keeps using the detector, which stays on its GPU
    for k in (2, 3):
        huge = np.tile(big, (2 * (k - 1), 2, 1))                     # 4320 x 7680, then 8640 x 7680: each needs staging 75+ MB beyond what the handle holds
        left = hog_device_memory()
        assert left < 32 << 20
        got = eng.run_inference(huge)
        assert isinstance(got, list) and len(got) > 0 and eng.oom_events == k
        del huge
    assert [e[0] for e in eng.log] == ["oom", "oom", "oom", "move_to_cpu_failed"]
    assert det.device == "cpu" and eng.device == "cpu" and det.model.engine is engine0
    assert eng.run_inference(small) == first
    assert engine0.stats()["failed_calls"] == 3

    # 5. what the engine would do had the GPU load failed (:233-270): the CPU twin does not load, load_model() is False
    eng2 = EngineReplay(RTDETRDetector, **{**kw, "model_path": "/nonexistent/rtdetrv2_r18vd.pth"})

    class Quick(RTDETRDetector):
        def load_model(self, max_retries=1):                        # the engine calls load_model() bare; keep the 1-2-4 s pauses out of the test
            return super().load_model(max_retries=1)
    eng2.detector_cls = Quick
    assert eng2.load_model() is False
    assert [e[0] for e in eng2.log] == ["gpu_load_failed", "cpu_load_failed"] and eng2.device == "cpu" and eng2.detector.model is None
    assert eng2.detector.detect(small) == []


def _engine_rows(model_path, config_path, frames, size, **kw):
    from telescope_cam_detection_amd.rtdetr_detector import RTDETRDetector
    det = RTDETRDetector(config_path=config_path, model_path=model_path, device="cuda:0", conf_threshold=0.0, input_size=size,
                         wildlife_only=False, max_batch=len(frames), **kw)
    assert det.load_model(max_retries=1) is True
    out = det.model.engine.infer_raw(frames)
    dets = det.detect_batch(frames)
    det.model.engine.close()
    return out, dets


@pytest.mark.gpu
@pytest.mark.parametrize("aname", ["r18", "r50"])
def test_file_format_checkpoints_load_through_the_detector_bit_for_bit(aname, tmp_path):
    """f4 on the GPU: the seeded weights written (i) as an upstream-layout `.pth` under `ckpt['ema']['module']` and (ii) under
    `ckpt['model']` (/root/reference/src/rtdetr_detector.py:134-141), (iii) as an HF-layout `.safetensors`, (iv) in this build's own
    format - each loaded through `RTDETRDetector(model_path=...)` - must give the engine of `synthetic:<arch>:<seed>` bit for bit.
    (The upstream KEY TABLE itself stays unverifiable offline - no upstream checkpoint exists here; this exercises the path: file ->
    layout sniffing -> arch from the shapes -> in_proj split -> BN / RepVGG folding -> blob -> device.)"""
    st = pytest.importorskip("safetensors.torch")
    from telescope_cam_detection_amd import checkpoint as ck
    from telescope_cam_detection_amd.arch import ARCHS
    from telescope_cam_detection_amd.synth import scene_frame
    from telescope_cam_detection_amd.weights import save_weights, synth_weights
    arch = ARCHS[aname]
    seed = 4
    w = synth_weights(arch, seed)
    size = (320, 320)
    frames = [scene_frame(70, 300, 400), scene_frame(71, 320, 320)]
    cfg = f"RT-DETR/rtdetrv2_pytorch/configs/rtdetrv2/rtdetrv2_{aname}vd_120e_coco.yml"

    up, fused = {}, {}
    for mine, key in ck.upstream_key_map(arch).items():
        if "#" in key:
            base, part = key.split("#")
            fused.setdefault(base, {})[part] = w[mine]
        else:
            up[key] = w[mine]
    for base, parts in fused.items():
        up[base] = torch.cat([parts["q"], parts["k"], parts["v"]], 0)
    files = {}
    files["upstream_ema"] = os.path.join(tmp_path, f"rtdetrv2_{aname}vd_ema.pth")
    torch.save({"ema": {"module": {"module." + k: v for k, v in up.items()}}, "last_epoch": 119}, files["upstream_ema"])
    files["upstream_model"] = os.path.join(tmp_path, f"rtdetrv2_{aname}vd_model.pth")
    torch.save({"model": up}, files["upstream_model"])
    files["hf_safetensors"] = os.path.join(tmp_path, "model.safetensors")
    st.save_file({hf: w[mine].contiguous() for mine, hf in ck.hf_key_map(arch).items()}, files["hf_safetensors"])
    files["native"] = os.path.join(tmp_path, "native.pth")
    save_weights(files["native"], arch, w)

    (rl, rb, rs), rdets = _engine_rows(f"synthetic:{aname}:{seed}", cfg, frames, size)
    assert np.isfinite(rs).all() and len(rdets[0]) == 300
    for kind, path in files.items():
        (l, b, s), dets = _engine_rows(path, cfg, frames, size)
        assert np.array_equal(l, rl) and np.array_equal(b, rb) and np.array_equal(s, rs), kind
        assert dets == rdets, kind
