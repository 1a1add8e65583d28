"""Drop-in `RTDETRDetector` backed by the MI355X-native HIP engine (libmi355rtdetr.so).

Mirrors the reference class /root/reference/src/rtdetr_detector.py:26-425 - same constructor
arguments, method names, return schemas and error behaviour - so that the reference's
`InferenceEngine` (src/inference_engine_yolox.py:196-212, :554) and
`SharedInferenceCoordinator` (src/shared_inference_coordinator.py:250) can use it unchanged
(see INTEGRATION.md for the two-line wiring).  All arithmetic runs in the HIP library; there is no
PyTorch or CPU fallback: on a machine without the library or without a GPU `load_model()` logs the
reason and returns False, exactly like the reference does when upstream RT-DETR is missing (:73-76).
"""
from __future__ import annotations

import logging
import time
from typing import Any, Dict, List, Optional, Union

import numpy as np

from . import _capi
from .arch import ARCHS, Arch, arch_from_config_path
from .coco_constants import COCO_CLASSES, MAMMAL_CLASS_IDS, WILDLIFE_CLASSES
from .weights import fold_weights, pack_blob, synth_weights

logger = logging.getLogger(__name__)


class _DeviceModel:
    """Stands in for `detector.model` (an nn.Module in the reference): the engine only ever calls
    `.to(device)` on it (src/inference_engine_yolox.py:744) and checks it for None (:248)."""

    def __init__(self, engine: "_capi.Engine", device: str):
        self.engine = engine
        self.device = device

    def to(self, device):
        if str(device) != str(self.device):
            raise RuntimeError(f"the MI355X-native RT-DETR engine cannot move to {device!r}; it has no CPU path")
        return self

    def eval(self):
        return self


def _device_index(device: str) -> int:
    s = str(device)
    if s.startswith("cuda") or s.startswith("hip"):
        return int(s.split(":")[1]) if ":" in s else 0
    raise ValueError(f"unsupported device {device!r}: this detector only runs on an AMD GPU ('cuda:N')")


def load_state(model_path: str):
    """`torch.load(model_path)['ema']['module']` else `['model']` (src/rtdetr_detector.py:134-141).

    `synthetic:<arch>:<seed>` produces the seeded benchmark weights instead of reading a file.
    Returns (state dict in this build's un-fused naming, arch name or None).
    """
    if str(model_path).startswith("synthetic:"):
        _, arch_name, seed = str(model_path).split(":")
        return synth_weights(ARCHS[arch_name], int(seed)), arch_name
    # this build's own files, HF `RTDetrV2ForObjectDetection` checkpoints (.safetensors / .bin) and upstream `.pth` key
    # names are recognised and converted (checkpoint.py)
    from .checkpoint import load_foreign_state

    return load_foreign_state(model_path)


class RTDETRDetector:
    """RT-DETRv2 detector for Stage 1 detection, MI355X-native."""

    def __init__(
        self,
        config_path: str = "RT-DETR/rtdetrv2_pytorch/configs/rtdetrv2/rtdetrv2_r18vd_120e_coco.yml",
        model_path: str = "models/rtdetr/rtdetrv2_r18vd.pth",
        device: str = "cuda:0",
        conf_threshold: float = 0.25,
        input_size: tuple = (640, 640),
        wildlife_only: bool = True,
        # build-specific knobs (keyword-only in spirit; the reference never passes them)
        # "f16x3" (default): hi/lo fp16 pairs, three MFMAs per product - the engine held to the reference tolerance (1e-3 on scores,
        # 1e-2 px on boxes against fp32 eager, 640- and 1280-px frames, tests/test_gpu_parity.py); "bf16": ~1.5x faster, far outside that
        # tolerance (opt-in);
        # "fp32": exact fp32 MFMAs
        precision: str = "f16x3",
        max_batch: int = 8,
        use_graph: bool = True,
        profile: str = "latency",
        prepare_batches=None,
    ):
        self.config_path = config_path
        self.model_path = model_path
        self.device = device
        self.conf_threshold = conf_threshold
        self.nms_threshold = None           # written by update_settings (src/inference_engine_yolox.py:684); RT-DETR has no NMS
        self.input_size = input_size
        self.wildlife_only = wildlife_only
        self.precision = precision
        self.max_batch = max_batch
        self.use_graph = use_graph
        # "latency": this detector usually has the GPU to itself; "throughput": several detectors keep batches in flight on one
        # GPU (batching.BatchCoordinator with pipeline_depth > 1) - see rtd_config.profile in include/rtdetr_mi355.h
        self.profile = profile
        # batch sizes whose plan + hipGraph load_model builds up front (None: 1 and max_batch); a batcher declares every size it may form,
        # so that the serving path only ever replays a graph.  Other sizes still work: they are built on first use.
        self.prepare_batches = prepare_batches

        # The engine lives on the device named at load time.  The reference's degrade path later WRITES `detector.device = "cpu"`,
        # `detector.input_size = ...` and calls `detector.model.to("cpu")` (src/inference_engine_yolox.py:726-748); those writes must
        # not re-route an engine that stays where it is, so everything below uses the ordinal / size recorded by load_model.
        self._dev_index: Optional[int] = None
        self._engine_input_size: Optional[tuple] = None
        self.model: Optional[_DeviceModel] = None
        self.postprocessor = None
        self.transforms = None
        self.arch: Optional[Arch] = None
        self.last_check: Optional[dict] = None    # report of the load-time self check (load_model(verify=True))

    # ------------------------------------------------------------------ load
    # a checkpoint passes the load-time self check when at least this share of the fp32 engine's rows is matched within 1e-3 / 1e-2 px;
    # near-ties at the two top-k cuts may cost single rows on any engine (tests/test_gpu_parity.py x3_check), a checkpoint outside the
    # pair format's range costs most of them
    VERIFY_MIN_MATCH = 0.97

    def load_model(self, max_retries: int = 3, verify: bool = True) -> bool:
        """Build the device engine.  Contract of src/rtdetr_detector.py:60-204: never raises, True / False; transient failures
        (I/O, runtime errors) are retried with a doubling pause (1 s, 2 s, 4 s ...), anything else fails at once.

        `verify` (build-specific, default on; the default "f16x3" precision only): the real-weights guard.  The library already refuses
        a blob with NaN / Inf or with a folded filter beyond the fp16 pair format's range (RTD_E_WEIGHTS); here one built-in frame runs
        through the library's exact fp32 engine and through this engine's arithmetic with THESE weights (rtd_self_check) - the report is
        logged, and a checkpoint whose rows do not agree within the reference tolerance is refused (False) instead of serving finite,
        plausible, wrong boxes.  `self.last_check` keeps the report."""
        try:
            dev = _device_index(self.device)
        except ValueError as e:
            logger.error("RT-DETR (MI355X): %s", e)
            return False
        tries = max(1, int(max_retries))
        for k in range(tries):
            t0 = time.perf_counter()
            try:
                blob, arch = self._blob()
                engine = _capi.Engine(arch, blob, device=dev, precision=_capi.precision_code(self.precision), max_batch=self.max_batch,
                                      input_size=tuple(self.input_size), use_graph=self.use_graph,
                                      profile=_capi.PROFILE_THROUGHPUT if str(self.profile).lower() == "throughput" else _capi.PROFILE_LATENCY)
                try:
                    if verify and str(self.precision).lower() == "f16x3" and not self._verified(engine, blob):
                        engine.close()
                        return False
                    self._prepare(engine)
                except BaseException:
                    engine.close()                            # a retry must not find this attempt's weights, arenas and graphs still resident
                    raise
            except (RuntimeError, OSError) as e:              # the reference retries this set (:190-198); IOError is OSError
                if k + 1 == tries:
                    logger.error("RT-DETR (MI355X): giving up on %s after %d attempt(s): %s", self.model_path, tries, e, exc_info=True)
                    return False
                pause = 1 << k
                logger.warning("RT-DETR (MI355X): attempt %d of %d to build the engine failed (%s); next try in %d s", k + 1, tries, e, pause)
                time.sleep(pause)
                continue
            except Exception as e:
                logger.error("RT-DETR (MI355X): cannot build the engine from %s: %s", self.model_path, e, exc_info=True)
                return False
            self.arch = arch
            self._dev_index = dev
            self._engine_input_size = tuple(self.input_size)
            self.model = _DeviceModel(engine, self.device)
            logger.info("RT-DETR (MI355X): %s engine for %s on %s ready in %.1f s - weights %s, input %dx%d, up to %d frames per call, "
                        "%s arithmetic, score threshold %.2f, 80 COCO classes", arch.name, self.config_path, self.device,
                        time.perf_counter() - t0, self.model_path, self.input_size[0], self.input_size[1], self.max_batch, self.precision,
                        self.conf_threshold)
            return True
        return False

    def _blob(self):
        """(packed weight blob, Arch) for `model_path`: through the on-disk blob cache (weights.cached_blob) when the variant is known
        without reading the file - `synthetic:<arch>:<seed>`, or a config_path that names it - so that the ranks of a node fold a
        checkpoint once, not once each."""
        from .weights import cached_blob
        mp = str(self.model_path)
        arch = None
        if mp.startswith("synthetic:"):
            arch = ARCHS[mp.split(":")[1]]
        else:
            try:
                arch = arch_from_config_path(self.config_path)
            except ValueError:
                arch = None
        if arch is None:
            state, arch_name = load_state(self.model_path)
            return pack_blob(fold_weights(ARCHS[arch_name], state)), ARCHS[arch_name]

        other = {}

        def make_state():
            state, arch_name = load_state(self.model_path)
            if arch_name and arch_name != arch.name:          # the file knows better than the config's name (the precedence this class always had)
                other["state"], other["arch"] = state, ARCHS[arch_name]
                raise LookupError(arch_name)
            return state
        try:
            return cached_blob(arch, mp, make_state), arch
        except LookupError:
            return pack_blob(fold_weights(other["arch"], other["state"])), other["arch"]

    def _verified(self, engine, blob) -> bool:
        import torch

        try:
            rep = engine.self_check(blob)
        except torch.cuda.OutOfMemoryError as e:
            # the check needs two temporary bs-1 engines beside this one: when they do not fit, the detector is served unverified - loudly -
            # rather than refused (its own arenas are allocated; the caller's OOM handling covers what happens later)
            logger.warning("RT-DETR (MI355X): load-time self check of %s SKIPPED, not enough device memory for the two temporary engines (%s)",
                           self.model_path, str(e)[:160])
            self.last_check = None
            return True
        self.last_check = rep
        share = rep["rows_matched"] / max(1, rep["rows"])
        line = ("%d of %d rows of the fp32 engine matched within %.0e / %.0e px (worst %.1e / %.1e px), %d activations at the fp16 pair "
                "format's saturation value, largest folded filter value %.3g (%s)")
        args = (rep["rows_matched"], rep["rows"], rep["score_tol"], rep["box_tol_px"], rep["worst_score_err"], rep["worst_box_err_px"],
                rep["saturated_values"], rep["max_abs_filter"], rep["max_abs_filter_name"])
        if share < self.VERIFY_MIN_MATCH:
            logger.error("RT-DETR (MI355X): %s FAILS the load-time self check: " + line + ".  The f16x3 engine is not valid for these weights; "
                         "use precision='fp32'.", self.model_path, *args)
            return False
        if share < 1.0 or rep["saturated_values"] > 0:
            logger.warning("RT-DETR (MI355X): load-time self check of %s: " + line, self.model_path, *args)
        else:
            logger.info("RT-DETR (MI355X): load-time self check of %s: " + line, self.model_path, *args)
        return True

    def _prepare(self, engine) -> None:
        """Plan, arena and hipGraph of every declared batch size, now - the serving path then only replays.  A size whose arena does not
        fit is left to its first use (where the caller's OOM handling applies, src/inference_engine_yolox.py:607-623): it must not
        fail the load of a detector whose smaller batches work."""
        import torch

        sizes = self.prepare_batches if self.prepare_batches is not None else (1, self.max_batch)
        for b in sorted({int(b) for b in sizes if 1 <= int(b) <= self.max_batch}):
            try:
                engine.prepare(b)
            except torch.cuda.OutOfMemoryError as e:
                logger.warning("RT-DETR (MI355X): batch size %d not prepared at load time (%s); it will be built on first use", b, str(e)[:160])
                break

    # ------------------------------------------------------------------ helpers
    @staticmethod
    def _as_frame(img):
        """HWC uint8 BGR ndarray, or a torch tensor (host or device) - src/rtdetr_detector.py:216-222."""
        if isinstance(img, np.ndarray):
            return img, False
        import torch

        if isinstance(img, torch.Tensor):
            if img.is_cuda:
                return img.contiguous(), True
            return img.numpy(), False
        raise TypeError(f"unsupported frame type {type(img)}")

    def _format(self, rows: np.ndarray) -> List[Dict[str, Any]]:
        """rows -> the dict schema of src/rtdetr_detector.py:290-301 (threshold / wildlife filter were
        applied in the library in the same order as :271,:277)."""
        detections = []
        for r in rows:
            class_id = int(r["class_id"])
            x1, y1, x2, y2 = float(r["x1"]), float(r["y1"]), float(r["x2"]), float(r["y2"])
            class_name = COCO_CLASSES[class_id] if class_id < len(COCO_CLASSES) else f"class_{class_id}"
            detections.append({
                "class_id": class_id,
                "class_name": class_name,
                "confidence": float(r["score"]),
                "bbox": {"x1": x1, "y1": y1, "x2": x2, "y2": y2, "area": int((x2 - x1) * (y2 - y1))},
            })
        return detections

    def _order_after_producer(self, eng) -> None:
        """Device-resident frames were produced on torch's current stream (a decode kernel, a slice made contiguous by `_as_frame`);
        the engine runs on its own non-blocking stream, which synchronises with nothing by itself: make it wait for that work."""
        import torch

        # an event of the LIBRARY is recorded on torch's stream and waited for on the engine's: torch never sees the engine's stream
        eng.wait_stream(torch.cuda.current_stream(torch.device("cuda", self._dev_index)).cuda_stream)

    def _infer(self, frames: list) -> List[np.ndarray]:
        arrs, on_dev = [], []
        for f in frames:
            a, d = self._as_frame(f)
            arrs.append(a)
            on_dev.append(d)
        if any(on_dev) and not all(on_dev):
            arrs = [a.cpu().numpy() if d else a for a, d in zip(arrs, on_dev)]
            on_dev = [False] * len(arrs)
        eng = self.model.engine
        if all(on_dev) and on_dev:
            self._order_after_producer(eng)
        out: List[np.ndarray] = []
        for i in range(0, len(arrs), eng.max_batch):     # larger lists run as several device batches
            out += eng.infer(arrs[i:i + eng.max_batch], self.conf_threshold, self.wildlife_only, on_device=all(on_dev))
        return out

    # ------------------------------------------------------------------ API of the reference class
    def preprocess(self, img: Union[np.ndarray, "torch.Tensor"]) -> tuple:
        """(preprocessed [1,3,H,W] fp32 tensor on the device, [[w, h]] tensor) - src/rtdetr_detector.py:206-236: BGR -> RGB, the PIL-exact
        antialiased stretch to the engine's input size, / 255.  Provided for interface parity; `detect` does not call it (the network's
        first kernel reads the uint8 frame directly).  One small launch on the device (rtd_preprocess): no forward pass, no host round trip."""
        import torch

        a, on_dev = self._as_frame(img)
        if not on_dev:
            a = np.ascontiguousarray(a)
        eng = self.model.engine
        if on_dev:
            self._order_after_producer(eng)
        dev = torch.device("cuda", self._dev_index)
        ih, iw = self._engine_input_size
        x = torch.empty((1, 3, ih, iw), dtype=torch.float32, device=dev)
        torch.cuda.current_stream(dev).synchronize()           # x's memory may be a recycled block that torch's stream is still using
        eng.preprocess_into(a, on_dev, x.data_ptr())
        h, w = a.shape[:2]
        return x, torch.tensor([[w, h]], device=dev)

    def detect(self, frame: Union[np.ndarray, "torch.Tensor"]) -> List[Dict[str, Any]]:
        if self.model is None:
            logger.error("Model not loaded")
            return []
        return self._format(self._infer([frame])[0])

    def detect_batch(self, frames: List[Union[np.ndarray, "torch.Tensor"]]) -> List[List[Dict[str, Any]]]:
        if self.model is None:
            logger.error("Model not loaded")
            return [[] for _ in frames]
        if not frames:
            return []
        return [self._format(rows) for rows in self._infer(list(frames))]

    # ------------------------------------------------------------------ pipelined use (batching.BatchCoordinator, depth > 1)
    def detect_batch_async(self, frames: List[Union[np.ndarray, "torch.Tensor"]]):
        """Enqueue one batch (<= max_batch frames) on this detector's stream and return a ticket at once; `detect_batch_collect`
        blocks for that batch only.  With two detectors a coordinator keeps two batches in flight: one batch's kernels fill the
        CUs the other's small grids leave idle (bench.py --streams 2 / 3: +33 % / +46 % frames/s on one MI355X).

        Everything between the two calls is the library's own HIP work (rtd_infer_async: pinned staging + one DMA for host frames;
        rtd_collect: D2H of the result block): no torch stream, event or allocator takes part, so this path cannot interact with
        whatever else the process does through torch."""
        if self.model is None:
            raise RuntimeError("Model not loaded")
        eng = self.model.engine
        if len(frames) > eng.max_batch:
            raise ValueError(f"detect_batch_async takes at most max_batch={eng.max_batch} frames")
        arrs, on_dev = [], []
        for f in frames:
            a, d = self._as_frame(f)
            arrs.append(a)
            on_dev.append(d)
        if any(on_dev) and not all(on_dev):
            arrs = [a.cpu().numpy() if d else a for a, d in zip(arrs, on_dev)]
            on_dev = [False] * len(arrs)
        dev = bool(on_dev) and all(on_dev)
        if dev:
            self._order_after_producer(eng)              # device frames: whatever produced them comes first
        if arrs:
            eng.infer_async(arrs, on_device=dev)
        return {"n": len(arrs), "frames": arrs if dev else None, "conf": self.conf_threshold, "wildlife": self.wildlife_only}

    def detect_batch_collect(self, ticket) -> List[List[Dict[str, Any]]]:
        n = ticket["n"]
        if n == 0:
            return []
        rows, counts = self.model.engine.collect(ticket["conf"], ticket["wildlife"])
        ticket["frames"] = None
        return [self._format(rows[i, : counts[i]]) for i in range(n)]

    def is_wildlife_relevant(self, class_id: int) -> bool:
        return class_id in WILDLIFE_CLASSES

    def get_class_category(self, class_id: int) -> str:
        if class_id == 0:
            return "person"
        elif class_id == 14:
            return "bird"
        elif class_id in MAMMAL_CLASS_IDS:
            return "mammal"
        else:
            return "other"
