"""Cross-camera batching for RT-DETR.

The reference batches only YOLOX: `_initialize_shared_coordinator` returns None for every other detector
type (/root/reference/main.py:793-797), although its `SharedInferenceCoordinator` only needs an object with
`detect_batch(frames)` (src/shared_inference_coordinator.py:250).  The MI355X detector provides that contract
(one HIP graph replay per batch), so this module

* builds the coordinator's detector from the reference's own config keys (`detection.rtdetr.*`,
  `detection.batching.*`, config/config.yaml:74,90-99,130-135,511-517), and
* supplies `BatchCoordinator`, this build's own batcher - a depth-N pipeline of detector lanes (N batches in flight, results in
  submission order; N = 1 is the plain case) behind the surface the unchanged callers use (`start/stop`,
  `infer_async(frame, callback, camera_id)`, `get_stats()` keys; evicted / failed / never-run frames are answered with []) -
  used for pipeline_depth > 1 and whenever the reference package is not importable, and
* `install(system_cls)`: the three-line hook a maintainer adds so `main.py` stays untouched.

The engines (src/inference_engine_yolox.py:341-350) call `coordinator.infer_async(frame, callback, camera_id)`
and never look at the detector, so nothing else changes.
"""
from __future__ import annotations

import logging
import threading
import time
from collections import deque
from typing import Any, Callable, Dict, List, Optional

logger = logging.getLogger(__name__)


class _Ask:
    """one frame waiting for its answer"""
    __slots__ = ("frame", "answer", "t_in", "camera_id")

    def __init__(self, frame, answer, camera_id):
        self.frame, self.answer, self.camera_id = frame, answer, camera_id
        self.t_in = time.monotonic()

    def reply(self, detections) -> None:
        try:
            self.answer(detections)
        except Exception as e:                       # a camera's callback must never take the pipeline down
            logger.error(f"result callback of camera {self.camera_id} raised: {e}")


class _Inbox:
    """Bounded FIFO between the camera threads and the batch former.  Full -> the OLDEST ask is evicted and handed back to the
    caller (who answers it with []): stale frames lose, like the reference's queue (src/shared_inference_coordinator.py:149-164)."""

    def __init__(self, depth: int):
        self.depth = depth
        self._q: deque = deque()
        self._cv = threading.Condition()
        self.closed = False

    def put(self, ask: _Ask) -> Optional[_Ask]:
        with self._cv:
            evicted = self._q.popleft() if len(self._q) >= self.depth else None
            self._q.append(ask)
            self._cv.notify()
        return evicted

    def take(self, limit: int, patience: float) -> List[_Ask]:
        """Block for the first ask, then gather up to `limit`, waiting at most `patience` seconds for stragglers."""
        with self._cv:
            while not self._q and not self.closed:
                self._cv.wait(0.1)
            if self.closed:
                return []
            got = [self._q.popleft()]
            t_end = time.monotonic() + patience
            while len(got) < limit:
                if self._q:
                    got.append(self._q.popleft())
                    continue
                left = t_end - time.monotonic()
                if left <= 0 or self.closed:
                    break
                self._cv.wait(left)
            return got

    def close(self) -> List[_Ask]:
        with self._cv:
            self.closed = True
            rest = list(self._q)
            self._q.clear()
            self._cv.notify_all()
        return rest

    def reopen(self) -> None:
        with self._cv:
            self.closed = False

    def __len__(self):
        return len(self._q)


class _Lane:
    """One detector of the pipeline.  `begin` hands it a batch, `finish` returns that batch's results; a lane holds at most one batch.
    Detectors with the pipelined pair (RTDETRDetector.detect_batch_async / detect_batch_collect: the work runs on the detector's own
    HIP stream between the two calls) overlap with the other lanes; a plain `detect_batch` object simply does its work in `finish`."""

    REBUILD_AFTER = 3        # consecutive ENGINE failures after which the lane rebuilds its detector (a wedged engine must not be reused forever)
    MAX_REBUILDS = 3         # ... at most this often: a detector that keeps failing after three fresh engines is not helped by a fourth

    def __init__(self, detector: Any):
        self.detector = detector
        self.idle = threading.Semaphore(1)
        self._overlapped = hasattr(detector, "detect_batch_async") and hasattr(detector, "detect_batch_collect")
        self._held = None
        self.fail_streak = 0
        self.rebuilds = 0
        self.rebuild_seconds = 0.0

    def begin(self, frames: list) -> None:
        self._held = self.detector.detect_batch_async(frames) if self._overlapped else frames

    def finish(self) -> list:
        held, self._held = self._held, None
        return self.detector.detect_batch_collect(held) if self._overlapped else self.detector.detect_batch(held)

    def succeeded(self) -> None:
        self.fail_streak = 0

    @staticmethod
    def engine_fault(e: BaseException) -> bool:
        """True for failures that say something about the ENGINE's state (a HIP runtime error, a call-order error).  A malformed
        frame (TypeError / ValueError, RTD_E_INVALID), an allocation failure (torch.cuda.OutOfMemoryError: a fresh engine would fail
        the same way, with the old arenas still resident) and a detector that miscounts its results are the input's or the caller's
        business: they are answered with [] and never cost a rebuild."""
        try:
            from ._capi import RTD_E_HIP, RTD_E_STATE, RtdError
        except Exception:                                    # a foreign detector without the library: any RuntimeError counts
            return isinstance(e, RuntimeError)
        if isinstance(e, RtdError):
            return e.code in (RTD_E_HIP, RTD_E_STATE)
        return False

    def failed(self, e: Optional[BaseException] = None) -> None:
        """Called by whoever holds the lane (former or finisher) when its batch raised.  After REBUILD_AFTER engine failures in a row
        the old engine is closed FIRST (its weights, arenas and graphs are gone before the new ones are allocated) and the detector is
        loaded afresh - a new handle, new streams, new graphs.  The rebuild runs on the calling pipeline thread: its duration is
        measured and logged, and it happens at most MAX_REBUILDS times per lane."""
        self._held = None
        if e is not None and not self.engine_fault(e):
            return
        self.fail_streak += 1
        if self.fail_streak < self.REBUILD_AFTER or not hasattr(self.detector, "load_model"):
            return
        self.fail_streak = 0
        if self.rebuilds >= self.MAX_REBUILDS:
            logger.error(f"batch coordinator: detector still failing after {self.rebuilds} rebuilds; keeping it as it is")
            return
        t0 = time.monotonic()
        model = getattr(self.detector, "model", None)
        old = getattr(model, "engine", None)
        if old is not None:
            try:
                self.detector.model = None                   # detect() / detect_batch() answer [] while there is no engine
                old.close()
            except Exception as ce:
                logger.error(f"batch coordinator: closing the failed engine raised: {ce}")
        try:
            ok = self.detector.load_model(max_retries=1)
        except Exception as le:                              # load_model of the drop-in never raises; a foreign detector might
            logger.error(f"batch coordinator: rebuilding a detector raised: {le}")
            ok = False
        self.rebuilds += 1
        dt = time.monotonic() - t0
        self.rebuild_seconds += dt
        logger.warning(f"batch coordinator: detector rebuilt after {self.REBUILD_AFTER} engine failures in a row "
                       f"({'ok' if ok else 'FAILED'}, {dt:.2f} s, rebuild {self.rebuilds} of at most {self.MAX_REBUILDS})")


class BatchCoordinator:
    """Cross-camera batcher as a depth-N pipeline: N lanes (detectors), a former thread that cuts batches out of the inbox and starts
    them on the next idle lane, a finisher thread that completes batches in submission order and answers the callbacks.  N = 1 with a
    plain detector is the degenerate case - the reference's SharedInferenceCoordinator behaviour - on the same code path.

    Surface the unchanged callers rely on (src/inference_engine_yolox.py:341-350, main.py:770-838): the constructor keywords,
    `start/stop`, `infer_async(frame, callback, camera_id)`, `get_stats()` keys, `max_batch_size`, `max_batch_wait_ms` (seconds),
    `dropped_frames`; every callback is answered exactly once - with [] when its frame was evicted, its batch raised, or the
    coordinator stopped before it ran."""

    def __init__(self, detector: Any, max_batch_size: int = 4, max_batch_wait_ms: float = 10.0,
                 enable_metrics: bool = True, max_queue_depth: int = 60, extra_detectors: Optional[List[Any]] = None):
        self.detector = detector
        self.detectors: List[Any] = [detector] + list(extra_detectors or [])
        self.max_batch_size = int(max_batch_size)
        self.max_batch_wait_ms = max_batch_wait_ms / 1000.0          # seconds (the reference keeps the converted value under this name)
        self.enable_metrics = enable_metrics
        self.max_queue_depth = int(max_queue_depth)
        self.running = False
        self.dropped_frames = 0
        self._inbox = _Inbox(self.max_queue_depth)
        self._lanes = [_Lane(d) for d in self.detectors]
        self._flying: deque = deque()                                # (lane, asks, t_begin) in submission order
        self._flying_cv = threading.Condition()
        self._threads: List[threading.Thread] = []
        # failures: counted, and the FIRST one kept verbatim (the reference answers [] and logs; a silent "no detections" must be visible)
        self.failed_batches = 0
        self.failed_frames = 0
        self.first_error: Optional[str] = None
        # meter
        self.total_batches = 0
        self.total_frames = 0
        self.total_batch_time_ms = 0.0
        self.batch_sizes: deque = deque(maxlen=1000)
        self.wait_times_ms: deque = deque(maxlen=1000)

    @property
    def pending_queue(self):                                         # len() of it is read by stats pages
        return self._inbox._q

    # ---- lifecycle ------------------------------------------------------------------------------------
    def start(self):
        if self.running:
            logger.warning("batch coordinator: start() on a running instance ignored")
            return
        self.running = True
        self._inbox.reopen()
        self._threads = [threading.Thread(target=self._form, name="BatchFormer", daemon=True),
                         threading.Thread(target=self._finish, name="BatchFinisher", daemon=True)]
        for t in self._threads:
            t.start()

    def stop(self):
        if not self.running:
            return
        self.running = False
        leftovers = self._inbox.close()                              # nobody will run these: answer them now
        with self._flying_cv:
            self._flying_cv.notify_all()
        for t in self._threads:
            t.join(timeout=5.0)
            if t.is_alive():
                logger.warning(f"batch coordinator: thread {t.name} still busy after stop()")
        self._threads = []
        for ask in leftovers:
            ask.reply([])

    def __enter__(self):
        self.start()
        return self

    def __exit__(self, *exc):
        self.stop()

    # ---- camera side ----------------------------------------------------------------------------------
    def infer_async(self, frame: Any, callback: Callable[[List[Dict[str, Any]]], None], camera_id: Optional[str] = None):
        if not self.running:
            raise RuntimeError("batch coordinator is stopped: start() it before infer_async()")
        evicted = self._inbox.put(_Ask(frame, callback, camera_id))
        if evicted is not None:
            self.dropped_frames += 1
            if self.dropped_frames % 10 == 0:
                logger.warning(f"batch coordinator overloaded: {self.dropped_frames} frames evicted so far")
            evicted.reply([])

    # ---- pipeline -------------------------------------------------------------------------------------
    def _form(self):
        turn = 0
        while self.running:
            lane = self._lanes[turn]
            # the lane FIRST, then the frames: while every lane is busy the frames stay in the inbox, where the drop-oldest rule
            # applies to them - the reference's loop cuts its next batch only after the previous one has been answered
            # (src/shared_inference_coordinator.py:176-190; traces in tests/golden/host_coordinator.json)
            if not lane.idle.acquire(timeout=0.1):
                continue
            asks = self._inbox.take(self.max_batch_size, self.max_batch_wait_ms)
            if not asks:
                lane.idle.release()
                continue
            turn = (turn + 1) % len(self._lanes)
            t_begin = time.monotonic()
            if self.enable_metrics:
                self.wait_times_ms.extend((t_begin - a.t_in) * 1000.0 for a in asks)
            try:
                lane.begin([a.frame for a in asks])
            except Exception as e:
                self._note_failure("begin", e, len(asks))
                lane.failed(e)
                lane.idle.release()
                logger.error(f"batch coordinator: could not start a batch of {len(asks)}: {e}", exc_info=True)
                for a in asks:
                    a.reply([])
                continue
            with self._flying_cv:
                self._flying.append((lane, asks, t_begin))
                self._flying_cv.notify()

    def _finish(self):
        while True:
            with self._flying_cv:
                while not self._flying and self.running:
                    self._flying_cv.wait(0.1)
                if not self._flying:
                    if self._threads and self._threads[0].is_alive():   # stopping: the former may still be handing over its last batch
                        self._flying_cv.wait(0.02)
                        continue
                    return
                lane, asks, t_begin = self._flying.popleft()
            failed = False
            try:
                results = list(lane.finish())
            except Exception as e:
                self._note_failure("finish", e, len(asks))
                lane.failed(e)
                logger.error(f"batch coordinator: batch of {len(asks)} failed: {e}", exc_info=True)
                results = [[] for _ in asks]
                failed = True
            else:
                lane.succeeded()
                if len(results) != len(asks):
                    # the reference pairs requests and results with zip (src/shared_inference_coordinator.py:253): the leading
                    # requests get their lists; here the ones left over are answered too (with []), and the miscount is on record
                    self._note_failure("finish", RuntimeError(f"detector returned {len(results)} results for {len(asks)} frames"), 0)
                    logger.error(f"batch coordinator: detector returned {len(results)} results for {len(asks)} frames")
                    results = results[:len(asks)] + [[] for _ in range(len(asks) - len(results))]
            # every callback of the batch has fired before the lane is handed back: the next batch reaches the detector afterwards
            for a, dets in zip(asks, results):
                a.reply(dets)
            if self.enable_metrics and not failed:
                self.total_batches += 1
                self.total_frames += len(asks)
                self.total_batch_time_ms += (time.monotonic() - t_begin) * 1000.0
                self.batch_sizes.append(len(asks))
            lane.idle.release()

    def _note_failure(self, where: str, e: BaseException, n_frames: int) -> None:
        self.failed_batches += 1
        self.failed_frames += n_frames
        if self.first_error is None:
            self.first_error = f"{where}: {type(e).__name__}: {e}"

    def failure_stats(self) -> Dict[str, Any]:
        """Build-specific (the reference's coordinator only logs): how many batches were answered with [] and why the first one was."""
        return {"failed_batches": self.failed_batches, "failed_frames": self.failed_frames, "first_error": self.first_error,
                "detector_rebuilds": sum(l.rebuilds for l in self._lanes),
                "detector_rebuild_seconds": round(sum(l.rebuild_seconds for l in self._lanes), 3)}

    def get_stats(self) -> Dict[str, Any]:
        if not self.enable_metrics or self.total_batches == 0:
            return {"enabled": False, "total_batches": 0, "total_frames": 0, **self.failure_stats()}
        busy_s = self.total_batch_time_ms / 1000.0
        return {
            **self.failure_stats(),
            "enabled": True,
            "total_batches": self.total_batches,
            "total_frames": self.total_frames,
            "avg_batch_size": round(sum(self.batch_sizes) / len(self.batch_sizes), 2),
            "avg_batch_time_ms": round(self.total_batch_time_ms / self.total_batches, 2),
            "avg_wait_time_ms": round(sum(self.wait_times_ms) / len(self.wait_times_ms), 2) if self.wait_times_ms else 0,
            "throughput_fps": round(self.total_frames / busy_s, 1) if busy_s > 0 else 0,
            "queue_depth": len(self._inbox),
        }


def _accepts(cls, keyword: str) -> bool:
    import inspect
    try:
        params = inspect.signature(cls.__init__).parameters
    except (TypeError, ValueError):
        return False
    return keyword in params or any(p.kind == p.VAR_KEYWORD for p in params.values())


def make_rtdetr_coordinator(config: Dict[str, Any], coordinator_cls=None, detector_cls=None):
    """What `_initialize_shared_coordinator` (main.py:770-838) would do for `detector_type: rtdetr`.

    Returns a started-able coordinator, or None when batching is disabled / the detector is not RT-DETR /
    the model fails to load (same "return None" behaviour as the reference)."""
    detection = config.get("detection", {})
    batching = detection.get("batching", {})
    if not batching.get("enabled", False):
        return None
    if detection.get("detector_type", "yolox").lower() != "rtdetr":
        return None
    if detector_cls is None:
        from .rtdetr_detector import RTDETRDetector as detector_cls
    if coordinator_cls is None:
        try:
            from src.shared_inference_coordinator import SharedInferenceCoordinator as coordinator_cls  # the reference's own class
        except Exception:
            coordinator_cls = BatchCoordinator
    rt = detection.get("rtdetr", {})
    max_batch = int(batching.get("max_batch_size", 4))
    depth = int(batching.get("pipeline_depth", 1))      # build-specific key: batches in flight (2 = +33 %, 3 = +46 % throughput on one MI355X)
    # several detectors share the GPU: their kernels lean towards throughput (rtd_config.profile); the reference's own
    # detector class knows no such argument and is never built with depth > 1
    prof = {"profile": "throughput"} if depth > 1 else {}
    prof["prepare_batches"] = tuple(range(1, max_batch + 1))   # every batch size the former can cut: planned, warmed and graphed inside load_model
    if detector_cls is not None and not _accepts(detector_cls, "prepare_batches"):
        prof.pop("prepare_batches")
    if "precision" in rt:                               # build-specific key detection.rtdetr.precision: f16x3 (default) | bf16 | fp32
        prof["precision"] = rt["precision"]
    try:
        detector = detector_cls(
            config_path=rt.get("config_path", "RT-DETR/rtdetrv2_pytorch/configs/rtdetrv2/rtdetrv2_r18vd_120e_coco.yml"),
            model_path=rt.get("weights", "models/rtdetr/rtdetrv2_r18vd.pth"),
            device=detection.get("device", "cuda:0"),
            conf_threshold=detection.get("conf_threshold", 0.25),
            input_size=tuple(detection.get("input_size", [640, 640])),
            wildlife_only=detection.get("wildlife_only", True),
            max_batch=max_batch,
            **prof,
        )
        if not detector.load_model():
            logger.error("Failed to load RT-DETR detector for coordinator")
            return None
        if depth > 1:
            extra = []
            for _ in range(depth - 1):
                d2 = detector_cls(config_path=detector.config_path, model_path=detector.model_path, device=detector.device,
                                  conf_threshold=detector.conf_threshold, input_size=detector.input_size,
                                  wildlife_only=detector.wildlife_only, max_batch=max_batch, **prof)
                if not d2.load_model():
                    logger.error("Failed to load a pipeline detector; falling back to one batch in flight")
                    extra = []
                    break
                extra.append(d2)
            if extra:       # the pipeline is a feature of this build's coordinator, whatever class was asked for
                return BatchCoordinator(detector=detector, max_batch_size=max_batch,
                                        max_batch_wait_ms=batching.get("max_batch_wait_ms", 10.0),
                                        enable_metrics=batching.get("enable_metrics", True), extra_detectors=extra)
        return coordinator_cls(detector=detector, max_batch_size=max_batch,
                               max_batch_wait_ms=batching.get("max_batch_wait_ms", 10.0),
                               enable_metrics=batching.get("enable_metrics", True))
    except Exception as e:
        logger.error(f"Failed to initialize shared coordinator: {e}")
        return None


def install(system_cls) -> None:
    """Let `TelescopeDetectionSystem` batch RT-DETR too, without editing main.py:

        import main, telescope_cam_detection_amd.batching as b
        b.install(main.TelescopeDetectionSystem)
    """
    original = system_cls._initialize_shared_coordinator

    def patched(self):
        coord = make_rtdetr_coordinator(self.config)
        return coord if coord is not None else original(self)

    system_cls._initialize_shared_coordinator = patched
