"""Scripted scenarios for a cross-camera batch coordinator (test infrastructure).

One driver, two users: `oracle/make_host_golden.py` runs it - in the build container only - on the REFERENCE's
`SharedInferenceCoordinator` (/root/reference/src/shared_inference_coordinator.py:27-338, a stdlib-only module) and writes
the traces to `tests/golden/host_coordinator.json`; `tests/test_host_golden.py` runs it on this build's `BatchCoordinator`
(one lane) and requires the same traces.  A trace is the ordered list of what the outside world can observe: the batches the
detector was handed (`["batch", k, [frame ids]]`) and the callbacks that fired (`["cb", frame id, detections]`).

Determinism: frames are integers, the scripted detector blocks on gates the scenario opens, and every step waits for the event
that makes the next one unambiguous; the only clocks involved are batch-wait windows of >= 150 ms against submission bursts
of microseconds.
"""
from __future__ import annotations

import threading
import time
from typing import Any, Dict, List


class ScriptedDetector:
    """`detect_batch(frames)` (the one method the coordinator calls, :250) over integer frames: logs the batch, optionally blocks on
    a gate, raises, or returns too few lists.  The answer for frame f is `[{"frame": f, "batch": k}]`."""

    def __init__(self, log: list, gated=(), raise_on=(), short_on=()):
        self.log = log
        self.calls = 0
        self.gates = {k: threading.Event() for k in gated}
        self.entered: Dict[int, threading.Event] = {}
        self.raise_on = set(raise_on)
        self.short_on = set(short_on)
        self._lock = threading.Lock()

    def entered_event(self, k: int) -> threading.Event:
        with self._lock:
            return self.entered.setdefault(k, threading.Event())

    def detect_batch(self, frames):
        k = self.calls
        self.calls += 1
        self.log.append(["batch", k, [int(f) for f in frames]])
        self.entered_event(k).set()
        if k in self.gates:
            assert self.gates[k].wait(timeout=20.0), f"scenario never opened gate {k}"
        if k in self.raise_on:
            raise RuntimeError(f"scripted failure of batch {k}")
        out = [[{"frame": int(f), "batch": k}] for f in frames]
        return out[:-1] if k in self.short_on else out


class _Run:
    def __init__(self, coordinator_cls, detector_kw=None, **coord_kw):
        self.log: list = []
        self.det = ScriptedDetector(self.log, **(detector_kw or {}))
        self.coord = coordinator_cls(detector=self.det, **coord_kw)
        self.answered = 0
        self._cv = threading.Condition()
        self.raising_callbacks = set()

    def cb(self, i: int):
        def answer(dets):
            self.log.append(["cb", i, dets])
            with self._cv:
                self.answered += 1
                self._cv.notify_all()
            if i in self.raising_callbacks:
                raise ValueError(f"scripted callback failure for frame {i}")
        return answer

    def submit(self, ids):
        for i in ids:
            self.coord.infer_async(i, self.cb(i), camera_id=f"cam{i % 4}")

    def wait_answers(self, n: int, timeout: float = 20.0):
        with self._cv:
            ok = self._cv.wait_for(lambda: self.answered >= n, timeout)
        assert ok, f"only {self.answered} of {n} callbacks fired"

    def wait_batch(self, k: int, timeout: float = 20.0):
        assert self.det.entered_event(k).wait(timeout), f"batch {k} never reached the detector"

    def stats(self, keys=("enabled", "total_batches", "total_frames", "avg_batch_size", "queue_depth")):
        s = self.coord.get_stats()
        return {"keys": sorted(s.keys()), "values": {k: s[k] for k in keys if k in s}}


def burst(coordinator_cls) -> Dict[str, Any]:
    """Ten frames arrive while batch 0 is on the GPU: batches of max_batch_size are cut in arrival order, the last one after the
    wait window (:205-223)."""
    r = _Run(coordinator_cls, {"gated": (0,)}, max_batch_size=4, max_batch_wait_ms=300.0)
    before = r.stats()
    r.coord.start()
    r.submit(range(10))
    r.wait_batch(0)
    r.det.gates[0].set()
    r.wait_answers(10)
    time.sleep(0.05)
    after = r.stats()
    r.coord.stop()
    return {"trace": r.log, "stats_before": before, "stats_after": after, "dropped_frames": r.coord.dropped_frames}


def drop_oldest(coordinator_cls, depth: int) -> Dict[str, Any]:
    """The queue holds `depth` frames while the detector is busy; each further frame evicts the OLDEST one, whose callback gets []
    on the submitting thread (:149-164)."""
    r = _Run(coordinator_cls, {"gated": (0,)}, max_batch_size=4, max_batch_wait_ms=150.0, max_queue_depth=depth)
    r.coord.start()
    r.submit([0])
    r.wait_batch(0)                       # batch 0 = [0] is inside the detector; nothing else leaves the queue until the gate opens
    r.submit(range(1, depth + 4))         # depth frames fill the queue, three more evict frames 1, 2, 3
    dropped = r.coord.dropped_frames
    depth_seen = len(r.coord.pending_queue)
    r.det.gates[0].set()
    r.wait_answers(depth + 4)
    time.sleep(0.05)
    after = r.stats()
    r.coord.stop()
    return {"trace": r.log, "dropped_frames": dropped, "queue_depth_while_busy": depth_seen, "stats_after": after}


def raising_batch(coordinator_cls) -> Dict[str, Any]:
    """A batch whose detect_batch raises: every request of THAT batch is answered with [], the coordinator goes on (:280-288);
    only successful batches are counted (:259-263)."""
    r = _Run(coordinator_cls, {"gated": (0,), "raise_on": (1,)}, max_batch_size=2, max_batch_wait_ms=200.0)
    r.coord.start()
    r.submit(range(4))
    r.wait_batch(0)
    r.det.gates[0].set()
    r.wait_answers(4)
    r.submit([4])
    r.wait_answers(5)
    time.sleep(0.05)
    after = r.stats()
    r.coord.stop()
    return {"trace": r.log, "stats_after": after}


def raising_callback(coordinator_cls) -> Dict[str, Any]:
    """A camera's callback that raises does not cost the other requests of the batch their answers (:253-257)."""
    r = _Run(coordinator_cls, {"gated": (0,)}, max_batch_size=4, max_batch_wait_ms=200.0)
    r.raising_callbacks.add(1)
    r.coord.start()
    r.submit([0])
    r.wait_batch(0)
    r.submit(range(1, 5))
    r.det.gates[0].set()
    r.wait_answers(5)
    time.sleep(0.05)
    after = r.stats()
    r.coord.stop()
    return {"trace": r.log, "stats_after": after}


def short_answer(coordinator_cls) -> Dict[str, Any]:
    """A detector that returns one list too few: the reference pairs requests and results with zip (:253), so the leading requests
    get their detections and the last one is never answered by it."""
    r = _Run(coordinator_cls, {"gated": (0,), "short_on": (1,)}, max_batch_size=3, max_batch_wait_ms=200.0)
    r.coord.start()
    r.submit([0])
    r.wait_batch(0)
    r.submit(range(1, 4))
    r.det.gates[0].set()
    r.wait_answers(3)
    time.sleep(0.3)                       # anything else the coordinator has to say about batch 1 has been said by now
    r.coord.stop()
    return {"trace": r.log}


def lifecycle(coordinator_cls) -> Dict[str, Any]:
    """infer_async on a stopped coordinator raises RuntimeError (:130-131); start() twice is harmless (:88-90); with metrics off
    get_stats() stays the three-key form (:311-316)."""
    out: Dict[str, Any] = {}
    r = _Run(coordinator_cls, None, max_batch_size=2, max_batch_wait_ms=150.0, enable_metrics=False)
    try:
        r.submit([0])
        out["before_start"] = "accepted"
    except Exception as e:
        out["before_start"] = type(e).__name__
    r.coord.start()
    r.coord.start()
    r.submit([1, 2])
    r.wait_answers(2)
    time.sleep(0.05)
    out["stats_metrics_off"] = r.stats()
    out["max_batch_wait_seconds"] = r.coord.max_batch_wait_ms      # the reference keeps the converted value under this name (:58)
    r.coord.stop()
    r.coord.stop()
    try:
        r.submit([3])
        out["after_stop"] = "accepted"
    except Exception as e:
        out["after_stop"] = type(e).__name__
    out["trace"] = r.log
    return out


def run_all(coordinator_cls) -> Dict[str, Any]:
    return {
        "burst": burst(coordinator_cls),
        "drop_oldest_depth6": drop_oldest(coordinator_cls, 6),
        "drop_oldest_depth60": drop_oldest(coordinator_cls, 60),
        "raising_batch": raising_batch(coordinator_cls),
        "raising_callback": raising_callback(coordinator_cls),
        "short_answer": short_answer(coordinator_cls),
        "lifecycle": lifecycle(coordinator_cls),
    }
