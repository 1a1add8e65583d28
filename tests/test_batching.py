"""Batching coordinator for RT-DETR: CPU tests with a fake detector mirror the behaviours
of the reference's src/shared_inference_coordinator.py; the GPU test drives the real detector from 4 camera threads."""
import threading
import time

import numpy as np
import pytest

from telescope_cam_detection_amd.batching import BatchCoordinator, install, make_rtdetr_coordinator


class FakeDetector:
    def __init__(self, delay=0.0, fail_on=None):
        self.batches = []
        self.delay = delay
        self.fail_on = fail_on

    def detect_batch(self, frames):
        self.batches.append(len(frames))
        if self.delay:
            time.sleep(self.delay)
        if self.fail_on is not None and any(int(f[0, 0, 0]) == self.fail_on for f in frames):
            raise RuntimeError("boom")
        return [[{"class_id": int(f[0, 0, 0]), "class_name": "x", "confidence": 1.0, "bbox": {}}] for f in frames]


def frame(tag):
    return np.full((4, 4, 3), tag, np.uint8)


def test_requests_are_batched_and_routed_to_their_callbacks():
    det = FakeDetector(delay=0.02)
    got = {}
    done = threading.Event()
    with BatchCoordinator(det, max_batch_size=4, max_batch_wait_ms=30.0) as c:
        def cb(tag):
            def f(d):
                got[tag] = d
                if len(got) == 10:
                    done.set()
            return f
        for i in range(10):
            c.infer_async(frame(i), cb(i), camera_id=f"cam{i % 4}")
        assert done.wait(5.0)
        stats = c.get_stats()
    assert all(got[i][0]["class_id"] == i for i in range(10))           # every frame answered by ITS result
    assert max(det.batches) <= 4 and sum(det.batches) == 10 and len(det.batches) < 10   # real batching happened
    assert stats["enabled"] and stats["total_frames"] == 10 and stats["total_batches"] == len(det.batches)
    # the reference's keys (src/shared_inference_coordinator.py:318-338) plus this build's failure account
    assert set(stats) == {"enabled", "total_batches", "total_frames", "avg_batch_size", "avg_batch_time_ms",
                          "avg_wait_time_ms", "throughput_fps", "queue_depth",
                          "failed_batches", "failed_frames", "first_error", "detector_rebuilds", "detector_rebuild_seconds"}
    assert stats["failed_batches"] == 0 and stats["first_error"] is None


def test_partial_batch_flushes_after_max_wait():
    det = FakeDetector()
    ev = threading.Event()
    with BatchCoordinator(det, max_batch_size=8, max_batch_wait_ms=20.0) as c:
        t0 = time.time()
        c.infer_async(frame(7), lambda d: ev.set())
        assert ev.wait(2.0)
        assert time.time() - t0 < 1.0
    assert det.batches == [1]


def test_queue_overflow_drops_oldest_with_empty_result():
    det = FakeDetector(delay=0.2)
    res = {}
    c = BatchCoordinator(det, max_batch_size=1, max_batch_wait_ms=1.0, max_queue_depth=2)
    c.start()
    c.infer_async(frame(0), lambda d: res.__setitem__(0, d))            # taken by the worker immediately
    time.sleep(0.05)
    for i in (1, 2, 3):                                                  # queue depth 2: request 1 must be dropped
        c.infer_async(frame(i), lambda d, i=i: res.__setitem__(i, d))
    deadline = time.time() + 5
    while len(res) < 4 and time.time() < deadline:
        time.sleep(0.01)
    c.stop()
    assert res[1] == [] and c.dropped_frames == 1
    assert res[2][0]["class_id"] == 2 and res[3][0]["class_id"] == 3


def test_detector_exception_gives_every_callback_an_empty_list_and_keeps_running():
    det = FakeDetector(fail_on=5)
    res = {}
    with BatchCoordinator(det, max_batch_size=2, max_batch_wait_ms=50.0) as c:
        c.infer_async(frame(5), lambda d: res.__setitem__("a", d))
        c.infer_async(frame(6), lambda d: res.__setitem__("b", d))
        time.sleep(0.3)
        c.infer_async(frame(9), lambda d: res.__setitem__("c", d))
        time.sleep(0.3)
    assert res["a"] == [] and res["b"] == [] and res["c"][0]["class_id"] == 9
    st = c.get_stats()                                                    # the failure is counted and its first cause kept verbatim
    assert st["failed_batches"] == 1 and st["failed_frames"] == 2 and "boom" in st["first_error"] and st["detector_rebuilds"] == 0
    with pytest.raises(RuntimeError):
        c.infer_async(frame(1), lambda d: None)                          # stopped


def test_stop_answers_every_request_that_never_ran():
    """requests still queued when the coordinator stops must not leave their camera waiting - each gets []"""
    det = FakeDetector(delay=0.15)
    res = {}
    c = BatchCoordinator(det, max_batch_size=1, max_batch_wait_ms=1.0, max_queue_depth=50)
    c.start()
    for i in range(6):
        c.infer_async(frame(i), lambda d, i=i: res.__setitem__(i, d))
    time.sleep(0.05)                                                     # batch 0 is running, 1 may be formed, the rest wait in the inbox
    c.stop()
    assert set(res) == set(range(6))                                     # everyone answered exactly once
    assert res[0][0]["class_id"] == 0
    assert sum(1 for v in res.values() if v == []) >= 3                  # the queued ones got []
    assert all(v == [] or v[0]["class_id"] == k for k, v in res.items())


def test_make_coordinator_follows_reference_config_keys():
    built = {}

    class Det:
        def __init__(self, **kw):
            built.update(kw)

        def load_model(self):
            return True

        def detect_batch(self, frames):
            return [[] for _ in frames]

    cfg = {"detection": {"detector_type": "rtdetr", "device": "cuda:0", "input_size": [640, 640], "conf_threshold": 0.3,
                         "wildlife_only": False, "rtdetr": {"config_path": "x_r50vd.yml", "weights": "w.pth"},
                         "batching": {"enabled": True, "max_batch_size": 6, "max_batch_wait_ms": 5.0}}}
    c = make_rtdetr_coordinator(cfg, coordinator_cls=BatchCoordinator, detector_cls=Det)
    assert isinstance(c, BatchCoordinator) and c.max_batch_size == 6 and abs(c.max_batch_wait_ms - 0.005) < 1e-9
    assert built["config_path"] == "x_r50vd.yml" and built["model_path"] == "w.pth" and built["max_batch"] == 6
    assert built["input_size"] == (640, 640) and built["conf_threshold"] == 0.3 and built["wildlife_only"] is False
    assert "profile" not in built and "precision" not in built         # one batch in flight: no kernel-profile / precision argument
    assert built["prepare_batches"] == (1, 2, 3, 4, 5, 6)              # every size the former can cut is planned + graphed at load time

    class RefSignature:                                                 # the reference class's constructor takes none of the build's keywords
        def __init__(self, config_path, model_path, device, conf_threshold, input_size, wildlife_only, max_batch=8):
            built["ref_ok"] = True

        def load_model(self):
            return True

        def detect_batch(self, frames):
            return [[] for _ in frames]

    assert isinstance(make_rtdetr_coordinator(cfg, coordinator_cls=BatchCoordinator, detector_cls=RefSignature), BatchCoordinator) and built["ref_ok"]

    # pipeline_depth > 1: every detector of the pipeline is built with the throughput kernel profile
    profiles = []

    class PipeDet(Det):
        def __init__(self, **kw):
            profiles.append(kw.get("profile"))
            self.config_path, self.model_path, self.device = kw["config_path"], kw["model_path"], kw["device"]
            self.conf_threshold, self.input_size, self.wildlife_only = kw["conf_threshold"], kw["input_size"], kw["wildlife_only"]

        def detect_batch_async(self, frames):
            return frames

        def detect_batch_collect(self, ticket):
            return [[] for _ in ticket]

    cfg["detection"]["batching"]["pipeline_depth"] = 3
    c3 = make_rtdetr_coordinator(cfg, detector_cls=PipeDet)
    assert isinstance(c3, BatchCoordinator) and len(c3.detectors) == 3 and profiles == ["throughput"] * 3
    del cfg["detection"]["batching"]["pipeline_depth"]
    cfg["detection"]["batching"]["enabled"] = False
    assert make_rtdetr_coordinator(cfg, detector_cls=Det) is None
    cfg["detection"]["batching"]["enabled"] = True
    cfg["detection"]["detector_type"] = "yolox"
    assert make_rtdetr_coordinator(cfg, detector_cls=Det) is None

    class System:                                                       # stands in for main.TelescopeDetectionSystem
        def __init__(self, config):
            self.config = config

        def _initialize_shared_coordinator(self):
            return "reference-path"

    install(System)
    assert System(cfg)._initialize_shared_coordinator() == "reference-path"     # yolox: reference behaviour untouched


class Wedged(FakeDetector):
    """fails with `error()` until load_model has run `heal_after` times"""

    def __init__(self, error, heal_after=1):
        super().__init__()
        self.loads = 0
        self.error = error
        self.heal_after = heal_after
        self.model = None

    def load_model(self, max_retries=3):
        self.loads += 1
        return True

    def detect_batch(self, frames):
        if self.loads < self.heal_after:
            raise self.error()
        return super().detect_batch(frames)


def _one_by_one(c, n, res):
    for i in range(n):
        ev = threading.Event()
        c.infer_async(frame(i), lambda d, i=i, ev=ev: (res.__setitem__(i, d), ev.set()))
        assert ev.wait(2.0)


def test_a_lane_rebuilds_its_detector_after_three_engine_failures_in_a_row():
    from telescope_cam_detection_amd._capi import RTD_E_HIP, RtdError
    det = Wedged(lambda: RtdError(RTD_E_HIP, "stream stuck"))
    res = {}
    with BatchCoordinator(det, max_batch_size=1, max_batch_wait_ms=1.0) as c:
        _one_by_one(c, 5, res)
        st = c.get_stats()
    assert [res[i] for i in range(3)] == [[], [], []] and res[3][0]["class_id"] == 3 and res[4][0]["class_id"] == 4
    assert det.loads == 1 and st["detector_rebuilds"] == 1 and st["failed_batches"] == 3 and "stream stuck" in st["first_error"]
    assert st["detector_rebuild_seconds"] >= 0.0


def test_malformed_frames_and_allocation_failures_never_cost_a_rebuild():
    """a camera that keeps delivering bad frames (TypeError / ValueError from the detector's frame checks,
    RTD_E_INVALID from the library), an out-of-memory batch and plain RuntimeErrors of a foreign detector are answered with [] and
    counted - the engine, its arenas and graphs stay as they are."""
    import torch
    from telescope_cam_detection_amd._capi import RTD_E_INVALID, RTD_E_OOM, RtdError
    errors = [lambda: TypeError("unsupported frame type <class 'str'>"), lambda: ValueError("frame is not HWC uint8"),
              lambda: RtdError(RTD_E_INVALID, "frame 0: 0 x 0"), lambda: torch.cuda.OutOfMemoryError("HIP out of memory"),
              lambda: RtdError(RTD_E_OOM, "arena"), lambda: RuntimeError("boom")]
    for make in errors:
        det = Wedged(make, heal_after=1)
        res = {}
        with BatchCoordinator(det, max_batch_size=1, max_batch_wait_ms=1.0) as c:
            _one_by_one(c, 7, res)
            st = c.get_stats()
        assert all(res[i] == [] for i in range(7)), make()
        assert det.loads == 0 and st["detector_rebuilds"] == 0 and st["failed_batches"] == 7, (make(), st)


def test_rebuilds_are_capped():
    from telescope_cam_detection_amd._capi import RTD_E_STATE, RtdError
    from telescope_cam_detection_amd.batching import _Lane
    det = Wedged(lambda: RtdError(RTD_E_STATE, "no weights"), heal_after=10 ** 6)
    res = {}
    n = _Lane.REBUILD_AFTER * (_Lane.MAX_REBUILDS + 3)
    with BatchCoordinator(det, max_batch_size=1, max_batch_wait_ms=1.0) as c:
        _one_by_one(c, n, res)
        st = c.get_stats()
    assert all(res[i] == [] for i in range(n))
    assert det.loads == _Lane.MAX_REBUILDS == st["detector_rebuilds"] and st["failed_batches"] == n


def test_the_failed_engine_is_closed_before_the_new_one_is_built():
    from telescope_cam_detection_amd._capi import RTD_E_HIP, RtdError
    order = []

    class Eng:
        def close(self):
            order.append("close")

    class Model:
        engine = Eng()

    class Det(Wedged):
        def load_model(self, max_retries=3):
            order.append(("load", self.model))
            self.model = Model()
            return super().load_model(max_retries)

    det = Det(lambda: RtdError(RTD_E_HIP, "fault"))
    det.model = Model()
    res = {}
    with BatchCoordinator(det, max_batch_size=1, max_batch_wait_ms=1.0) as c:
        _one_by_one(c, 4, res)
    assert order == ["close", ("load", None)] and res[3][0]["class_id"] == 3


class FakeAsyncDetector(FakeDetector):
    """detect_batch_async / detect_batch_collect like RTDETRDetector's pipelined API: the work 'runs' between the two calls"""

    def __init__(self, delay=0.0, fail_on=None):
        super().__init__(delay, fail_on)
        self.outstanding = 0
        self.max_outstanding = 0
        self.lock = threading.Lock()

    def detect_batch_async(self, frames):
        with self.lock:
            self.outstanding += 1
            self.max_outstanding = max(self.max_outstanding, self.outstanding)
        return {"frames": list(frames), "t": time.time()}

    def detect_batch_collect(self, ticket):
        left = self.delay - (time.time() - ticket["t"])
        if left > 0:
            time.sleep(left)
        with self.lock:
            self.outstanding -= 1
        return FakeDetector.detect_batch(FakeDetector(0.0, self.fail_on), ticket["frames"])


def test_pipelined_coordinator_keeps_two_batches_in_flight_and_preserves_routing():
    d0, d1 = FakeAsyncDetector(delay=0.05), FakeAsyncDetector(delay=0.05)
    got = {}
    done = threading.Event()
    n = 32
    t0 = time.time()
    with BatchCoordinator(d0, max_batch_size=4, max_batch_wait_ms=2.0, extra_detectors=[d1]) as c:
        def cb(tag):
            def f(d):
                got[tag] = d
                if len(got) == n:
                    done.set()
            return f
        for i in range(n):
            c.infer_async(frame(i), cb(i), camera_id=f"cam{i % 4}")
        assert done.wait(10.0)
        stats = c.get_stats()
    elapsed = time.time() - t0
    assert all(got[i][0]["class_id"] == i for i in range(n))
    assert stats["total_frames"] == n
    assert d0.max_outstanding == 1 and d1.max_outstanding == 1          # a detector never holds two batches
    # 8 batches of 50 ms: serial = 0.4 s, two in flight ~0.2 s
    assert elapsed < 0.34, elapsed


def test_pipelined_coordinator_answers_every_callback_when_a_batch_fails():
    d0, d1 = FakeAsyncDetector(fail_on=3), FakeAsyncDetector(fail_on=3)
    got = {}
    done = threading.Event()
    with BatchCoordinator(d0, max_batch_size=2, max_batch_wait_ms=1.0, extra_detectors=[d1]) as c:
        def cb(tag):
            def f(d):
                got[tag] = d
                if len(got) == 8:
                    done.set()
            return f
        for i in range(8):
            c.infer_async(frame(i), cb(i))
        assert done.wait(5.0)
    assert got[3] == [] and sum(1 for v in got.values() if v == []) in (1, 2)     # the failing batch (<= 2 frames) got []
    assert all(got[i][0]["class_id"] == i for i in got if got[i])


def _same_detections(got, want, conf_tol=1e-5, box_tol=1e-3):
    """conf_tol == 0: identical lists (same engine, same arithmetic).  Otherwise order-tolerant (near-equal scores may swap places
    between two arithmetics): every wanted row has its own partner of the same class within the tolerances."""
    if conf_tol == 0:
        assert [(x["class_id"], x["confidence"], x["bbox"]) for x in got] == [(x["class_id"], x["confidence"], x["bbox"]) for x in want]
        return
    assert len(got) == len(want), (len(got), len(want))
    used = set()
    for w in want:
        for j, g in enumerate(got):
            if j in used or g["class_id"] != w["class_id"] or abs(g["confidence"] - w["confidence"]) > conf_tol:
                continue
            if all(abs(g["bbox"][k] - w["bbox"][k]) <= box_tol for k in ("x1", "y1", "x2", "y2")):
                used.add(j)
                break
        else:
            raise AssertionError(f"no partner for {w} in {got}")


def _oracle_detections(arch, wseed, frames, input_size, conf, wildlife_only):
    """the CPU oracle's answer in the detector's dict schema (the oracle is the checker here, never the product)"""
    from oracle import rtdetr_oracle as orc
    from tests.util import weights_for
    return orc.detect_batch(arch, weights_for(arch, wseed), frames, input_size, conf_threshold=conf, wildlife_only=wildlife_only)


@pytest.mark.gpu
def test_four_camera_threads_through_the_real_detector():
    from telescope_cam_detection_amd.rtdetr_detector import RTDETRDetector
    from tests.util import load_case
    arch, wseed, input_size, frames, g = load_case("t_tinyb_192x128")
    det = RTDETRDetector(config_path="tinyb", model_path=f"synthetic:tinyb:{wseed}", device="cuda:0", conf_threshold=0.2,
                         input_size=input_size, wildlife_only=False, precision="fp32", max_batch=4, prepare_batches=(1, 2, 3, 4))
    assert det.load_model()
    want = _oracle_detections(arch, wseed, frames, input_size, 0.2, False)          # parity: the coordinator's answers against the ORACLE
    assert sum(len(w) for w in want) > 0
    results = {}
    lock = threading.Lock()

    def camera(cam):
        for it in range(6):
            f = frames[(cam + it) % len(frames)]
            ev = threading.Event()
            def cb(d, key=(cam, it)):
                with lock:
                    results[key] = d
                ev.set()
            coord.infer_async(f, cb, camera_id=f"cam{cam}")
            assert ev.wait(10.0)

    with BatchCoordinator(det, max_batch_size=4, max_batch_wait_ms=5.0) as coord:
        threads = [threading.Thread(target=camera, args=(c,)) for c in range(4)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        stats = coord.get_stats()
    assert stats["first_error"] is None and stats["failed_batches"] == 0, stats
    assert len(results) == 24 and stats["total_frames"] == 24
    for (cam, it), d in results.items():
        _same_detections(d, want[(cam + it) % len(frames)], conf_tol=1e-4, box_tol=2e-2)
        _same_detections(d, det.detect(frames[(cam + it) % len(frames)]), conf_tol=0, box_tol=0)     # and bit-equal to the synchronous call


def _tinyc_pipeline_config(input_size, wseed, depth=2, max_batch=2):
    return {"detection": {"detector_type": "rtdetr", "device": "cuda:0", "conf_threshold": 0.2, "input_size": list(input_size),
                          "wildlife_only": False, "rtdetr": {"config_path": "tinyc", "weights": f"synthetic:tinyc:{wseed}"},
                          "batching": {"enabled": True, "max_batch_size": max_batch, "max_batch_wait_ms": 2.0, "pipeline_depth": depth}}}


@pytest.mark.gpu
def test_pipelined_coordinator_on_two_real_detectors_matches_detect():
    from telescope_cam_detection_amd.rtdetr_detector import RTDETRDetector
    from tests.util import load_case
    arch, wseed, input_size, frames, g = load_case("t_tinyc_160x224")     # default engine (f16x3): widths in whole 32-channel groups
    coord = make_rtdetr_coordinator(_tinyc_pipeline_config(input_size, wseed))
    assert isinstance(coord, BatchCoordinator) and len(coord.detectors) == 2
    for d in coord.detectors:                                             # every batch size was planned and graphed inside load_model
        st = d.model.engine.stats()
        assert st["plans"] == 2 and st["graphs"] == 2 and st["stream_capture_status"] == 0, st
    ref = RTDETRDetector(config_path="tinyc", model_path=f"synthetic:tinyc:{wseed}", device="cuda:0", conf_threshold=0.2,
                         input_size=input_size, wildlife_only=False, max_batch=2)
    assert ref.load_model()
    want = [ref.detect(f) for f in frames]
    oracle_want = _oracle_detections(arch, wseed, frames, input_size, 0.2, False)
    results = {}
    done = threading.Event()
    n = 18
    with coord:
        for i in range(n):
            def cb(d, key=i):
                results[key] = d
                if len(results) == n:
                    done.set()
            coord.infer_async(frames[i % len(frames)], cb, camera_id=f"cam{i % 3}")
        assert done.wait(30.0)
        stats = coord.get_stats()
    assert stats["first_error"] is None and stats["failed_batches"] == 0, stats      # the primary error, not `[] == [...]`, if it ever fails
    for d in coord.detectors:                                             # the serving path only replayed: no plan or graph was built after load
        st = d.model.engine.stats()
        assert st["plans"] == 2 and st["graphs"] == 2 and st["failed_calls"] == 0 and st["submits"] == st["collects"] > 0, st
    for i in range(n):
        _same_detections(results[i], want[i % len(frames)], conf_tol=0, box_tol=0)   # pipelined == synchronous, bit for bit
        _same_detections(results[i], oracle_want[i % len(frames)], conf_tol=1e-3, box_tol=1e-2)


@pytest.mark.gpu
def test_engines_are_created_pipelined_and_destroyed_while_another_coordinator_serves():
    """detectors come and go (load_model builds plans and hipGraphs, a second pipeline runs, engines are closed)
    while a first coordinator keeps serving from its own threads - and torch allocates, copies and records events on its own streams
    in between, as the rest of a real process does.  Every answer of both coordinators must be right and no batch may fail."""
    import torch
    from telescope_cam_detection_amd.rtdetr_detector import RTDETRDetector
    from tests.util import load_case
    arch, wseed, input_size, frames, g = load_case("t_tinyc_160x224")
    ref = RTDETRDetector(config_path="tinyc", model_path=f"synthetic:tinyc:{wseed}", device="cuda:0", conf_threshold=0.2,
                         input_size=input_size, wildlife_only=False, max_batch=2)
    assert ref.load_model()
    want = [ref.detect(f) for f in frames]
    serving = make_rtdetr_coordinator(_tinyc_pipeline_config(input_size, wseed))
    stop = threading.Event()
    bad, served = [], [0]

    def camera(cam):
        k = cam
        while not stop.is_set():
            ev, box = threading.Event(), {}
            serving.infer_async(frames[k % len(frames)], lambda d, ev=ev, box=box: (box.__setitem__("d", d), ev.set()), camera_id=f"cam{cam}")
            if not ev.wait(20.0):
                bad.append(("timeout", cam, k))
                return
            try:
                _same_detections(box["d"], want[k % len(frames)], conf_tol=0, box_tol=0)
            except AssertionError as e:
                bad.append(("mismatch", cam, k, str(e)[:200]))
            served[0] += 1
            k += 3

    with serving:
        cams = [threading.Thread(target=camera, args=(c,)) for c in range(3)]
        for t in cams:
            t.start()
        try:
            for round_ in range(3):
                # torch activity of the host process on ITS streams: pinned uploads, events, a side stream
                side = torch.cuda.Stream()
                with torch.cuda.stream(side):
                    t = torch.from_numpy(frames[0]).pin_memory().to("cuda:0", non_blocking=True).float().mean()
                ev = torch.cuda.Event()
                ev.record(side)
                # a second pipeline is born, serves, and dies
                other = make_rtdetr_coordinator(_tinyc_pipeline_config(input_size, wseed, depth=2, max_batch=2))
                assert other is not None
                got, done = {}, threading.Event()
                with other:
                    for i in range(8):
                        other.infer_async(frames[i % len(frames)], lambda d, i=i: (got.__setitem__(i, d), len(got) == 8 and done.set()), camera_id="x")
                    assert done.wait(20.0)
                    ost = other.get_stats()
                assert ost["first_error"] is None and ost["failed_batches"] == 0, ost
                for i in range(8):
                    _same_detections(got[i], want[i % len(frames)], conf_tol=0, box_tol=0)
                for d in other.detectors:
                    d.model.engine.close()
                ev.synchronize()
                assert ev.query() and torch.isfinite(t).item()
        finally:
            stop.set()
            for t in cams:
                t.join(30.0)
        stats = serving.get_stats()
    assert not bad, bad[:3]
    assert stats["first_error"] is None and stats["failed_batches"] == 0 and served[0] >= 6, (stats, served)
    for d in serving.detectors:
        assert d.model.engine.stats()["stream_capture_status"] == 0
