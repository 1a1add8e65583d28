"""ctypes binding of libmi355rtdetr.so (include/rtdetr_mi355.h).  No compute happens in Python.

The library is required: if it cannot be built / loaded this module raises - there is no CPU or
PyTorch fallback for the hot path.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

from . import build as _build
from .arch import Arch

RTD_OK, RTD_E_INVALID, RTD_E_OOM, RTD_E_HIP, RTD_E_WEIGHTS, RTD_E_STATE = range(6)
PREC_BF16, PREC_FP32, PREC_F16X3 = 0, 1, 2
DT_BF16, DT_F32, DT_F16X2 = 0, 1, 4
SPLIT_GROUP = 32      # channels per [hi | lo] group of a F16X2 tensor (csrc/common.h)


def precision_code(name) -> int:
    """'f16x3' (default engine: hi/lo fp16 pairs, three MFMAs per product - meets the reference tolerance), 'bf16', 'fp32'."""
    if isinstance(name, int):
        return name
    s = str(name).lower()
    if s in ("f16x3", "fp16x3", "pair"):
        return PREC_F16X3
    if s in ("fp32", "f32", "float32"):
        return PREC_FP32
    if s in ("bf16", "bfloat16"):
        return PREC_BF16
    raise ValueError(f"unknown precision {name!r} (f16x3 | bf16 | fp32)")


def to_split(x: np.ndarray) -> np.ndarray:
    """fp32 [..., C] (C % 32 == 0) -> F16X2 storage as uint16 [..., 2C]: per 32-channel group [32 hi | 32 lo],
    hi = fp16_rne(x), lo = fp16_rne(x - hi) (subnormals kept, +-65504 saturation).  Host-side mirror of csrc/common.h
    split_store8 (tests / tools)."""
    x = np.clip(np.ascontiguousarray(x, np.float32), -65504.0, 65504.0)
    C_ = x.shape[-1]
    assert C_ % SPLIT_GROUP == 0
    hi = x.astype(np.float16)
    lo = (x - hi.astype(np.float32)).astype(np.float16)
    g = x.shape[:-1] + (C_ // SPLIT_GROUP, SPLIT_GROUP)
    out = np.stack([hi.view(np.uint16).reshape(g), lo.view(np.uint16).reshape(g)], axis=-2)          # [..., groups, 2, 32]
    return np.ascontiguousarray(out.reshape(x.shape[:-1] + (2 * C_,)))


def from_split(s: np.ndarray) -> np.ndarray:
    """F16X2 storage (uint16 [..., 2C]) -> fp32 [..., C] (hi + lo, exact in fp32)."""
    s = np.ascontiguousarray(s, np.uint16)
    C_ = s.shape[-1] // 2
    g = s.reshape(s.shape[:-1] + (C_ // SPLIT_GROUP, 2, SPLIT_GROUP))
    f = g.view(np.float16).astype(np.float32)
    return np.ascontiguousarray((f[..., 0, :] + f[..., 1, :]).reshape(s.shape[:-1] + (C_,)))


ACT = {"none": 0, "relu": 1, "silu": 2, "gelu": 3}


class RtdConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32), ("device", C.c_int32), ("precision", C.c_int32), ("max_batch", C.c_int32),
        ("input_h", C.c_int32), ("input_w", C.c_int32), ("use_graph", C.c_int32),
        ("layer_type", C.c_int32), ("depths", C.c_int32 * 4), ("hidden_sizes", C.c_int32 * 4),
        ("embedding_size", C.c_int32),
        ("enc_dim", C.c_int32), ("enc_ffn", C.c_int32), ("enc_heads", C.c_int32), ("csp_hidden", C.c_int32),
        ("d_model", C.c_int32), ("dec_ffn", C.c_int32), ("dec_heads", C.c_int32), ("dec_layers", C.c_int32),
        ("num_queries", C.c_int32), ("num_classes", C.c_int32), ("n_levels", C.c_int32), ("n_points", C.c_int32),
        ("offset_scale", C.c_float), ("profile", C.c_int32),
    ]


class RtdDet(C.Structure):
    _fields_ = [("class_id", C.c_int32), ("score", C.c_float), ("x1", C.c_float), ("y1", C.c_float),
                ("x2", C.c_float), ("y2", C.c_float)]


class RtdStats(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("last_error_code", C.c_int32), ("stream_capture_status", C.c_int32), ("in_flight", C.c_int32),
                ("plans", C.c_int64), ("graphs", C.c_int64), ("graph_nodes", C.c_int64), ("graph_launches", C.c_int64),
                ("eager_passes", C.c_int64), ("submits", C.c_int64), ("collects", C.c_int64), ("failed_calls", C.c_int64),
                ("saturated_values", C.c_int64), ("max_abs_filter", C.c_float), ("reserved", C.c_int32)]


class RtdCheckReport(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("rows", C.c_int32), ("rows_matched", C.c_int32), ("worst_score_err", C.c_float),
                ("worst_box_err_px", C.c_float), ("score_tol", C.c_float), ("box_tol_px", C.c_float), ("max_abs_filter", C.c_float),
                ("saturated_values", C.c_int64), ("max_abs_filter_name", C.c_char * 64)]


class RtdLayerTime(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("kernel", C.c_char * 24), ("ms", C.c_float), ("flops", C.c_double),
                ("bytes", C.c_double)]


DET_DTYPE = np.dtype([("class_id", "<i4"), ("score", "<f4"), ("x1", "<f4"), ("y1", "<f4"), ("x2", "<f4"), ("y2", "<f4")])

_lib: Optional[C.CDLL] = None

# every symbol include/rtdetr_mi355.h declares: the product C ABI (what a reference-side binding uses)
EXPORTS = [
    "rtd_version", "rtd_create", "rtd_load_weights", "rtd_infer", "rtd_infer_raw", "rtd_infer_async", "rtd_collect", "rtd_prepare",
    "rtd_result_block", "rtd_sync", "rtd_stream", "rtd_wait_stream", "rtd_signal_stream", "rtd_get_stats", "rtd_arena_bytes", "rtd_destroy",
    "rtd_last_error", "rtd_crop_resize_batch", "rtd_self_check", "rtd_preprocess",
]
# every symbol include/rtdetr_mi355_test.h declares: kernel-level test / bench / debug entry points (csrc/testapi.hip)
TEST_EXPORTS = [
    "rtd_debug_tensor", "rtd_debug_force_topk", "rtd_profile", "rtd_debug_option", "rtd_op_conv", "rtd_op_conv_dual", "rtd_op_conv_next",
    "rtd_op_layernorm", "rtd_op_attention", "rtd_op_msdeform", "rtd_op_topk", "rtd_op_resize", "rtd_bench_conv", "rtd_bench_conv_pair",
    "rtd_bench_mfma_rate",
]


def lib() -> C.CDLL:
    """Load (building first if stale) the HIP library.  Raises if that is impossible."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.LIB_PATH
    override = os.environ.get("RTD_LIB_PATH")   # tools only (tools/ab_lib.sh): A/B an older build of the library on the same box
    if override:
        if not os.path.exists(override):
            raise RuntimeError(f"RTD_LIB_PATH={override} does not exist")
        path = override
    elif _build.needs_build():
        try:
            path = _build.build(verbose=False)
        except Exception as e:  # no hipcc on this box: use the prebuilt library if it is there
            if not os.path.exists(path):
                raise RuntimeError(f"libmi355rtdetr.so is missing and cannot be built: {e}") from e
    # PyTorch-ROCm bundles its own libamdhip64: load it FIRST so that this library's libamdhip64.so.7 dependency resolves
    # to the same runtime instance.  Loaded the other way round (this library, then torch) the process holds two HIP runtimes
    # and the second to touch the GPU reports "no ROCm-capable device" (seen with build() followed by smoke() in one process).
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    L = C.CDLL(path)
    vp, i32, f32, i64 = C.c_void_p, C.c_int32, C.c_float, C.c_int64
    L.rtd_version.restype = C.c_char_p
    L.rtd_last_error.restype = C.c_char_p
    L.rtd_last_error.argtypes = [vp]
    L.rtd_create.argtypes = [C.POINTER(RtdConfig), C.POINTER(vp)]
    L.rtd_load_weights.argtypes = [vp, vp, C.c_size_t]
    L.rtd_infer.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(i32), i32, f32, i32, vp, C.POINTER(i32)]
    L.rtd_infer_raw.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(i32), i32, vp, vp, vp]
    L.rtd_infer_async.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(i32), i32]
    L.rtd_collect.argtypes = [vp, f32, i32, vp, C.POINTER(i32)]
    L.rtd_prepare.argtypes = [vp, i32]
    L.rtd_wait_stream.argtypes = [vp, vp]
    L.rtd_signal_stream.argtypes = [vp, vp]
    L.rtd_get_stats.argtypes = [vp, C.POINTER(RtdStats)]
    if hasattr(L, "rtd_self_check") or not override:     # (an OLDER build under RTD_LIB_PATH, tools/ab_lib.sh, may predate these two)
        L.rtd_self_check.argtypes = [vp, C.c_char_p, C.c_size_t, C.POINTER(RtdCheckReport)]
        L.rtd_self_check.restype = C.c_int
        L.rtd_preprocess.argtypes = [vp, vp, i32, i32, i32, vp]
        L.rtd_preprocess.restype = C.c_int
    L.rtd_result_block.argtypes = [vp, C.POINTER(vp), C.POINTER(i64)]
    L.rtd_sync.argtypes = [vp]
    L.rtd_stream.argtypes = [vp]
    L.rtd_stream.restype = vp
    L.rtd_destroy.argtypes = [vp]
    L.rtd_destroy.restype = None
    L.rtd_debug_tensor.argtypes = [vp, C.c_char_p, vp, i64, C.POINTER(i64)]
    L.rtd_debug_force_topk.argtypes = [vp, vp, i32]
    L.rtd_profile.argtypes = [vp, i32, i32, vp, i32, C.POINTER(i32)]
    L.rtd_arena_bytes.argtypes = [vp]
    L.rtd_arena_bytes.restype = i64
    L.rtd_debug_option.argtypes = [C.c_char_p, i32]
    L.rtd_op_conv.argtypes = [i32, vp, vp, vp, vp, vp] + [i32] * 12
    L.rtd_op_conv_dual.argtypes = [i32, vp, vp, vp, vp, vp, vp] + [i32] * 13
    L.rtd_op_conv_next.argtypes = [i32] + [vp] * 9 + [i32] * 10
    L.rtd_op_layernorm.argtypes = [i32, vp, vp, vp, vp, vp, i32, i32, i32]
    L.rtd_op_attention.argtypes = [i32, vp, vp, vp, i32, i32, i32, i32]
    L.rtd_op_msdeform.argtypes = [i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, C.POINTER(i32), i32, f32]
    L.rtd_op_topk.argtypes = [vp, i32, i32, i32, vp, vp]
    L.rtd_op_resize.argtypes = [vp, i32, i32, vp, i32, i32, i32]
    L.rtd_bench_conv.argtypes = [i32] * 12 + [C.POINTER(f32)]
    L.rtd_bench_conv_pair.argtypes = [C.POINTER(i32), C.POINTER(i32), i32, C.POINTER(f32)]
    if hasattr(L, "rtd_bench_mfma_rate"):      # (absent from older builds loaded through RTD_LIB_PATH)
        L.rtd_bench_mfma_rate.argtypes = [i32, i32, C.POINTER(f32)]
    L.rtd_crop_resize_batch.argtypes = [i32, C.POINTER(vp), C.POINTER(i32), C.POINTER(i32), i32, C.POINTER(f32), C.POINTER(f32), vp, vp]
    _lib = L
    return L


def debug_option(name: str, value: int) -> None:
    rc = lib().rtd_debug_option(name.encode(), int(value))
    if rc != RTD_OK:
        raise ValueError(f"unknown debug option {name!r}")


class RtdError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"rtd error {code}: {msg}")
        self.code = code


def _stats_of(handle) -> dict:
    st = RtdStats()
    if not handle or lib().rtd_get_stats(handle, C.byref(st)) != RTD_OK:
        return {}
    return {k: (float(getattr(st, k)) if k == "max_abs_filter" else int(getattr(st, k))) for k, _ in RtdStats._fields_ if k not in ("struct_size", "reserved")}


def _raise(code: int, handle) -> None:
    msg = (lib().rtd_last_error(handle) or b"").decode(errors="replace")
    st = _stats_of(handle)
    if st:          # a failure describes itself: what the handle had done when it happened, and whether its stream is in capture state
        msg += " | handle: " + ", ".join(f"{k}={v}" for k, v in st.items())
    if code == RTD_E_OOM:
        import torch
        # the only exception the reference's degrade path reacts to (src/inference_engine_yolox.py:607)
        raise torch.cuda.OutOfMemoryError(f"HIP out of memory in libmi355rtdetr: {msg}")
    raise RtdError(code, msg)


PROFILE_LATENCY, PROFILE_THROUGHPUT = 0, 1


def make_config(arch: Arch, device: int, precision: int, max_batch: int, input_size, use_graph: bool,
                profile: int = PROFILE_LATENCY) -> RtdConfig:
    c = RtdConfig()
    c.struct_size = C.sizeof(RtdConfig)
    c.device, c.precision, c.max_batch = device, precision, max_batch
    c.input_h, c.input_w = int(input_size[0]), int(input_size[1])
    c.use_graph = 1 if use_graph else 0
    c.layer_type = 1 if arch.layer_type == "bottleneck" else 0
    c.depths = (C.c_int32 * 4)(*arch.depths)
    c.hidden_sizes = (C.c_int32 * 4)(*arch.hidden_sizes)
    c.embedding_size = arch.embedding_size
    c.enc_dim, c.enc_ffn, c.enc_heads, c.csp_hidden = arch.enc_dim, arch.enc_ffn, arch.enc_heads, arch.csp_hidden
    c.d_model, c.dec_ffn, c.dec_heads, c.dec_layers = arch.d_model, arch.dec_ffn, arch.dec_heads, arch.dec_layers
    c.num_queries, c.num_classes, c.n_levels, c.n_points = arch.num_queries, arch.num_classes, arch.n_levels, arch.n_points
    c.offset_scale = arch.offset_scale
    c.profile = profile
    return c


class Engine:
    """Thin RAII wrapper of one rtd_handle."""

    def __init__(self, arch: Arch, blob: bytes, device: int = 0, precision: int = PREC_BF16, max_batch: int = 8,
                 input_size=(640, 640), use_graph: bool = True, profile: int = PROFILE_LATENCY, prepare=()):
        self.arch = arch
        self.num_queries = arch.num_queries
        self.max_batch = max_batch
        self._h = C.c_void_p()
        cfg = make_config(arch, device, precision, max_batch, input_size, use_graph, profile)
        rc = lib().rtd_create(C.byref(cfg), C.byref(self._h))
        if rc != RTD_OK:
            self._h = C.c_void_p()
            _raise(rc, None)
        buf = (C.c_char * len(blob)).from_buffer_copy(blob)
        rc = lib().rtd_load_weights(self._h, buf, len(blob))
        if rc != RTD_OK:
            try:
                _raise(rc, self._h)
            finally:
                self.close()
        for n in sorted({int(b) for b in prepare}):        # plan + arena + hipGraph of every declared batch size: the serving path only replays
            try:
                self.prepare(n)
            except BaseException:
                self.close()
                raise

    def prepare(self, n: int):
        rc = lib().rtd_prepare(self._h, int(n))
        if rc != RTD_OK:
            _raise(rc, self._h)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().rtd_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- helpers
    @staticmethod
    def _frame_args(frames, on_device: bool):
        n = len(frames)
        ptrs = (C.c_void_p * n)()
        hw = (C.c_int32 * (2 * n))()
        keep = []
        for i, f in enumerate(frames):
            if on_device:
                assert f.dtype.__str__() == "torch.uint8" and f.is_contiguous() and f.dim() == 3 and f.shape[2] == 3
                ptrs[i] = f.data_ptr()
                hw[2 * i], hw[2 * i + 1] = int(f.shape[0]), int(f.shape[1])
            else:
                a = np.ascontiguousarray(f, dtype=np.uint8)
                assert a.ndim == 3 and a.shape[2] == 3, "frames must be HxWx3 uint8 BGR"
                keep.append(a)
                ptrs[i] = a.ctypes.data
                hw[2 * i], hw[2 * i + 1] = a.shape[0], a.shape[1]
        return n, ptrs, hw, keep

    def infer(self, frames, conf: float, wildlife_only: bool, on_device: bool = False):
        n, ptrs, hw, keep = self._frame_args(frames, on_device)
        out = np.zeros((n, self.num_queries), dtype=DET_DTYPE)
        counts = (C.c_int32 * n)()
        rc = lib().rtd_infer(self._h, n, ptrs, hw, int(on_device), float(conf), int(bool(wildlife_only)),
                             out.ctypes.data, counts)
        if rc != RTD_OK:
            _raise(rc, self._h)
        return [out[i, : counts[i]] for i in range(n)]

    def infer_raw(self, frames, on_device: bool = False):
        n, ptrs, hw, keep = self._frame_args(frames, on_device)
        Q = self.num_queries
        labels = np.zeros((n, Q), np.int32)
        boxes = np.zeros((n, Q, 4), np.float32)
        scores = np.zeros((n, Q), np.float32)
        rc = lib().rtd_infer_raw(self._h, n, ptrs, hw, int(on_device), labels.ctypes.data, boxes.ctypes.data, scores.ctypes.data)
        if rc != RTD_OK:
            _raise(rc, self._h)
        return labels, boxes, scores

    def infer_async(self, frames, on_device: bool = True):
        """Enqueue one batch and return.  Host frames are staged inside the library (pinned buffer + one DMA): the arrays may be
        released at once; device frames must stay alive until collect() / sync()."""
        n, ptrs, hw, keep = self._frame_args(frames, on_device)
        rc = lib().rtd_infer_async(self._h, n, ptrs, hw, int(on_device))
        if rc != RTD_OK:
            _raise(rc, self._h)
        return n

    def collect(self, conf: float, wildlife_only: bool):
        """Wait for the batch of the last infer_async and return what infer() returns for it."""
        out = np.zeros((self.max_batch, self.num_queries), dtype=DET_DTYPE)
        counts = (C.c_int32 * self.max_batch)()
        rc = lib().rtd_collect(self._h, float(conf), int(bool(wildlife_only)), out.ctypes.data, counts)
        if rc != RTD_OK:
            _raise(rc, self._h)
        return out, counts

    def make_async_args(self, frames_dev):
        """Pre-marshal (n, ptrs, hw) once so a benchmark loop does no Python work per step."""
        return self._frame_args(frames_dev, True)

    def infer_async_prepared(self, args):
        rc = lib().rtd_infer_async(self._h, args[0], args[1], args[2], 1)
        if rc != RTD_OK:
            _raise(rc, self._h)

    def wait_stream(self, producer_stream: int):
        """The engine's stream waits for everything enqueued so far on `producer_stream` (a raw hipStream_t value, e.g.
        torch.cuda.current_stream().cuda_stream; 0 = the default stream)."""
        rc = lib().rtd_wait_stream(self._h, C.c_void_p(int(producer_stream) or None))
        if rc != RTD_OK:
            _raise(rc, self._h)

    def signal_stream(self, consumer_stream: int):
        """`consumer_stream` waits for everything enqueued so far on the engine's stream."""
        rc = lib().rtd_signal_stream(self._h, C.c_void_p(int(consumer_stream) or None))
        if rc != RTD_OK:
            _raise(rc, self._h)

    def stats(self) -> dict:
        return _stats_of(self._h)

    def preprocess_into(self, frame, on_device: bool, out_ptr: int) -> None:
        """rtd_preprocess: one HWC uint8 BGR frame (numpy array, or a contiguous device tensor when on_device) -> [3, H, W] fp32 at out_ptr (device)."""
        h, w = int(frame.shape[0]), int(frame.shape[1])
        src = frame.data_ptr() if on_device else frame.ctypes.data
        rc = lib().rtd_preprocess(self._h, C.c_void_p(src), h, w, int(on_device), C.c_void_p(out_ptr))
        if rc != RTD_OK:
            _raise(rc, self._h)

    def self_check(self, blob: bytes) -> dict:
        """rtd_self_check: this engine's arithmetic against the library's exact fp32 engine on one built-in frame, with the weights of
        `blob` (the container this engine was loaded from: the handle keeps no host copy)."""
        rep = RtdCheckReport()
        rep.struct_size = C.sizeof(RtdCheckReport)
        rc = lib().rtd_self_check(self._h, blob, len(blob), C.byref(rep))
        if rc != RTD_OK:
            _raise(rc, self._h)
        return {"rows": rep.rows, "rows_matched": rep.rows_matched, "worst_score_err": rep.worst_score_err, "worst_box_err_px": rep.worst_box_err_px,
                "score_tol": rep.score_tol, "box_tol_px": rep.box_tol_px, "saturated_values": rep.saturated_values,
                "max_abs_filter": rep.max_abs_filter, "max_abs_filter_name": rep.max_abs_filter_name.decode(errors="replace")}

    def result_block(self):
        p = C.c_void_p()
        n = C.c_int64()
        rc = lib().rtd_result_block(self._h, C.byref(p), C.byref(n))
        if rc != RTD_OK:
            _raise(rc, self._h)
        return p.value, n.value

    def sync(self):
        rc = lib().rtd_sync(self._h)
        if rc != RTD_OK:
            _raise(rc, self._h)

    def stream(self) -> int:
        return lib().rtd_stream(self._h) or 0

    def debug_tensor(self, name: str) -> np.ndarray:
        shape = (C.c_int64 * 4)()
        rc = lib().rtd_debug_tensor(self._h, name.encode(), None, 0, shape)
        if rc != RTD_OK:
            _raise(rc, self._h)
        out = np.zeros(tuple(shape), np.float32)
        rc = lib().rtd_debug_tensor(self._h, name.encode(), out.ctypes.data, out.size, shape)
        if rc != RTD_OK:
            _raise(rc, self._h)
        return out

    def force_topk(self, idx: Optional[np.ndarray]):
        if idx is None:
            rc = lib().rtd_debug_force_topk(self._h, None, 0)
        else:
            a = np.ascontiguousarray(idx, np.int32)
            rc = lib().rtd_debug_force_topk(self._h, a.ctypes.data, a.shape[0])
        if rc != RTD_OK:
            _raise(rc, self._h)

    def profile(self, n: int, reps: int = 5):
        cnt = C.c_int32()
        rc = lib().rtd_profile(self._h, n, reps, None, 0, C.byref(cnt))
        if rc != RTD_OK:
            _raise(rc, self._h)
        arr = (RtdLayerTime * cnt.value)()
        rc = lib().rtd_profile(self._h, n, reps, arr, cnt.value, C.byref(cnt))
        if rc != RTD_OK:
            _raise(rc, self._h)
        return [dict(name=a.name.decode(), kernel=a.kernel.decode(), ms=a.ms, flops=a.flops, bytes=a.bytes) for a in arr]

    def arena_bytes(self) -> int:
        return lib().rtd_arena_bytes(self._h)
