/*
 * rtdetr_mi355.h - C ABI of libmi355rtdetr.so: RT-DETRv2 detection inference on MI355X (gfx950).
 *
 * Drop-in boundary.  The reference has no FFI or plugin registry; its boundary is the duck-typed
 * Python class `RTDETRDetector` constructed by name in the inference engine
 * (/root/reference/src/inference_engine_yolox.py:196-212, class body src/rtdetr_detector.py:26-425).
 * This library is what the replacement class (telescope_cam_detection_amd/rtdetr_detector.py) binds
 * with ctypes; every entry point below cites the reference interface it stands in for.
 * Plain pointers and sizes only - no torch types cross this boundary.
 *
 * Threading: handles are independent; calls on one handle are serialised internally; every entry
 * point selects the handle's device and uses the handle's own HIP stream (the reference runs one
 * detector instance per camera thread, src/inference_engine_yolox.py:320-381).
 */
#ifndef RTDETR_MI355_H
#define RTDETR_MI355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rtd_engine* rtd_handle;

/* return codes; RTD_E_OOM is distinct so the Python shim can re-raise torch.cuda.OutOfMemoryError,
 * which is the only exception the reference's degrade path reacts to
 * (src/inference_engine_yolox.py:607-623). */
enum {
  RTD_OK = 0,
  RTD_E_INVALID = 1, /* bad argument / unsupported shape */
  RTD_E_OOM = 2,     /* hipMalloc / arena exhaustion */
  RTD_E_HIP = 3,     /* any other HIP runtime error */
  RTD_E_WEIGHTS = 4, /* blob malformed or tensor missing / wrong shape */
  RTD_E_STATE = 5    /* call order (e.g. infer before load_weights) */
};

/* RTD_PREC_F16X3 (the default engine): every trunk activation and filter is a hi + lo pair of IEEE fp16 (x = hi + lo to 2^-22 relative,
 * 2^-25 absolute below 2^-3, saturating at +-65504) and every contraction runs as three fp16 MFMAs (hi*hi + hi*lo + lo*hi, fp32
 * accumulate): fp32-grade results at the 16-bit MFMA rate / 3.  It meets the reference tolerance (1e-3 on scores, 1e-2 px on boxes
 * against fp32 eager, src/rtdetr_detector.py:256-257) on 640-px AND 1280-px frames (tests/test_gpu_parity.py) at >= 1000 frames/s.
 * RTD_PREC_BF16: plain bf16 storage, one MFMA per product - faster, outside that tolerance (~1 px).  RTD_PREC_FP32: the reference
 * arithmetic on fp32 MFMAs. */
enum { RTD_PREC_BF16 = 0, RTD_PREC_FP32 = 1, RTD_PREC_F16X3 = 2 };
enum { RTD_LAYER_BASIC = 0, RTD_LAYER_BOTTLENECK = 1 };

/* Constructor arguments of RTDETRDetector (src/rtdetr_detector.py:29-58) that matter to the device
 * side, plus the network description the reference obtains from upstream's YAML config
 * (src/rtdetr_detector.py:132).  Field order is ABI; struct_size guards it. */
enum { RTD_PROFILE_LATENCY = 0, RTD_PROFILE_THROUGHPUT = 1 };

typedef struct rtd_config {
  int32_t struct_size;      /* = sizeof(rtd_config) */
  int32_t device;           /* HIP ordinal  <- `device="cuda:N"` (:33) */
  int32_t precision;        /* RTD_PREC_* : storage/MFMA type of conv + large token GEMMs */
  int32_t max_batch;        /* largest n accepted by rtd_infer* */
  int32_t input_h, input_w; /* <- `input_size` (:35); must be multiples of 32 */
  int32_t use_graph;        /* 1: one hipGraph per batch size, built node by node from the plan (never stream-captured), replayed per call */
  /* architecture (HF:rt_detr/configuration_rt_detr_resnet.py, rt_detr_v2/configuration_rt_detr_v2.py) */
  int32_t layer_type;       /* RTD_LAYER_* */
  int32_t depths[4];
  int32_t hidden_sizes[4];
  int32_t embedding_size;
  int32_t enc_dim, enc_ffn, enc_heads, csp_hidden;
  int32_t d_model, dec_ffn, dec_heads, dec_layers;
  int32_t num_queries, num_classes, n_levels, n_points;
  float offset_scale;
  /* RTD_PROFILE_*: which way the kernel dispatch leans.  LATENCY (0): one batch in flight owns the GPU - tiles sized so that
   * every launch fills the CUs with the shortest critical path.  THROUGHPUT (1): several handles keep batches in flight on one
   * GPU (batching pipeline_depth > 1, bench.py --streams > 1) - other launches fill idle CUs anyway, so the convs take the
   * 256-pixel tile with the least LDS traffic per MFMA (+2.6 % frames/s with 3 handles, -6 % for a lone handle). */
  int32_t profile;
} rtd_config;

/* One detection row: what the per-row loop of src/rtdetr_detector.py:267-303 emits before it
 * becomes a Python dict (class_name / area are derived host-side from these). */
typedef struct rtd_det {
  int32_t class_id;
  float score;
  float x1, y1, x2, y2;
} rtd_det;

const char* rtd_version(void);

/* RTDETRDetector.__init__ (src/rtdetr_detector.py:29-58): allocates nothing on the device yet. */
int rtd_create(const rtd_config* cfg, rtd_handle* out);

/* RTDETRDetector.load_model (src/rtdetr_detector.py:132-173: load_state_dict + .deploy() + .to(device)).
 * `blob` is the flat container written by telescope_cam_detection_amd.weights.pack_blob (already
 * BN-folded / RepVGG-fused fp32 tensors); it is copied, converted to the handle's precision and laid
 * out for the kernels.  Builds the execution plan + activation arena for cfg.input_h x input_w. */
int rtd_load_weights(rtd_handle h, const void* blob, size_t nbytes);

/* RTDETRDetector.detect / detect_batch (src/rtdetr_detector.py:238-403) for n <= max_batch frames.
 * frames[i] : HWC uint8 BGR, hw[2*i] rows x hw[2*i+1] cols (the capture contract,
 *             src/stream_capture.py:228-239); host pointers, or device pointers if frames_on_device.
 * Frames whose size differs from input_h x input_w are stretch-resized exactly as PIL's antialiased
 * bilinear `T.Resize` does (src/rtdetr_detector.py:176-180).
 * out[i*num_queries ...] receives counts[i] rows in descending score order after the confidence
 * threshold (:271) and the wildlife filter {0,14,15,16,21} (:277, src/coco_constants.py:23-29).
 * Blocks until the results are on the host (the reference's three .cpu() syncs, :263-265). */
int rtd_infer(rtd_handle h, int32_t n, const uint8_t* const* frames_bgr_hwc, const int32_t* hw,
              int32_t frames_on_device, float conf_threshold, int32_t wildlife_only,
              rtd_det* out, int32_t* counts);

/* The raw tuple `labels, boxes, scores = self.model(img, orig_size)` (src/rtdetr_detector.py:257):
 * labels[n][Q] (int32), boxes[n][Q][4] xyxy in original-frame pixels, scores[n][Q], descending. */
int rtd_infer_raw(rtd_handle h, int32_t n, const uint8_t* const* frames_bgr_hwc, const int32_t* hw,
                  int32_t frames_on_device, int32_t* labels, float* boxes, float* scores);

/* RTDETRDetector.preprocess (src/rtdetr_detector.py:206-236) as a value: BGR -> RGB, PIL-exact antialiased stretch to input_h x input_w
 * when the frame has another size, / 255 - written as [3][input_h][input_w] fp32 to out_chw_dev (DEVICE memory of the caller, e.g. a torch
 * tensor).  Synchronous.  detect / detect_batch never call it (the network reads the uint8 frames directly); it exists so that a caller
 * of the reference's preprocess() gets the tensor without a forward pass or a host round trip. */
int rtd_preprocess(rtd_handle h, const uint8_t* frame_bgr_hwc, int32_t frame_h, int32_t frame_w, int32_t frame_on_device, float* out_chw_dev);

/* Pipelined form of detect_batch (src/rtdetr_detector.py:307-403 called by the batcher, src/shared_inference_coordinator.py:250),
 * also used by the multi-camera shard and the benchmark: rtd_infer_async enqueues upload + preprocess + network + post-process on the
 * handle's stream and returns; rtd_collect blocks for the LAST submitted batch and returns rtd_infer's rows for it.  A handle has ONE
 * result block: submit, then collect.  A second rtd_infer_async before rtd_collect is legal - batches run in submission order on the
 * handle's stream (the benchmark's back-to-back loop) - but it overwrites the block, so the earlier batch's rows can no longer be
 * collected; with host frames (frames_on_device = 0) it first waits for the previous batch, because the handle's pinned staging
 * buffer holds one batch.  Host frames are copied into that buffer before the call returns (the caller's buffers are free again)
 * and reach HBM by one asynchronous DMA; device frames must stay alive until rtd_collect / rtd_sync.  A failed rtd_infer_async
 * drains the handle's stream before it returns: nothing of the failed batch is still reading the staging buffers.  Everything is plain HIP inside the library: no torch stream, event or
 * allocator takes part.  The result block also stays on the device: [n][Q][6] fp32 rows (label, score, x1, y1, x2, y2) - the
 * fixed-size block each rank contributes to the all-gather - see rtd_result_block. */
int rtd_infer_async(rtd_handle h, int32_t n, const uint8_t* const* frames_bgr_hwc, const int32_t* hw, int32_t frames_on_device);
int rtd_collect(rtd_handle h, float conf_threshold, int32_t wildlife_only, rtd_det* out, int32_t* counts);
int rtd_result_block(rtd_handle h, float** dev_ptr, int64_t* n_floats);
int rtd_sync(rtd_handle h);
void* rtd_stream(rtd_handle h); /* hipStream_t of the handle: for profilers / HIP-event timing only - never wrap it in a torch stream */

/* Build everything a later rtd_infer* of batch size n needs - plan, activation arena, one eager pass on blank frames, the hipGraph -
 * so that the serving path only replays (RTDETRDetector.load_model prepares the sizes its caller declares; part of
 * src/rtdetr_detector.py:132-173's "model ready after load_model").  A size that was not prepared is still built on first use. */
int rtd_prepare(rtd_handle h, int32_t n);

/* Ordering against a stream the CALLER owns (torch's current stream that produced device-resident frames,
 * src/stream_capture_gpu_ffmpeg.py:253,277-278; the stream RCCL's all-gather runs on).  rtd_wait_stream: the handle's stream waits for
 * everything enqueued on `producer_stream` so far.  rtd_signal_stream: `consumer_stream` waits for everything enqueued on the handle's
 * stream so far.  Both use an event that belongs to the handle; streams are hipStream_t values (NULL = the legacy default stream). */
int rtd_wait_stream(rtd_handle h, void* producer_stream);
int rtd_signal_stream(rtd_handle h, void* consumer_stream);

/* What the handle did so far - carried into error reports so that a failure describes itself (batching.BatchCoordinator.get_stats). */
typedef struct rtd_stats {
  int32_t struct_size;           /* = sizeof(rtd_stats) */
  int32_t last_error_code;       /* RTD_E_* of the most recent failed call, 0 = none */
  int32_t stream_capture_status; /* hipStreamIsCapturing of the handle's stream now: 0 none (the only value this library produces) */
  int32_t in_flight;             /* 1: a submitted batch has not been collected */
  int64_t plans, graphs, graph_nodes, graph_launches, eager_passes, submits, collects, failed_calls;
  int64_t saturated_values;      /* F16X2 activations found AT the format's saturation value (+-65504) by the last rtd_self_check, -1 = never run */
  float max_abs_filter;          /* largest |folded filter value| of the loaded blob (> 65504 is refused by rtd_load_weights on the pair engine) */
  int32_t reserved;
} rtd_stats;
int rtd_get_stats(rtd_handle h, rtd_stats* out);

/* Real-weights guard (part of "the model is ready after load_model", src/rtdetr_detector.py:132-173).  Every parity claim of this library
 * was measured on seeded synthetic weights; the pair engine's fp16 halves SATURATE at +-65504 - silently, finitely.  Two things keep a
 * trained checkpoint honest:
 *  (1) rtd_load_weights returns RTD_E_WEIGHTS for a tensor that holds NaN / Inf and - RTD_PREC_F16X3 - for a folded filter value
 *      beyond 65504 (rtd_last_error names the tensor);
 *  (2) rtd_self_check runs ONE built-in frame of the handle's input size through the library's exact fp32 engine and through an engine of
 *      the handle's precision, both built from the blob the handle was loaded with, counts the activations that sit at the saturation value and
 *      matches the detection rows (same label, |dscore| <= score_tol = 1e-3, max |dbox| <= box_tol_px = 1e-2, the reference tolerance).
 *      The caller decides: RTDETRDetector.load_model(verify=True) logs the report and refuses a checkpoint whose rows do not match.
 *      Costs two temporary bs-1 engines (a second or two at load time); the handle itself is not touched. */
typedef struct rtd_check_report {
  int32_t struct_size;             /* in: = sizeof(rtd_check_report) */
  int32_t rows, rows_matched;      /* rows of the fp32 engine's answer / of them matched by the handle's precision within the tolerance */
  float worst_score_err, worst_box_err_px;   /* over the matched rows */
  float score_tol, box_tol_px;
  float max_abs_filter;
  int64_t saturated_values;        /* F16X2 activations at +-65504 after that forward (0 for the other precisions) */
  char max_abs_filter_name[64];
} rtd_check_report;
int rtd_self_check(rtd_handle h, const void* blob, size_t nbytes, rtd_check_report* out);   /* blob: the one given to rtd_load_weights (the handle keeps no host copy) */

/* mutable attribute `model.to(device)` / teardown (src/inference_engine_yolox.py:743-744) */
void rtd_destroy(rtd_handle h);
const char* rtd_last_error(rtd_handle h); /* h may be NULL: last error of a failed rtd_create */

/* device memory held by the handle's activation arenas (all prepared batch sizes), bytes */
int64_t rtd_arena_bytes(rtd_handle h);

/* ---- Stage 2: crop + classifier pre-processing for a whole batch of detections ------------------
 * For crop i: frame slice [y1:y2, x1:x2] (src/two_stage_pipeline_yolox.py:289) of a device-resident HWC uint8 BGR frame,
 * then SpeciesClassifier.preprocess (src/species_classifier.py:298-352): BGR->RGB, F.interpolate(bilinear,
 * align_corners=False) to out_size x out_size, /255, (x-mean)/std.  out_dev: [n][3][out_size][out_size] fp32 (what the
 * classifier network consumes).  rects = [n][4] (x1, y1, x2, y2), frame_hw = [n][2].  ASYNCHRONOUS: the launch is
 * enqueued on `stream` (a hipStream_t the caller owns, e.g. torch's current stream; NULL = the legacy default stream) and the call
 * returns at once - the batch is ready when that stream reaches it, so a consumer on the same stream (the classifier forward) needs no
 * synchronisation and the next detect batch overlaps with it.  Frames and out_dev must stay alive until then; the rectangles, sizes and
 * normalisation constants are copied into the launch before the call returns. */
int rtd_crop_resize_batch(int32_t n, const uint8_t* const* frames_dev, const int32_t* frame_hw, const int32_t* rects,
                          int32_t out_size, const float* mean3, const float* std3, float* out_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RTDETR_MI355_H */
