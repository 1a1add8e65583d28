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

/* per-kernel timing record filled by rtd_profile */
typedef struct rtd_layer_time {
  char name[48];
  char kernel[24];  /* kernel family: "conv_igemm", "layernorm", ... */
  float ms;         /* mean HIP-event time over `reps` launches on the handle's stream */
  double flops;     /* algorithmic flops of one launch (2 flop / MAC) */
  double bytes;     /* algorithmic bytes of one launch (in + out + weights, unfused) */
} rtd_layer_time;

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
} rtd_stats;
int rtd_get_stats(rtd_handle h, rtd_stats* out);

/* mutable attribute `model.to(device)` / teardown (src/inference_engine_yolox.py:743-744) */
void rtd_destroy(rtd_handle h);
const char* rtd_last_error(rtd_handle h); /* h may be NULL: last error of a failed rtd_create */

/* ---- introspection used by tests / bench (no reference counterpart) -------------------------- */
/* copy a named intermediate tensor of the last forward to the host as fp32; shape = (n, h, w, c) */
int rtd_debug_tensor(rtd_handle h, const char* name, float* out, int64_t capacity, int64_t shape[4]);
/* force the encoder top-k selection of the next forwards (idx[n][Q] memory-token ids, NULL = off):
 * lets stage-level parity tests separate selection flips from decoder arithmetic */
int rtd_debug_force_topk(rtd_handle h, const int32_t* idx, int32_t n);
/* time every kernel of one forward of batch n with HIP events on the handle's stream */
int rtd_profile(rtd_handle h, int32_t n, int32_t reps, rtd_layer_time* out, int32_t capacity, int32_t* count);
int64_t rtd_arena_bytes(rtd_handle h);
/* A/B switches for tests and profiling (defaults in brackets; unknown names return RTD_E_INVALID).  Every switch edits a process-wide
 * TEMPLATE that rtd_create snapshots into the handle: a call changes handles created AFTERWARDS (and the kernel-level rtd_op_* /
 * rtd_bench_* entry points below, which read the template when called) - never a live handle, so two handles of one process cannot see
 * each other's settings and all plans of a handle (one per batch size, built lazily) agree with each other.
 * Conv dispatch, bf16 / fp32 operands (csrc/common.h ConvOpts):
 *   conv_mode [0]   0 auto | 1 register-staged fallback kernel only | 3, 4 wave-specialised LDS-DMA tile with 4 / 2 stages everywhere |
 *                   7 256-pixel tile | 8 A-stationary kernel wherever eligible | 9 streaming 1x1 kernels whatever the grid size |
 *                   10 128 x 64 tile everywhere
 *   conv_reg [3: bit 0 direct 3x3 kernels for the narrow stem / stage-0 layers, bit 1 the 64-channel pair kernel], conv_stream [1], stream2 [1],
 *   stream2_max_n [2048], stream_slab [1], stream_min_tiles [2048], wsa_min_ntn [8], ws2_min_blocks [257], ws64_max_blocks [160],
 *   ws256_min_blocks [0], glds_min_blocks [4], glds_min_n [128], reg_epilogue [1], prefetch [1],
 *   glds_drop [0: timing-only probes, results wrong when set]
 * Conv dispatch, pair operands (RTD_PREC_F16X3): split_ws2_min_blocks [257] | split_ws64_max_blocks [160] |
 *   split_flex [1: flexible tile heights on grids of <= split_flex_small_max [200] tiles], split_flex_min_nk [4] |
 *   split_k2 [1: two-pass split-K on >= 128 K-steps with <= 16 tiles per image] |
 *   split_wsq [1: 160..256-pixel tiles at one block per CU on grids of >= split_wsq_min_blocks [257] tiles with >= split_wsq_min_nk [24] K-steps] |
 *   split_sx [3: streaming 1x1 kernel 0 off, 1 stage-0 shapes, 2 + K = 128, 3 + K = 256 -> N >= 1024 (value projection), 4 + with residual (slower)]
 * Plan building:
 *   sc_fold [1] projection shortcut folded into the block's last conv | up_fold [1] FPN upsample folded into the CSP's first conv |
 *   c1_fuse [1] a block's reduce conv computed inside the previous block's expand conv (bf16: stage 0/1; f16x3: stage 0 and the first block
 *   of stage 1) | attn_split [2] self-attention on fp16-pair MFMAs (bit 0 AIFI, bit 1 decoder) |
 *   arena_reuse [1] | stem_fused_split [1] (f16x3): stem.0 straight from the uint8 frames | stem_pool_fuse [1] (f16x3): stem.2 and the 3x3/s2
 *   max-pool in one pass | avg_fuse [1] (f16x3): a stage's last expand conv also writes the next stage's vd-shortcut average |
 *   aifi_pair [1] (f16x3): the un-fused AIFI's linears on the pair kernels | side_stream [7: bit 0 query
 *   selection on a second stream beside the value projection, bit 1 decoder input projections beside the PAN path, bit 2 encoder input
 *   projections beside stages 2 / 3 and AIFI] | dec_fused [1],
 *   dec_split [1: 0 fp32 MFMA, 2 hi-only filters], sel_fused [1] | dec_stamps [0]
 * Tools: profile_twice [0], bench_rewarm [0: bit 5 = rtd_bench_conv fills its operands with random fp16 values instead of zeros].
 * "reset" (any value): every template back to the values in brackets. */
int rtd_debug_option(const char* name, int value);

/* ---- Stage 2: crop + classifier pre-processing for a whole batch of detections ------------------
 * For crop i: frame slice [y1:y2, x1:x2] (src/two_stage_pipeline_yolox.py:289) of a device-resident HWC uint8 BGR frame,
 * then SpeciesClassifier.preprocess (src/species_classifier.py:298-352): BGR->RGB, F.interpolate(bilinear,
 * align_corners=False) to out_size x out_size, /255, (x-mean)/std.  out_dev: [n][3][out_size][out_size] fp32 (what the
 * classifier network consumes).  rects = [n][4] (x1, y1, x2, y2), frame_hw = [n][2].  Enqueued on `stream` (may be NULL);
 * returns after the work completed (the reference's classify() is synchronous too). */
int rtd_crop_resize_batch(int32_t n, const uint8_t* const* frames_dev, const int32_t* frame_hw, const int32_t* rects,
                          int32_t out_size, const float* mean3, const float* std3, float* out_dev, void* stream);

/* ---- kernel-level test entry points (device pointers; dtype 0 = bf16, 1 = fp32, 4 = F16X2: hi/lo fp16 pairs in 32-channel groups
 * [32 hi | 32 lo], 4 bytes per channel - the storage of RTD_PREC_F16X3, rtd_op_conv / rtd_op_conv_dual only) --------------- */
int rtd_op_conv(int dtype, const void* x, const void* w_ohwi_f32, const float* bias, const void* res,
                void* y, int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                int act, int res_mode, int out_f32);
/* conv with a second input x2 [B,OH,OW,C2] read as an extra 1x1 tap at output resolution; w_f32 = [Cout][KH*KH*Cin + C2]
 * (how the plan folds a bottleneck's projection shortcut into its last conv).  x_up2 = 1 (1x1 only): x is [B,H/2,W/2,Cin] and is
 * read through a nearest 2x upsampling, H and W being the output extents (the FPN's conv over cat([upsample(lat), proj])). */
int rtd_op_conv_dual(int dtype, const void* x, const void* x2, const void* w_f32, const float* bias, const void* res,
                     void* y, int B, int H, int W, int Cin, int C2, int Cout, int KH, int stride, int pad,
                     int act, int res_mode, int out_f32, int x_up2);
/* 1x1 conv (optionally with a second input, as rtd_op_conv_dual) with the FOLLOWING 1x1 conv Cout -> Cnext fused into the launch
 * (how the plan runs a bottleneck's reduce conv inside the previous block's expand conv): y = act(W [x | x2] + b (+ res)),
 * y1 = next_act(W1 y + b1), both written.  dtype 1 (bf16) / 4 (f16x2); RTD_E_INVALID for shapes the streaming kernels do not take. */
int rtd_op_conv_next(int dtype, const void* x, const void* x2, const void* w_f32, const float* bias, const void* res, void* y,
                     const void* w1_f32, const float* bias1, void* y1, int B, int H, int W, int Cin, int C2, int Cout, int Cnext,
                     int act, int res_mode, int next_act);
int rtd_op_layernorm(int dtype, const void* x, const void* res, const float* g, const float* b,
                     void* y, int rows, int dim, int out_f32);
int rtd_op_attention(int dtype, const void* qk, const void* v, void* o, int B, int L, int heads, int hd);
int rtd_op_msdeform(int dtype, const void* value, const float* offaw, const float* ref, float* out,
                    int B, int Q, int heads, int hd, int n_levels, int n_points, const int32_t* level_hw,
                    int value_ld, float offset_scale);
int rtd_op_topk(const float* keys, int B, int N, int K, int32_t* idx_out, float* val_out);
int rtd_op_resize(const uint8_t* src, int sh, int sw, void* dst, int dh, int dw, int dtype);
/* kernel micro-benchmark (tools/conv_bench.py): one conv layer on zero-filled buffers, timed with HIP events.
 * us_out[0] = mean of `reps` back-to-back launches (operands warm in L2 / Infinity Cache),
 * us_out[1] = mean of `reps` launches each preceded by a `flush_mb` MiB memset (operands come from HBM). */
/* two convs on two streams: shape = {B, HW, Cin, Cout, K, stride, pad}; us_out = {A alone, B alone, A and B together} per repetition */
int rtd_bench_conv_pair(const int* shape_a, const int* shape_b, int reps, float* us_out);
int rtd_bench_conv(int dtype, int B, int H, int W, int Cin, int Cout, int KH, int stride, int pad, int with_res,
                   int reps, int flush_mb, float* us_out);

/* What the matrix pipes of THIS device sustain (tools/mfma_rate_probe.hip inside the library; bench.py reports it beside `roofline`):
 * v_mfma_f32_16x16x32_f16 back to back on every CU, operands in registers, `random_operands` 0 = all-zero bits / 1 = random finite fp16
 * (the chip lowers its clock under matrix load on real data).  out[0] = TFLOP/s by HIP events, out[1] = in-kernel core clock in GHz
 * (s_memtime per s_memrealtime), out[2] = kernel milliseconds.  ~`ms_target` milliseconds of work per timed launch (3 launches). */
int rtd_bench_mfma_rate(int random_operands, int ms_target, float* out);

#ifdef __cplusplus
}
#endif
#endif /* RTDETR_MI355_H */
