/* rtdetr_mi355_test.h - kernel-level TEST, BENCH and DEBUG entry points of libmi355rtdetr.so.
 *
 * Nothing in this header has a counterpart in the reference and nothing here is on the detection path: the product API
 * (what a reference-side binding uses) is rtdetr_mi355.h.  Callers: tests/, tools/ and the diagnostic legs of bench.py
 * (per-kernel timing for the roofline object, the sustained-MFMA probe).  Implemented in csrc/testapi.hip.
 */
#ifndef RTDETR_MI355_TEST_H
#define RTDETR_MI355_TEST_H

#include "rtdetr_mi355.h"

#ifdef __cplusplus
extern "C" {
#endif

/* per-kernel timing record filled by rtd_profile */
typedef struct rtd_layer_time {
  char name[48];
  char kernel[24];  /* kernel family: "conv_igemm", "layernorm", ... */
  float ms;         /* mean HIP-event time over `reps` launches on the handle's stream */
  double flops;     /* algorithmic flops of one launch (2 flop / MAC) */
  double bytes;     /* algorithmic bytes of one launch (in + out + weights, unfused) */
} rtd_layer_time;

/* ---- introspection used by tests / bench (no reference counterpart) -------------------------- */
/* copy a named intermediate tensor of the last forward to the host as fp32; shape = (n, h, w, c) */
int rtd_debug_tensor(rtd_handle h, const char* name, float* out, int64_t capacity, int64_t shape[4]);
/* force the encoder top-k selection of the next forwards (idx[n][Q] memory-token ids, NULL = off):
 * lets stage-level parity tests separate selection flips from decoder arithmetic */
int rtd_debug_force_topk(rtd_handle h, const int32_t* idx, int32_t n);
/* time every kernel of one forward of batch n with HIP events on the handle's stream */
int rtd_profile(rtd_handle h, int32_t n, int32_t reps, rtd_layer_time* out, int32_t capacity, int32_t* count);
/* A/B switches for tests and profiling (defaults in brackets; unknown names return RTD_E_INVALID).  Every switch edits a process-wide
 * TEMPLATE that rtd_create snapshots into the handle: a call changes handles created AFTERWARDS (and the kernel-level rtd_op_* /
 * rtd_bench_* entry points below, which read the template when called) - never a live handle, so two handles of one process cannot see
 * each other's settings and all plans of a handle (one per batch size, built lazily) agree with each other.
 * Conv dispatch, bf16 / fp32 operands (csrc/common.h ConvOpts; one tile family since round 5):
 *   conv_mode [0]   0 auto | 1 register-staged fallback kernel only | 3, 4 wave-specialised LDS-DMA tile with 4 / 2 stages everywhere |
 *                   10 128 x 64 tile everywhere
 *   ws2_min_blocks [257], ws64_max_blocks [160], glds_min_blocks [4], glds_min_n [128], reg_epilogue [1], prefetch [1],
 *   glds_drop [0: timing-only probes, results wrong when set]
 *   conv_reg [3] (pair operands too): bit 0 direct 3x3 kernels for the narrow stem / stage-0 layers, bit 1 the 64-channel pair kernel
 * Conv dispatch, pair operands (RTD_PREC_F16X3): split_ws2_min_blocks [257] | split_ws64_max_blocks [160] |
 *   split_flex [1: flexible tile heights on grids of <= split_flex_small_max [200] tiles], split_flex_min_nk [4] |
 *   split_k2 [1: two-pass split-K on >= 128 K-steps with <= 16 tiles per image] |
 *   split_wsq [1: 160..256-pixel tiles at one block per CU on grids of >= split_wsq_min_blocks [257] tiles with >= split_wsq_min_nk [24] K-steps] |
 *   split_sx [3: streaming 1x1 kernel 0 off, 1 stage-0 shapes, 2 + K = 128, 3 + K = 256 -> N >= 1024 (value projection), 4 + with residual (slower)]
 * Plan building:
 *   sc_fold [1] projection shortcut folded into the block's last conv | up_fold [1] FPN upsample folded into the CSP's first conv |
 *   c1_fuse [1] a block's reduce conv computed inside the previous block's expand conv (bf16: stage 0/1; f16x3: stage 0 and the first block
 *   of stage 1) | attn_split [3] self-attention on fp16-pair MFMAs (bit 0 the fused AIFI layer, bit 1 decoder) |
 *   arena_reuse [1] | stem_fused_split [1] (f16x3): stem.0 straight from the uint8 frames | stem_pool_fuse [1] (f16x3): stem.2 and the 3x3/s2
 *   max-pool in one pass | avg_fuse [1] (f16x3): a stage's last expand conv also writes the next stage's vd-shortcut average |
 *   post_fused [1] sigmoid + top-k + box decode of the post-processor in one launch |
 *   dead_out [1] (f16x3): the stage-0 output is not written when its only readers are that launch's fused follower conv and fused average |
 *   aifi_pair [1] (f16x3): the un-fused AIFI's linears on the pair kernels | side_stream [7: bit 0 query
 *   selection on a second stream beside the value projection, bit 1 decoder input projections beside the PAN path, bit 2 encoder input
 *   projections beside stages 2 / 3 and AIFI] | dec_fused [1],
 *   dec_split [1: 0 fp32 MFMA, 2 hi-only filters], sel_fused [1] | dec_stamps [0]
 * Tools: profile_twice [0], bench_rewarm [0: bit 5 = rtd_bench_conv fills its operands with random fp16 values instead of zeros].
 * "reset" (any value): every template back to the values in brackets. */
int rtd_debug_option(const char* name, int value);

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
#endif /* RTDETR_MI355_TEST_H */
