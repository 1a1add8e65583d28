// common.h - shared types for the gfx950 RT-DETR engine (host + device).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <stdexcept>
#include <string>
#include <tuple>
#include <type_traits>
#include <utility>
#include <vector>

namespace rtd {

// F16X2 ("split"): every value is carried as two IEEE fp16, x = hi + lo with hi = fp16(x), lo = fp16(x - hi) (22 significant bits; see sp16 below).
// Storage: a pixel's channels in groups of 32, each group 128 bytes = [32 x hi | 32 x lo]; 4 bytes per channel, so `ld`, `c` and
// slice_c() count real channels exactly as for F32 (slices on multiples of 32 channels).  A K-step of the LDS-DMA conv kernels (128
// bytes per pixel row) is then one channel group, and hi*hi + hi*lo + lo*hi runs as three fp16 MFMAs on the same staged bytes
// (rtd_config.precision = RTD_PREC_F16X3: fp32-grade products at 3/16 of the fp32 MFMA cost).
enum DType : int { BF16 = 0, F32 = 1, U8 = 2, I32 = 3, F16X2 = 4 };
constexpr int SPLIT_GROUP = 32;   // channels per [hi | lo] group of a F16X2 tensor
enum Act : int { ACT_NONE = 0, ACT_RELU = 1, ACT_SILU = 2, ACT_GELU = 3 };
enum ResMode : int { RES_NONE = 0, RES_PRE = 1, RES_POST = 2 };

inline size_t dtype_size(int dt) { return dt == BF16 ? 2 : (dt == U8 ? 1 : 4); }

typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
// element type of a pair tensor (F16X2 / "f16x3" engine): IEEE fp16.  hi = fp16(x), lo = fp16(x - hi): 22 significant bits for
// |x| >= 2^-3, an ABSOLUTE error <= 2^-25 below that (lo goes subnormal; gfx950's matrix cores keep fp16 subnormal operands -
// tools/f16_denorm_probe.hip), saturating at +-65504.  Round 2 carried bf16 pairs (2^-17 relative): measured 6x further from the fp32
// reference on R101 1280-px frames (tools/pair_sim.py), where 1e-2 px is 7.8e-6 of the frame.
typedef _Float16 sp16;
typedef _Float16 sp16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 sp16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// An NHWC view.  `ld` is the element stride between consecutive pixels (>= c when the view is a
// channel slice of a wider buffer), `bstride` the element stride between images.
struct Tensor {
  void* p = nullptr;
  int dt = F32;
  int n = 0, h = 0, w = 0, c = 0;
  int64_t ld = 0;
  int64_t bstride = 0;
  int64_t pixels() const { return (int64_t)n * h * w; }
  Tensor slice_c(int c0, int cn) const {   // F16X2: c0 must be a multiple of SPLIT_GROUP (group g lives at byte 128 g)
    Tensor t = *this;
    t.p = (char*)p + (size_t)c0 * dtype_size(dt);
    t.c = cn;
    return t;
  }
};

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

#define RTD_CHECK(cond, code, msg)                                                     \
  do {                                                                                 \
    if (!(cond)) throw ::rtd::Error((code), std::string(msg) + " [" #cond "] at " __FILE__ ":" + std::to_string(__LINE__)); \
  } while (0)

#define HIP_CHECK(expr)                                                                \
  do {                                                                                 \
    hipError_t _e = (expr);                                                            \
    if (_e != hipSuccess) {                                                            \
      (void)hipGetLastError(); /* clear the runtime's sticky last-error: the next launch's hipGetLastError() must not see this one */ \
      int _c = (_e == hipErrorOutOfMemory) ? 2 : 3;                                    \
      throw ::rtd::Error(_c, std::string(#expr) + ": " + hipGetErrorString(_e) + " at " __FILE__ ":" + std::to_string(__LINE__)); \
    }                                                                                  \
  } while (0)

// ------------------------------------------------------------------------------------------
// Kernel launches.  Every launcher of the library goes through rtd_launch().  On a live stream it is hipLaunchKernelGGL.  While a plan's
// hipGraph is being BUILT (engine.hip build_exec: a GraphBuild is active on the calling thread) nothing is enqueued and no stream is
// involved: the launch becomes a kernel node, chained behind the last node of its lane (lane 0 = the engine's stream, lane 1 = its side
// stream; fork / join markers of the plan become dependency edges).  The library therefore NEVER puts a stream into capture mode: the
// states HIP attaches to capturing streams and to events recorded on them (hipErrorStreamCaptureUnsupported, hipErrorCapturedEvent - the
// failures rounds 2 and 3 met in processes that also run torch's allocator and RCCL's watchdog on other threads) cannot arise from here.
struct GraphBuild {
  hipGraph_t graph = nullptr;
  hipStream_t lane_stream[2] = {nullptr, nullptr};
  std::vector<hipGraphNode_t> tail[2];      // nodes the lane's next node depends on
  int nodes = 0;
  void add_kernel(void* fn, dim3 grid, dim3 block, unsigned shmem, void** params, hipStream_t s) {
    const int lane = (s == lane_stream[1] && lane_stream[1] != nullptr) ? 1 : 0;
    if (lane == 0 && s != lane_stream[0]) throw Error(3, "graph build: a launch names a stream that is neither lane of the plan");
    hipKernelNodeParams kp;
    memset(&kp, 0, sizeof kp);
    kp.func = fn; kp.gridDim = grid; kp.blockDim = block; kp.sharedMemBytes = shmem; kp.kernelParams = params; kp.extra = nullptr;
    hipGraphNode_t node = nullptr;
    const hipError_t er = hipGraphAddKernelNode(&node, graph, tail[lane].empty() ? nullptr : tail[lane].data(), tail[lane].size(), &kp);
    if (er != hipSuccess) { (void)hipGetLastError(); throw Error(3, std::string("hipGraphAddKernelNode: ") + hipGetErrorString(er)); }
    tail[lane].assign(1, node);
    ++nodes;
  }
  void fork() { for (hipGraphNode_t n : tail[0]) tail[1].push_back(n); }     // the side lane waits for everything on the main lane so far
  void join() { for (hipGraphNode_t n : tail[1]) tail[0].push_back(n); tail[1].clear(); }   // the main lane waits for the side lane
};
extern __thread GraphBuild* t_graph_build;   // engine.hip
extern __thread long long t_launches;        // launches (or nodes) issued by this thread: build_exec checks node count == eager launch count

template <typename... KArgs, typename... Args>
inline void rtd_launch(void (*kernel)(KArgs...), dim3 grid, dim3 block, unsigned shmem, hipStream_t s, Args&&... args) {
  static_assert(sizeof...(KArgs) == sizeof...(Args), "rtd_launch: argument count differs from the kernel's parameter count");
  ++t_launches;
  if (GraphBuild* gb = t_graph_build) {
    std::tuple<std::remove_cv_t<KArgs>...> vals{static_cast<KArgs>(std::forward<Args>(args))...};   // the node copies the values now
    void* ptrs[sizeof...(KArgs) + 1];
    std::apply([&](auto&... v) { size_t i = 0; ((ptrs[i++] = (void*)&v), ...); (void)i; }, vals);
    gb->add_kernel((void*)kernel, grid, block, shmem, ptrs, s);
  } else {
    hipLaunchKernelGGL(kernel, grid, block, shmem, s, static_cast<KArgs>(std::forward<Args>(args))...);
  }
}

// ------------------------------------------------------------------------------------------
// kernel launch wrappers (implemented in conv_igemm.hip / ops.hip)
// ------------------------------------------------------------------------------------------
// Dispatch switches of the conv family (rtd_debug_option names in the comments).  A handle SNAPSHOTS the process-wide template
// (conv_opts_template(), which rtd_debug_option edits) at rtd_create and every launch of its plans carries that snapshot
// (ConvArgs::opts): a later rtd_debug_option call, or the settings another handle was created with, cannot change what a live handle
// runs.  The kernel-level entry points (rtd_op_*, rtd_bench_*) read the template at call time.
struct ConvOpts {
  // ---- bf16 / fp32 operands
  int conv_mode = 0;          // 0 auto; test / A-B modes: 1 = register-staged fallback kernel only, 3 / 4 = wave-specialised tile with 4 / 2 stages
                              // everywhere, 7 = 256-pixel tile, 8 = A-stationary kernel, 9 = streaming kernels on any grid, 10 = 128 x 64 tile
  int glds_min_blocks = 4;    // bf16: smallest grid of 128 x 128 tiles the LDS-DMA kernels take (fp32: 512, fixed)
  int glds_min_n = 128;       // bf16: smallest Cout the LDS-DMA kernels take (64 measured slower on the stage-0 reduce convs: 45 vs 42 us)
  int ws2_min_blocks = 257;   // grids that do not fit one block per CU run the 2-stage kernel at 2 blocks per CU
  int ws64_max_blocks = 160;  // 128 x 128-tile grids below this take the 128 x 64 tile (0 = never)
  int reg_epilogue = 1;       // residual-free bf16 tiles finish in registers
  int conv_reg = 3;           // direct 3x3 kernels for the narrow layers (bit 1: the 64-channel pair kernel)
  int prefetch = 1;           // 0 = launches ignore ConvArgs::pf
  int glds_drop = 0;          // timing-only probes (results wrong): 1 = x descriptor has 0 records, 2 = w, 4 = no DMA at all, 32 = block stamps
  // ---- pair (F16X2) operands
  int split_ws2_min_blocks = 257;
  int split_ws64_max_blocks = 160;
  int split_flex = 1;             // flexible tile heights (conv_igemm_wsf_kernel) on grids of <= split_flex_small_max 128 x 128 tiles
  int split_flex_min_nk = 4;      // ... from this many K-steps on (same-box sweep, R50 bs 8: 16 / 8 / 4 / 2 -> 4.884 / 4.878 / 4.854 / 4.866 ms)
  int split_flex_small_max = 200; // (sweep 128 / 200 / 256 / 400 -> 4.816 / 4.776 / 4.779 / 4.777 ms per step)
  int split_sx = 3;               // streaming pair kernel: 0 off, 1 = K = 64 (+ 64) -> 256 (stage 0), 2 = also K = 128 (stage 1), 3 = also K = 256 -> N >= 1024
                                  // without a residual (value projection), 4 = also with a residual from 40^2 maps on (51.5 vs 39.5 us on the tiled kernel: off)
  int split_k2 = 1;               // two-pass split-K on long-K layers with few tiles per image
  int split_wsq = 1;              // 160..256-pixel tiles at one block per CU (conv_igemm_wsq_kernel) on grids of >= split_wsq_min_blocks 128 x 128 tiles
                                  // with >= split_wsq_min_nk K-steps (2 = on every grid: tests)
  int split_wsq_min_blocks = 257;
  int split_wsq_min_nk = 24;         // same-box A/B (tools/profile_layers.py --ab split_wsq): 3x3 layers 1.05-1.18x, 1x1 layers of 8-16 K-steps 0.80-0.96x
};
ConvOpts& conv_opts_template();
bool conv_set_option(const char* name, int value);   // edits the template; false = not a conv option

// scratch of one plan for the two-pass split-K of the pair kernels (fp32 partial sums); launches on one stream are sequential
struct ConvWorkspace {
  float* slab = nullptr;
  size_t slab_bytes = 0;
};

struct ConvArgs {
  Tensor x;            // input  [B,H,W,Cin] (view)
  const void* w;       // filter [Npad][Kpad] in x.dt, K = KH*KW*Cin (tap-major, channel-minor)
  const float* bias;   // [Npad] fp32
  Tensor res;          // optional residual, indexed like y
  // optional SECOND input [B,OH,OW,C2] read as an extra 1x1 / stride 1 tap at output resolution: the filter rows are
  // [taps of x | C2 of x2] (K = KH*KW*Cin + C2).  This is how a bottleneck's projection shortcut is folded into its last conv:
  // y = act(W3 * t + Wsc * x_in + (b3 + bsc)) without writing / re-reading the shortcut tensor.  KH*KW*Cin must be a multiple of
  // one K-step (64 bf16 / 32 fp32); see conv_dual_supported().
  Tensor x2;
  // x is read through a nearest-neighbour 2x upsampling: x is [B,OH/2,OW/2,Cin], output pixel (oy,ox) reads x(oy/2, ox/2).  1x1 /
  // stride 1 / pad 0 with a second input only (the FPN's `cat([upsample(lat), proj])` -> 1x1 conv without the upsampled tensor).
  int x_up2 = 0;
  // optional FOLLOWING 1x1 conv fused into this launch (streaming kernel only, see conv_next_supported()): after a 32-pixel tile of
  // y = act(...) is complete, the block also computes y_next = act_next(W_next * y + b_next) for those pixels from the tile it
  // still holds in LDS - the next block's reduce conv never re-reads y from HBM (stage 0: 256 -> 64 channels, 105 MB per launch)
  const void* next_w = nullptr;     // [Npad][Kpad] bf16, K = y.c
  const float* next_bias = nullptr;
  Tensor next_y;                    // [B,OH,OW,64]
  int next_kpad = 0, next_act = ACT_NONE;
  // optional (streaming pair kernel, a stage's last expand conv): the 2 x 2 / stride-2 average of y, [B, OH / 2, OW / 2, N] F16X2, written by
  // the same launch (the next stage's vd-shortcut input); see conv_avg_supported()
  Tensor avg_y;
  // optional (streaming pair kernel only; every other kernel ignores it and writes y): nothing reads y after this launch - its consumers
  // are the fused following conv (next_y, from the tile in LDS) and the fused 2 x 2 average (avg_y): the stores of y are dropped
  // (buffer descriptor of zero bytes).  A bottleneck net's stage-0 output at the stage-1 boundary: 210 MB per R50 bs-8 step.
  int y_dead = 0;
  int prefer256 = 0;   // throughput profile (rtd_config.profile): take the 256-pixel tile from 100 blocks on (bf16 / fp32 operands)
  const ConvOpts* opts = nullptr;   // the handle's snapshot of the dispatch switches; nullptr = the process-wide template
  Tensor y;            // output [B,OH,OW,N] (view)
  int KH = 1, KW = 1, stride = 1, pad = 0;
  int Kpad = 0, Npad = 0;
  int act = ACT_NONE, res_mode = RES_NONE;
  ConvWorkspace ws;    // optional: enables the two-pass split-K (pair kernels)
  // optional: bytes the launch pulls towards the Infinity Cache for a LATER launch (the next layer's filter): one dword per
  // 128-byte line, spread over the grid, issued while the first tile is in flight (tools/conv_bench.py: cold filters cost the
  // K-heavy small-grid layers 5-13 us each; after a whole step of activation traffic they are cold in every step)
  const void* pf = nullptr;
  size_t pf_bytes = 0;
};
void launch_conv(const ConvArgs& a, hipStream_t s);
bool conv_dual_supported(const ConvArgs& a);
bool conv_next_supported(const ConvArgs& a);   // can `a` (shapes for ONE image) carry a fused following 1x1 conv (ConvArgs::next_*)?   // can this build's kernels run `a` with its second input? (the plan builder asks before fusing)
int conv_kpad(int K);                 // padded filter row length the kernels expect
int conv_kpad_split(int K);           // F16X2 filter row length in bf16 elements (K real taps x channels)
bool conv_split_supported(const ConvArgs& a);   // F16X2 input: does the split kernel take this launch?
int conv_npad(int N);
// stem.2 + the 3x3 / stride-2 max-pool in one pass (f16x3 engine): `a` = the conv whose output would be pooled into `pooled`
bool conv_avg_supported(const ConvArgs& a);      // can this launch (shapes for ONE image) carry ConvArgs::avg_y?
bool conv_sx_batch_fits(const ConvArgs& a);      // ... and do the tensors of THIS plan's batch fit the streaming kernel's 2 GiB descriptors?
bool conv_pool_supported(const ConvArgs& a, const Tensor& pooled);
size_t conv_pool_side_bytes(const ConvArgs& a);
void launch_conv_pool(const ConvArgs& a, const Tensor& pooled, void* side, hipStream_t s);
size_t conv_split_slab_bytes(const ConvArgs& a);   // workspace the launch would use for its two-pass split-K (0 = it does not split)

void launch_layernorm(const Tensor& x, const Tensor* res, const float* g, const float* b, const Tensor& y,
                      float eps, hipStream_t s);
// y = a + b (b broadcast over batch when b.n == 1)
void launch_add(const Tensor& a, const Tensor& b, const Tensor& y, hipStream_t s);
void launch_maxpool3x3s2(const Tensor& x, const Tensor& y, hipStream_t s);
void launch_upsample2x(const Tensor& x, const Tensor& y, hipStream_t s);
void launch_avgpool2(const Tensor& x, const Tensor& y, hipStream_t s);
// qk: [B,L,2*D] (q | k), v: [B,L,D] -> o [B,L,D]; softmax(q k^T / sqrt(hd)) v per head
void launch_attention(const Tensor& qk, const Tensor& v, const Tensor& o, int heads, hipStream_t s);
void launch_set_rows(const Tensor& y, const int32_t* rows, int nrows, int rows_per_image, const float* vec,
                     hipStream_t s);
void launch_rowmax(const Tensor& x, float* out, hipStream_t s);
void launch_topk(const float* keys, int B, int N, int K, int32_t* idx, float* vals, hipStream_t s);
void launch_gather_rows(const Tensor& src, const int32_t* idx, int rows_per_image, const Tensor& dst, hipStream_t s);
// dst[b,q,0:4] = src[b, idx[b,q], 0:4] + anchors[idx[b,q]] ; dst rows are 8 floats (4..7 = 0)
void launch_ref_init(const Tensor& boxdelta, const float* anchors, const int32_t* idx, int S, float* ref_unact8,
                     float* ref8, hipStream_t s);
void launch_msdeform(const Tensor& value, int value_coff, const Tensor& offaw, const float* ref8, const Tensor& out,
                     int heads, int hd, int n_levels, int n_points, const int32_t* level_hw_start, float offset_scale,
                     hipStream_t s);
void launch_box_refine(const Tensor& delta, float* ref8, hipStream_t s);
void launch_postprocess_scores(const Tensor& logits, float* scores, hipStream_t s);
bool launch_postprocess_fused(const Tensor& logits, const float* ref8, const float* scale_wh, int B, int Q, float* block6, hipStream_t s);
void launch_postprocess_gather(const float* topv, const int32_t* topi, const float* ref8, const float* scale_wh,
                               int B, int Q, int C, float* block6, hipStream_t s);
// per-call frame arguments travel as KERNEL ARGUMENTS (captured at launch): an async host->device copy of them would read the
// host staging when it executes, i.e. possibly after the next call has overwritten it, and puts two copies on the stream per step
constexpr int RTD_MAX_BATCH = 64;
struct FrameArgs {
  const uint8_t* ptr[RTD_MAX_BATCH];   // HWC uint8 BGR frames on the device
  float scale_wh[2 * RTD_MAX_BATCH];   // (w, h) of each original frame (src/rtdetr_detector.py:234)
  int n;
};
void launch_preprocess_identity(const FrameArgs& fa, int H, int W, const Tensor& y, float* scale_wh_dev, hipStream_t s);
void launch_set_scale(const FrameArgs& fa, float* scale_wh_dev, hipStream_t s);
// fused uint8 stem (bf16 engine): the per-call frame pointers are written to a DEVICE table by a tiny launch outside the
// hipGraph; the captured stem kernel reads its frames through that table (fixed address, per-call contents)
void launch_set_frame_table(const FrameArgs& fa, const uint8_t** table_dev, float* scale_wh_dev, hipStream_t s);
// backbone.stem.0 straight from HWC uint8 BGR frames: BGR->RGB, /255, bf16, 3x3 stride-2 conv (K = 27 padded to 32, two
// MFMA steps), bias + activation; w = the packed [Npad][Kpad] filter with k = tap * 8 + channel (Cin padded 3 -> 8)
void launch_stem0_u8(const uint8_t* const* table_dev, int n, int H, int W, const void* w, int Kpad, const float* bias, const Tensor& y, int act,
                     hipStream_t s);
// device-resident coefficient tables of one (src size -> dst size) PIL resize
struct ResizeCoef {
  const int32_t* hb;  // [dw][2] xmin, count
  const int32_t* hk;  // [dw][hks]
  const int32_t* vb;  // [dh][2]
  const int32_t* vk;  // [dh][vks]
  int hks, vks;
};
void launch_resize_pil(const uint8_t* src, int sh, int sw, uint8_t* tmp, const Tensor& y, int image, const ResizeCoef& coef,
                       hipStream_t s);
void launch_u8_hwc_to_chw_f32(const uint8_t* src, int H, int W, float* out, hipStream_t s);
void launch_resize_pil_u8(const uint8_t* src, int sh, int sw, uint8_t* tmp, uint8_t* dst, int dh, int dw, const ResizeCoef& c, hipStream_t s);
// ---- fused decoder layer (decoder.hip) ----
struct DecLin {
  // DecArgs::split == 0: fragment-major fp32: [tile = n/16][chunk = k/16][lane 0..63][4] = W[16 tile + (lane & 15)][16 chunk + 4 (lane >> 4) + j]
  // DecArgs::split == 1: fragment-major bf16 hi|lo pairs (W = hi + lo to ~2^-17): [tile = n/16][chunk = k/32][hi, lo][lane 0..63][8] =
  //                      W[16 tile + (lane & 15)][32 chunk + 8 (lane >> 4) + j]
  const float* w;
  const float* b;   // [Npad]
  int ldw, N, K;
};
struct DecLN {
  const float* g;
  const float* b;
};

struct DecArgs {
  int mode;                 // 0 = prologue (enc_bbox head + ref init + next projections), 1 = layer, 2 = last layer (+ class head),
                            // 3 = AIFI prologue (x + pos -> q, K/V fragments), 4 = AIFI encoder layer
  int attn_split;           // bf16 / f16x3 engines: self-attention on hi/lo fp16 MFMAs with K / V stored as split fragments (0: exact fp32 MFMAs, A/B + tests)
  int split;                // 1: the linear layers run as 3 fp16 MFMAs on hi/lo splits of both operands (bf16 / f16x3 engines), 0: exact fp32 MFMA
  int B, Q, D, heads, S, n_levels, n_points, ffn, C;
  float offset_scale;
  // per-row state (global, fp32)
  // self-attention inputs of this layer (written by the previous launch): q rows + K / V in MFMA-fragment order
  const float* q_in;        // [B*Q, D]
  const float* kfrag_in;    // [B][heads][tiles][2][64 lanes][4]: K[key 16t+(lane&15)][32h + 16c + 4(lane>>4) + u]
  const float* vfrag_in;    // [B][heads][tiles][2][64 lanes][4]: V[key 16t+4(lane>>4)+u][32h + 16d + (lane&15)]
  const float* hs_in;       // [B*Q, D]
  float* hs_out;            // [B*Q, D]
  const float* qpos_in;     // [B*Q, D]
  float* ref8;              // [B*Q, 8]   sigmoid boxes, refined in place
  float* ref_unact8;        // mode 0: raw enc boxes (debug / parity)
  const float* anchors;     // mode 0: [S,4]
  const int32_t* tk_idx;    // mode 0: [B*Q]
  // cross attention
  const void* value;        // [B, S, value_ld] (+ value_coff), bf16 or fp32
  int value_ld, value_coff, value_f32;
  const int32_t* lvl;       // [n_levels][3] h, w, start
  // outputs for the next layer
  float* qpos_out;          // [B*Q, D]
  float* q_out;             // [B*Q, D]
  float* kfrag_out;         // same layouts as *_in, for the next layer (ping-pong buffers)
  float* vfrag_out;
  float* logits;            // mode 2: [B*Q, C]
  void* out_bf16;           // mode 4: write the output tokens as bf16 here instead of fp32 hs_out (nullptr = fp32)
  float* stamps;            // diagnostic: [blocks][16] phase end times (10 ns units) or nullptr
  int probe;                // diagnostic, timing only (results wrong), rtd_debug_option "dec_stamps" bits 1 / 2: the linear layers skip their MFMAs (filter
                            // stream alone) / their filter loads (arithmetic alone): tools/dec_stamps.py RTD_DEC_PROBE
  // weights
  DecLin o, offaw, op, fc1, fc2, bb0, bb1, bb2, qp0, qp1, qk, v, cls;
  DecLN ln1, ln2, ln3;
};

void launch_dec_layer(const DecArgs& a, hipStream_t s);

// Query selection scores (HF:v2.py:1580-1586): per memory token LayerNorm(enc_output.fc) -> enc_score_head -> max over classes,
// one launch; neither the normalised memory nor the class logits are written (the 300 selected rows are normalised again by
// launch_gather_ln).  x: [rows][256] fp32.
struct SelArgs {
  const float* x;
  int64_t ldx;
  int rows, C, rows_per_image;
  DecLin score;      // fragment-major fp32 (DecArgs::split == 0 layout)
  DecLN ln;
  float* mx;         // [rows]
};
void launch_select_score(const SelArgs& a, hipStream_t s);
// dst[b][q][:] = LayerNorm(x[b][idx[b][q]][:]) (dim 256)
void launch_gather_ln(const float* x, int64_t ldx, int rows_per_image, const int32_t* idx, int B, int Q, const DecLN& ln, float* dst, int64_t ldd,
                      hipStream_t s);

// ---- Stage-2 crop batcher (ops.hip): up to 64 crops per launch, parameters by value ----
struct CropBatch {
  const uint8_t* frame[64];   // HWC uint8 BGR frames on the device
  int fh[64], fw[64];
  int x1[64], y1[64], x2[64], y2[64];   // crop rectangle [y1:y2, x1:x2] inside the frame
};
void launch_crop_resize(const CropBatch& cb, int n, int out_size, const float mean[3], const float stdv[3], float* out, hipStream_t s);

void launch_f32_to(const float* src, void* dst, int dt, int64_t n, hipStream_t s);
void launch_to_f32(const void* src, int dt, float* dst, int64_t n, hipStream_t s);
// dense [rows][C] fp32 <-> F16X2 rows (C % 32 == 0); `ld*` in channels
void launch_f32_to_split(const float* src, int64_t lds, void* dst, int64_t ldd, int64_t rows, int C, hipStream_t s);
void launch_split_to_f32(const void* src, int64_t lds, float* dst, int64_t ldd, int64_t rows, int C, hipStream_t s);
void launch_count_saturated(const void* split_buf, int64_t n16, unsigned long long* count_dev, hipStream_t s);

#if defined(__HIPCC__)
// ---- F16X2 helpers: element offset (in bf16 units, from the tensor base) of the hi half of channel c of a pixel whose first
// channel sits at channel offset `pix_off` (= pixel index * ld); the lo half is SPLIT_GROUP elements further
__device__ __forceinline__ long long split_off(long long pix_off, int c) { return 2 * pix_off + ((c >> 5) << 6) + (c & 31); }
__device__ __forceinline__ void split2(float v, sp16& hi, sp16& lo) {
  // saturate instead of inf (hi = inf would make lo = NaN) - but a NaN stays a NaN (v_med3 would turn it into a finite bound and hide
  // an upstream fault from every isfinite check; the host mirror _capi.to_split keeps it as well)
  v = (v != v) ? v : __builtin_amdgcn_fmed3f(v, -65504.f, 65504.f);
  hi = (sp16)v;
  lo = (sp16)(v - (float)hi);
}
// the three products of a pair x pair contraction step are issued as a*b on these (fp32 accumulate): hi*lo + lo*hi + hi*hi
__device__ __forceinline__ f32x4 mfma_pair16(const sp16x8& a, const sp16x8& b, const f32x4& c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 mfma_pair32(const sp16x8& a, const sp16x8& b, const f32x16& c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
// 8 consecutive channels (c % 8 == 0) of a F16X2 pixel <-> fp32
__device__ __forceinline__ void split_load8(const sp16* base, long long pix_off, int c, float (&v)[8]) {
  const sp16* q = base + split_off(pix_off, c);
  const sp16x8 h = *(const sp16x8*)q, l = *(const sp16x8*)(q + SPLIT_GROUP);
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = (float)h[k] + (float)l[k];
}
__device__ __forceinline__ void split_store8(sp16* base, long long pix_off, int c, const float (&v)[8]) {
  sp16* q = base + split_off(pix_off, c);
  sp16x8 h, l;
#pragma unroll
  for (int k = 0; k < 8; ++k) { sp16 a, b; split2(v[k], a, b); h[k] = a; l[k] = b; }
  *(sp16x8*)q = h;
  *(sp16x8*)(q + SPLIT_GROUP) = l;
}
// Reductions over the four lanes {l, l ^ 16, l ^ 32, l ^ 48} (the 16-lane rows of a wave: an MFMA 16x16 accumulator's row groups)
// on gfx950's row-swap VALU ops instead of two ds_bpermute round trips each: v_permlane16_swap exchanges odd rows of its first
// operand with even rows of its second, v_permlane32_swap the upper half of the first with the lower half of the second; fed the
// same value twice, the two results are {this row pair's even row, its odd row} / {lower half, upper half} in every lane.  max and
// fp add are commutative, so every lane gets the bits the xor-shuffle version produced.
__device__ __forceinline__ float rows4_max(float v) {
  const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  const float t = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(t), __float_as_uint(t), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float rows4_sum(float v) {
  const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  const float t = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(t), __float_as_uint(t), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
// Wave-wide sum / max without LDS traffic: `__shfl_xor` compiles to ds_bpermute_b32 (an LDS-crossbar round trip, ~100+ cycles,
// six in a dependent chain per reduction); here four DPP steps reduce each 16-lane row in the VALU and four v_readlane combine
// the rows.  Every lane receives the result.  (Summation order differs from a butterfly - callers are not bit-pinned to one.)
#define RTD_DPP_ROW_REDUCE(OP)                                                           \
  v = OP(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false)));  /* quad_perm [1,0,3,2] */ \
  v = OP(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false)));  /* quad_perm [2,3,0,1] */ \
  v = OP(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false))); /* row_ror:4 */ \
  v = OP(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false))); /* row_ror:8 */
__device__ __forceinline__ float rtd_addf(float a, float b) { return a + b; }
__device__ __forceinline__ float wave_sum64(float v) {
  RTD_DPP_ROW_REDUCE(rtd_addf)
  const int iv = __builtin_bit_cast(int, v);
  return (__builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16))) +
         (__builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48)));
}
__device__ __forceinline__ float wave_max64(float v) {
  RTD_DPP_ROW_REDUCE(fmaxf)
  const int iv = __builtin_bit_cast(int, v);
  return fmaxf(fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0)), __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16))),
               fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32)), __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48))));
}
// sum / max over each 16-lane row only (the result is in every lane of the row)
__device__ __forceinline__ float row16_max(float v) {
  RTD_DPP_ROW_REDUCE(fmaxf)
  return v;
}
#endif

}  // namespace rtd
