// engine_internal.h - what engine.hip (the product C ABI) and testapi.hip (kernel-level test / bench / debug entry points) share:
// the handle, its plans and the few host helpers both translation units call.  Not installed, not part of the ABI.
#pragma once
#include <math.h>
#include <string.h>

#include <chrono>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/rtdetr_mi355.h"
#include "../../include/rtdetr_mi355_test.h"
#include "common.h"

namespace rtd_eng {
using namespace rtd;

// Plan-build switches (rtd_debug_option): process-wide defaults that every handle SNAPSHOTS at rtd_create (rtd_engine::opts), so all
// plans of a handle (one per batch size, built lazily) agree with each other and a later rtd_debug_option call - e.g. by another
// test - cannot change a live handle.
struct PlanOpts {
  int dec_stamps = 0;   // record per-phase stamps of decoder layer 2 into debug tensor "dec_stamps"
  int dec_split = 1;    // bf16 / f16x3 engines run the fused decoder / AIFI linears as bf16 hi/lo splits (0: fp32 MFMA, 2: bf16 filters)
  int sc_fold = 1;      // fold a block's projection shortcut into its last conv (ConvArgs::x2)
  int c1_fuse = 1;      // bf16 plans run a stage-0 block's reduce conv inside the previous block's last conv
  // self-attention on hi/lo fp16 MFMAs - bit 0 the fused AIFI layer, bit 1 decoder.  Both since round 5: with round 2's bf16 pairs AIFI's
  // softmax arguments (tens) turned a 2^-16 product error into 1.7e-4 on the layer output and the layer stayed on fp32 MFMAs; with fp16
  // pairs (2^-22) every x3_check case keeps its rows and its worst errors (R50 640 bs 8: 1.4e-6 / 4.9e-4 px either way), and the layer
  // takes 52 instead of 61 us.  The un-fused AIFI attention of encoders wider than 256 channels (k_attention_mfma_f32) is fp32 as before.
  int attn_split = 3;
  int up_fold = 1;      // read the FPN's upsampled lateral straight from the half-size tensor (ConvArgs::x_up2)
  int arena_reuse = 1;  // backbone stages recycle their activation buffers
  int stem_fused_split = 1;   // f16x3 engine: backbone.stem.0 straight from the uint8 frames (hi/lo pairs made on the fly from the bytes)
  int aifi_pair = 1;          // f16x3 engine, un-fused AIFI (encoders wider than 256 channels): its linears on the pair kernels instead of fp32 MFMAs
  int post_fused = 1;         // sigmoid + top-k + box decode of the post-processor in one launch (ops.hip TopkPost)
  int dead_out = 1;           // f16x3 engine: a stage output whose only readers are fused into the launch that produces it is not written (ConvArgs::y_dead)
  int avg_fuse = 1;           // f16x3 engine: a stage's last expand conv also writes the 2 x 2 average the next stage's vd shortcut reads (ConvArgs::avg_y)
  int stem_pool_fuse = 1;     // f16x3 engine: backbone.stem.2 and the 3x3 / stride-2 max-pool in one pass (the conv rows are never written)
  int sel_fused = 1;    // LayerNorm + score head + class max of the query selection in one launch
  int dec_fused = 1;    // 0 = one launch per decoder op
  int side_stream = 7;  // bit 0: the query-selection chain runs on a second stream beside the value projection; bit 1: the decoder input
                        // projections of the two larger levels run there beside the PAN path; bit 2: the encoder input projections of the two
                        // larger levels (they only need the stage-1 / stage-2 maps) run there beside stages 2 / 3 and AIFI, whose 40^2 / 20^2
                        // grids and row kernels leave CUs idle (0: one stream)
};

struct HostTensor {
  const float* data = nullptr;
  std::vector<int64_t> shape;
  int64_t numel() const {
    int64_t n = 1;
    for (auto d : shape) n *= d;
    return n;
  }
};

struct DevWeight {
  void* w = nullptr;
  float* bias = nullptr;
  int N = 0, K = 0, Kpad = 0, Npad = 0, dt = F32;
};

struct Op {
  std::string name;
  const char* kernel;
  double flops, bytes;
  std::function<void(hipStream_t)> run;
  bool debug_only = false;   // runs (and becomes a graph node) only on handles that asked for it (rtd_debug_force_topk)
  // lane 1 = the engine's side stream: independent work that runs BESIDE the main stream's (the query-selection chain - enc_output,
  // scoring, top-k, gather: narrow grids, 150 us - next to the value projection of all decoder layers: 210 us).  kind: 0 launch,
  // 1 fork (side waits for everything enqueued on main so far), 2 join (main waits for the side stream)
  int lane = 0, kind = 0;
};

struct Plan {
  int n = 0;
  void* arena = nullptr;
  size_t arena_bytes = 0;
  std::vector<Op> ops;
  std::map<std::string, Tensor> named;
  Tensor input;            // [n,H,W,8]
  const uint8_t** frame_table = nullptr;   // fused uint8 stem: device table of the n frame pointers of the current call
  bool stem_fused = false;
  std::vector<std::pair<void*, size_t>> split_acts;   // every F16X2 activation the builder allocated (pointer, bytes): rtd_self_check's saturation scan
  float* block6 = nullptr; // [n,Q,6]
  float* scale_wh = nullptr;
  int32_t* tk_idx = nullptr;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
};

struct ResizeTables {
  ResizeCoef coef;
  std::vector<void*> dev;
};


extern PlanOpts g_opts;
extern thread_local std::string g_create_error;   // last error of a call that has no handle (rtd_create, rtd_op_*, rtd_bench_*)

}  // namespace rtd_eng

struct rtd_engine {
  typedef rtd::ConvOpts ConvOpts; typedef rtd::FrameArgs FrameArgs;
  static constexpr int BF16 = rtd::BF16;
  typedef rtd_eng::PlanOpts PlanOpts; typedef rtd_eng::HostTensor HostTensor; typedef rtd_eng::DevWeight DevWeight;
  typedef rtd_eng::Plan Plan; typedef rtd_eng::ResizeTables ResizeTables;
  rtd_config cfg;
  PlanOpts opts;           // snapshot of g_opts at rtd_create
  ConvOpts conv_opts;      // snapshot of the conv dispatch switches at rtd_create (every launch of this handle's plans points here)
  bool force_used = false; // rtd_debug_force_topk was called on this handle: plans include the (debug-only) index override launch
  std::mutex mu;
  std::string err;
  hipStream_t stream = nullptr;
  hipStream_t side = nullptr;                      // Op::lane 1
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;  // eager passes only (a graph holds these edges as node dependencies)
  hipEvent_t ev_xs = nullptr;                      // rtd_wait_stream / rtd_signal_stream: this handle's own event, never handed out
  bool loaded = false;
  // counters (rtd_get_stats): what this handle did, so that a failure report describes itself
  int64_t st_plans = 0, st_graphs = 0, st_graph_nodes = 0, st_graph_launches = 0, st_eager = 0, st_submits = 0, st_collects = 0, st_failed = 0;
  int32_t st_last_code = 0;
  // real-weights guard: the largest |folded filter value| of the blob (rtd_load_weights) and the saturated activations the last rtd_self_check saw
  float max_abs_filter = 0.f; std::string max_abs_filter_name; int64_t st_saturated = -1;
  bool in_flight = false;                          // rtd_infer_async enqueued a batch that rtd_collect / rtd_sync has not waited for yet
  uint8_t* pin_stage = nullptr; size_t pin_stage_bytes = 0;    // rtd_infer_async on host frames: pinned staging (one batch in flight per handle)
  int P = BF16;  // storage / MFMA type of the conv trunk
  std::vector<char> blob;
  std::map<std::string, HostTensor> host;
  std::map<std::string, DevWeight> wcache;
  std::map<std::string, float*> vcache;
  std::vector<void*> allocs;
  std::map<int, std::unique_ptr<Plan>> plans;
  std::map<std::pair<int, int>, ResizeTables> resize;
  // per-call staging
  uint8_t* frame_stage = nullptr; size_t frame_stage_bytes = 0;
  uint8_t* resize_tmp = nullptr; size_t resize_tmp_bytes = 0;
  uint8_t* u8_stage = nullptr;             // fused uint8 stem: resized frames, max_batch x H x W x 3
  FrameArgs last_fa;                       // frame table of the last call (rtd_debug_tensor("input") re-runs the preprocess from it)
  float* block_host = nullptr;
  int32_t* forced_idx = nullptr; int32_t* force_flag = nullptr;
  // geometry
  int lvl_h[3], lvl_w[3], lvl_start[3], S = 0;
  float* anchors_dev = nullptr; int32_t* invalid_rows_dev = nullptr; int n_invalid = 0;
  int32_t* lvl_dev = nullptr;
  float* pos_dev = nullptr;
  int last_n = 0;

  void* dmalloc(size_t bytes) {
    void* p = nullptr;
    HIP_CHECK(hipMalloc(&p, bytes ? bytes : 16));
    allocs.push_back(p);
    return p;
  }
};

namespace rtd_eng {

// engine.hip
void check_n(rtd_engine* e, int n);
Plan* get_plan(rtd_engine* e, int n);
void point_at_blank_frames(rtd_engine* h, Plan* p, int n);
void pil_coeffs(int in_size, int out_size, std::vector<int32_t>& bounds, std::vector<int32_t>& kk, int& ksize);

template <typename F>
int guarded(rtd_engine* e, F&& f) {
  if (!e) return RTD_E_INVALID;
  std::lock_guard<std::mutex> lk(e->mu);
  try {
    f();
    return RTD_OK;
  } catch (const Error& er) {
    e->err = er.what();
    e->st_failed++; e->st_last_code = er.code;
    return er.code;
  } catch (const std::bad_alloc&) {
    e->err = "host allocation failed";
    e->st_failed++; e->st_last_code = RTD_E_OOM;
    return RTD_E_OOM;
  } catch (const std::exception& ex) {
    e->err = ex.what();
    e->st_failed++; e->st_last_code = RTD_E_HIP;
    return RTD_E_HIP;
  }
}

template <typename F>
inline int op_guard(F&& f) {
  try {
    f();
    HIP_CHECK(hipDeviceSynchronize());
    return RTD_OK;
  } catch (const Error& er) {
    g_create_error = er.what();
    return er.code;
  } catch (const std::exception& ex) {
    g_create_error = ex.what();
    return RTD_E_HIP;
  }
}
inline Tensor mk(const void* p, int dt, int n, int h, int w, int c) {
  Tensor t;
  t.p = (void*)p; t.dt = dt; t.n = n; t.h = h; t.w = w; t.c = c; t.ld = c; t.bstride = (int64_t)h * w * c;
  return t;
}

}  // namespace rtd_eng
