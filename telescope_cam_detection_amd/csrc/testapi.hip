// testapi.hip - kernel-level test, bench and debug entry points of libmi355rtdetr.so (include/rtdetr_mi355_test.h).
// Nothing here is on the detection path: tests/ , tools/ and bench.py's diagnostic legs are the only callers.
#include "engine_internal.h"

using namespace rtd;
using namespace rtd_eng;

extern "C" {

int rtd_debug_tensor(rtd_handle h, const char* name, float* out, int64_t capacity, int64_t shape[4]) {
  return guarded(h, [&] {
    RTD_CHECK(name && shape && h->last_n > 0, RTD_E_STATE, "no forward has run");
    Plan* p = h->plans[h->last_n].get();
    auto it = p->named.find(name);
    RTD_CHECK(it != p->named.end(), RTD_E_INVALID, std::string("unknown debug tensor ") + name);
    const Tensor& t = it->second;
    if (p->stem_fused && strcmp(name, "input") == 0 && out) {      // the fused stem never wrote it: run the stand-alone preprocess now
      HIP_CHECK(hipSetDevice(h->cfg.device));
      launch_preprocess_identity(h->last_fa, h->cfg.input_h, h->cfg.input_w, p->input, p->scale_wh, h->stream);
    }
    shape[0] = t.n; shape[1] = t.h; shape[2] = t.w; shape[3] = t.c;
    const int64_t numel = t.pixels() * t.c;
    if (!out) return;
    RTD_CHECK(capacity >= numel, RTD_E_INVALID, "debug tensor: output capacity too small");
    RTD_CHECK(t.bstride == (int64_t)t.h * t.w * t.ld, RTD_E_INVALID, "debug tensor: non-dense batch stride");
    HIP_CHECK(hipSetDevice(h->cfg.device));
    const size_t es = dtype_size(t.dt);
    if (t.dt == I32) {                                             // index tensors: exact as fp32 (token ids < 2^24)
      std::vector<int32_t> tmp((size_t)numel);
      HIP_CHECK(hipStreamSynchronize(h->stream));
      HIP_CHECK(hipMemcpy(tmp.data(), t.p, (size_t)numel * 4, hipMemcpyDeviceToHost));
      for (int64_t i = 0; i < numel; ++i) out[i] = (float)tmp[(size_t)i];
      return;
    }
    if (t.dt == F16X2) RTD_CHECK(t.c % SPLIT_GROUP == 0, RTD_E_INVALID, "debug tensor: split tensor with a partial channel group");
    void* dense = nullptr;
    float* f32 = nullptr;
    HIP_CHECK(hipMalloc(&dense, (size_t)numel * es));
    hipError_t er = hipMalloc((void**)&f32, (size_t)numel * 4);
    if (er == hipSuccess) er = hipMemcpy2DAsync(dense, (size_t)t.c * es, t.p, (size_t)t.ld * es, (size_t)t.c * es, (size_t)t.pixels(), hipMemcpyDeviceToDevice, h->stream);
    if (er == hipSuccess) {
      if (t.dt == F16X2) launch_split_to_f32(dense, t.c, f32, t.c, t.pixels(), t.c, h->stream);
      else launch_to_f32(dense, t.dt, f32, numel, h->stream);
      er = hipMemcpyAsync(out, f32, (size_t)numel * 4, hipMemcpyDeviceToHost, h->stream);
    }
    if (er == hipSuccess) er = hipStreamSynchronize(h->stream);
    (void)hipFree(dense);
    if (f32) (void)hipFree(f32);
    HIP_CHECK(er);
  });
}

int rtd_debug_force_topk(rtd_handle h, const int32_t* idx, int32_t n) {
  return guarded(h, [&] {
    RTD_CHECK(h->loaded, RTD_E_STATE, "weights not loaded");
    HIP_CHECK(hipSetDevice(h->cfg.device));
    if (idx && !h->force_used) {
      // first use on this handle: the override launch joins the plans; graphs built without it are rebuilt on their next run
      h->force_used = true;
      HIP_CHECK(hipStreamSynchronize(h->stream));
      for (auto& kv : h->plans) {
        if (kv.second->exec) { (void)hipGraphExecDestroy(kv.second->exec); kv.second->exec = nullptr; }
        if (kv.second->graph) { (void)hipGraphDestroy(kv.second->graph); kv.second->graph = nullptr; }
      }
    }
    int32_t flag = 0;
    if (idx) {
      RTD_CHECK(n >= 1 && n <= h->cfg.max_batch, RTD_E_INVALID, "batch size");
      for (int64_t i = 0; i < (int64_t)n * h->cfg.num_queries; ++i)
        RTD_CHECK(idx[i] >= 0 && idx[i] < h->S, RTD_E_INVALID, "forced token index out of range");
      HIP_CHECK(hipMemcpyAsync(h->forced_idx, idx, (size_t)n * h->cfg.num_queries * 4, hipMemcpyHostToDevice, h->stream));
      flag = 1;
    }
    HIP_CHECK(hipMemcpyAsync(h->force_flag, &flag, 4, hipMemcpyHostToDevice, h->stream));
    HIP_CHECK(hipStreamSynchronize(h->stream));
  });
}

static int g_profile_twice = 0;
int rtd_profile(rtd_handle h, int32_t n, int32_t reps, rtd_layer_time* out, int32_t capacity, int32_t* count) {
  return guarded(h, [&] {
    check_n(h, n);
    RTD_CHECK(count && reps >= 1, RTD_E_INVALID, "arguments");
    HIP_CHECK(hipSetDevice(h->cfg.device));
    Plan* p = get_plan(h, n);
    std::vector<Op*> ops;                                     // what a forward of this handle launches
    for (auto& op : p->ops)
      if (op.kind == 0 && (!op.debug_only || h->force_used)) ops.push_back(&op);     // every launch, in plan order, on the main stream (no overlap while timing)
    const int nops = (int)ops.size();
    *count = nops;
    if (!out) return;
    // the fused stem reads its frames through the device table: without a preceding forward of this batch size point it at
    // zero-filled staging frames (timings do not depend on pixel values)
    if (p->stem_fused && (h->last_n != n || h->last_fa.n != n)) point_at_blank_frames(h, p, n);
    RTD_CHECK(capacity >= nops, RTD_E_INVALID, "profile: capacity too small");
    std::vector<hipEvent_t> ev((size_t)nops + 1);
    for (auto& x : ev) HIP_CHECK(hipEventCreate(&x));
    std::vector<double> acc(nops, 0.0);
    for (Op* op : ops) op->run(h->stream);   // warm-up
    for (int r = 0; r < reps; ++r) {
      if (g_profile_twice) {
        // diagnostic: every op runs twice back to back and only the SECOND run is timed (operands, filter and TLB entries
        // warm from the first) - the gap to the normal profile is what the op pays for cold operands inside the network
        std::vector<hipEvent_t> ev2((size_t)nops);
        for (auto& x : ev2) HIP_CHECK(hipEventCreate(&x));
        for (int i = 0; i < nops; ++i) {
          ops[i]->run(h->stream);
          HIP_CHECK(hipEventRecord(ev2[i], h->stream));
          ops[i]->run(h->stream);
          HIP_CHECK(hipEventRecord(ev[i + 1], h->stream));
        }
        HIP_CHECK(hipStreamSynchronize(h->stream));
        for (int i = 0; i < nops; ++i) {
          float ms = 0.f;
          HIP_CHECK(hipEventElapsedTime(&ms, ev2[i], ev[i + 1]));
          acc[i] += ms;
        }
        for (auto& x : ev2) (void)hipEventDestroy(x);
        continue;
      }
      HIP_CHECK(hipEventRecord(ev[0], h->stream));
      for (int i = 0; i < nops; ++i) {
        ops[i]->run(h->stream);
        HIP_CHECK(hipEventRecord(ev[i + 1], h->stream));
      }
      HIP_CHECK(hipStreamSynchronize(h->stream));
      for (int i = 0; i < nops; ++i) {
        float ms = 0.f;
        HIP_CHECK(hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
        acc[i] += ms;
      }
    }
    for (auto& x : ev) (void)hipEventDestroy(x);
    for (int i = 0; i < nops; ++i) {
      rtd_layer_time& t = out[i];
      memset(&t, 0, sizeof t);
      strncpy(t.name, ops[i]->name.c_str(), sizeof(t.name) - 1);
      strncpy(t.kernel, ops[i]->kernel, sizeof(t.kernel) - 1);
      t.ms = (float)(acc[i] / reps);
      t.flops = ops[i]->flops;
      t.bytes = ops[i]->bytes;
    }
  });
}

// ---- kernel-level test entry points ---------------------------------------------------------------
// reads one dword per 128-byte line (bench only: does a READ bring lines into the Infinity Cache?)
__global__ void k_touch_read(const unsigned* p, size_t lines, unsigned* sink) {
  unsigned acc = 0;
  for (size_t l = (size_t)blockIdx.x * blockDim.x + threadIdx.x; l < lines; l += (size_t)gridDim.x * blockDim.x) acc += p[l * 32];
  if (acc == 0x12345u) *sink = acc;
}
// bench only: finite fp16 values of mixed sign, exponents 2^-5 .. 2^2, random mantissas (matrix-core power depends on the operand bits:
// zero-filled operands let the chip hold a clock that real activations do not)
__global__ void k_fill_rand16(uint16_t* p, size_t n, unsigned seed) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t step = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) {
    unsigned h = (unsigned)i * 2654435761u + seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = (uint16_t)((h & 0x8000u) | ((10u + ((h >> 16) & 7u)) << 10) | (h & 0x3ffu));
  }
}
// rtd_bench_mfma_rate: 16 independent accumulators per wave, 4 waves per CU (one per SIMD), operands in registers
__global__ __launch_bounds__(256) void k_mfma_rate(const unsigned* __restrict__ seed, int iters, long long* __restrict__ out, float* __restrict__ sink) {
  const int tid = threadIdx.x, lane = tid & 63;
  sp16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 8; ++j) {
      const unsigned h = seed[(lane * 8 + j + i * 512) & 4095];
      a[i][j] = __builtin_bit_cast(sp16, (unsigned short)h);
      b[i][j] = __builtin_bit_cast(sp16, (unsigned short)(h >> 16));
    }
  f32x4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  // in-place accumulators pinned by inline asm: compiled from the builtin inside this translation unit, hipcc rotated the accumulators
  // through AGPR copies on the loop's back edge (27 instead of 17 cycles per MFMA)
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i & 3]), "v"(b[(i >> 2) & 3]));
  }
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");      // the last MFMAs' results are written before anything reads them
  __builtin_amdgcn_s_waitcnt(0);
  const long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
  if (s == 123.456f) sink[0] = s;
  if (lane == 0) { const int w = blockIdx.x * 4 + (tid >> 6); out[2 * w] = c1 - c0; out[2 * w + 1] = r1 - r0; }
}
static int g_bench_rewarm = 0;   // "bench_rewarm": rtd_bench_conv rewrites 1 = activations, 2 = weights after its flush (back into the Infinity Cache)
int rtd_debug_option(const char* name, int value) {
  if (!name) return RTD_E_INVALID;
  if (strcmp(name, "reset") == 0) {                             // every switch back to its default (tests call this after each case)
    g_opts = PlanOpts();
    g_profile_twice = 0; g_bench_rewarm = 0;
    conv_opts_template() = ConvOpts();
    return RTD_OK;
  }
  // plan-build and conv-dispatch switches: process-wide TEMPLATES that rtd_create snapshots into the handle - a call here changes handles
  // created afterwards (and the kernel-level rtd_op_* / rtd_bench_* entry points), never a live handle
  const struct { const char* n; int* p; } plan_opts[] = {
      {"dec_stamps", &g_opts.dec_stamps}, {"dec_fused", &g_opts.dec_fused}, {"side_stream", &g_opts.side_stream}, {"sel_fused", &g_opts.sel_fused},
      {"stem_fused_split", &g_opts.stem_fused_split}, {"stem_pool_fuse", &g_opts.stem_pool_fuse}, {"avg_fuse", &g_opts.avg_fuse}, {"dead_out", &g_opts.dead_out}, {"post_fused", &g_opts.post_fused}, {"aifi_pair", &g_opts.aifi_pair}, {"sc_fold", &g_opts.sc_fold}, {"arena_reuse", &g_opts.arena_reuse},
      {"up_fold", &g_opts.up_fold}, {"attn_split", &g_opts.attn_split}, {"c1_fuse", &g_opts.c1_fuse}, {"dec_split", &g_opts.dec_split},
      {"profile_twice", &g_profile_twice}, {"bench_rewarm", &g_bench_rewarm},
  };
  for (const auto& t : plan_opts)
    if (strcmp(name, t.n) == 0) { *t.p = value; return RTD_OK; }
  if (conv_set_option(name, value)) return RTD_OK;
  return RTD_E_INVALID;
}

static int op_conv_impl(int dtype, const void* x, const void* x2, int C2, const void* w_ohwi_f32, const float* bias, const void* res, void* y,
                        int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int act, int res_mode, int out_f32, int x_up2 = 0,
                        const void* w1_f32 = nullptr, const float* bias1 = nullptr, void* y1 = nullptr, int Cnext = 0, int next_act = 0) {
  return op_guard([&] {
    RTD_CHECK(KH == KW, RTD_E_INVALID, "square filters only");
    const int K = KH * KW * Cin + (x2 ? C2 : 0), Npad = conv_npad(Cout);
    const int Kpad = dtype == F16X2 ? conv_kpad_split(K) : conv_kpad(K);
    const int kcols = dtype == F16X2 ? Kpad / 2 : Kpad;          // fp32 staging row (F16X2: 2 bf16 per column)
    const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
    float* wpad = nullptr; void* wdev = nullptr; float* bpad = nullptr;
    HIP_CHECK(hipMalloc((void**)&wpad, (size_t)Npad * kcols * 4));
    HIP_CHECK(hipMemset(wpad, 0, (size_t)Npad * kcols * 4));
    HIP_CHECK(hipMemcpy2D(wpad, (size_t)kcols * 4, w_ohwi_f32, (size_t)K * 4, (size_t)K * 4, Cout, hipMemcpyDeviceToDevice));
    HIP_CHECK(hipMalloc((void**)&bpad, (size_t)Npad * 4));
    HIP_CHECK(hipMemset(bpad, 0, (size_t)Npad * 4));
    HIP_CHECK(hipMemcpy(bpad, bias, (size_t)Cout * 4, hipMemcpyDeviceToDevice));
    if (dtype == BF16) {
      HIP_CHECK(hipMalloc(&wdev, (size_t)Npad * Kpad * 2));
      launch_f32_to(wpad, wdev, BF16, (int64_t)Npad * Kpad, nullptr);
    } else if (dtype == F16X2) {
      HIP_CHECK(hipMalloc(&wdev, (size_t)Npad * Kpad * 2));
      launch_f32_to_split(wpad, kcols, wdev, kcols, Npad, kcols, nullptr);
    } else wdev = wpad;
    ConvArgs a;
    a.x = x_up2 ? mk(x, dtype, B, H / 2, W / 2, Cin) : mk(x, dtype, B, H, W, Cin);      // x_up2: H, W are the OUTPUT extents
    a.x_up2 = x_up2;
    a.y = mk(y, out_f32 ? F32 : dtype, B, OH, OW, Cout);
    a.w = wdev; a.bias = bpad; a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad; a.Kpad = Kpad; a.Npad = Npad;
    a.act = act; a.res_mode = res ? res_mode : RES_NONE;
    if (res) a.res = mk(res, dtype, B, OH, OW, Cout);
    if (x2) a.x2 = mk(x2, dtype, B, OH, OW, C2);
    float* w1pad = nullptr; void* w1dev = nullptr; float* b1pad = nullptr;
    if (y1) {                                                    // a following 1x1 conv Cout -> Cnext fused into this launch (ConvArgs::next_*)
      RTD_CHECK(dtype == BF16 || dtype == F16X2, RTD_E_INVALID, "fused following conv: bf16 / f16x2 only");
      const int N1 = conv_npad(Cnext), K1 = dtype == F16X2 ? conv_kpad_split(Cout) : conv_kpad(Cout), k1cols = dtype == F16X2 ? K1 / 2 : K1;
      HIP_CHECK(hipMalloc((void**)&w1pad, (size_t)N1 * k1cols * 4));
      HIP_CHECK(hipMemset(w1pad, 0, (size_t)N1 * k1cols * 4));
      HIP_CHECK(hipMemcpy2D(w1pad, (size_t)k1cols * 4, w1_f32, (size_t)Cout * 4, (size_t)Cout * 4, Cnext, hipMemcpyDeviceToDevice));
      HIP_CHECK(hipMalloc((void**)&b1pad, (size_t)N1 * 4));
      HIP_CHECK(hipMemset(b1pad, 0, (size_t)N1 * 4));
      HIP_CHECK(hipMemcpy(b1pad, bias1, (size_t)Cnext * 4, hipMemcpyDeviceToDevice));
      HIP_CHECK(hipMalloc(&w1dev, (size_t)N1 * K1 * 2));
      if (dtype == BF16) launch_f32_to(w1pad, w1dev, BF16, (int64_t)N1 * K1, nullptr);
      else launch_f32_to_split(w1pad, k1cols, w1dev, k1cols, N1, k1cols, nullptr);
      a.next_w = w1dev; a.next_bias = b1pad; a.next_y = mk(y1, dtype, B, OH, OW, Cnext); a.next_kpad = K1; a.next_act = next_act;
      RTD_CHECK(conv_next_supported(a), RTD_E_INVALID, "fused following conv: shape not taken by the streaming kernels");
    }
    ConvWorkspace ws;
    ws.slab_bytes = conv_split_slab_bytes(a);
    if (ws.slab_bytes) HIP_CHECK(hipMalloc((void**)&ws.slab, ws.slab_bytes));
    a.ws = ws;
    launch_conv(a, nullptr);
    HIP_CHECK(hipDeviceSynchronize());
    if (ws.slab) (void)hipFree(ws.slab);
    HIP_CHECK(hipDeviceSynchronize());
    if (wdev != wpad) (void)hipFree(wdev);
    (void)hipFree(wpad); (void)hipFree(bpad);
    if (w1pad) { (void)hipFree(w1pad); (void)hipFree(w1dev); (void)hipFree(b1pad); }
  });
}

int rtd_op_conv(int dtype, const void* x, const void* w_ohwi_f32, const float* bias, const void* res, void* y, int B, int H,
                int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int act, int res_mode, int out_f32) {
  return op_conv_impl(dtype, x, nullptr, 0, w_ohwi_f32, bias, res, y, B, H, W, Cin, Cout, KH, KW, stride, pad, act, res_mode, out_f32);
}

int rtd_op_conv_dual(int dtype, const void* x, const void* x2, const void* w_f32, const float* bias, const void* res, void* y, int B,
                     int H, int W, int Cin, int C2, int Cout, int KH, int stride, int pad, int act, int res_mode, int out_f32, int x_up2) {
  if (!x2 || (x_up2 && ((H | W) & 1))) return RTD_E_INVALID;
  return op_conv_impl(dtype, x, x2, C2, w_f32, bias, res, y, B, H, W, Cin, Cout, KH, KH, stride, pad, act, res_mode, out_f32, x_up2);
}

int rtd_op_conv_next(int dtype, const void* x, const void* x2, const void* w_f32, const float* bias, const void* res, void* y, const void* w1_f32,
                     const float* bias1, void* y1, int B, int H, int W, int Cin, int C2, int Cout, int Cnext, int act, int res_mode, int next_act) {
  if (!y1 || !w1_f32 || !bias1) return RTD_E_INVALID;
  return op_conv_impl(dtype, x, x2, x2 ? C2 : 0, w_f32, bias, res, y, B, H, W, Cin, Cout, 1, 1, 1, 0, act, res_mode, 0, 0, w1_f32, bias1, y1, Cnext, next_act);
}

int rtd_bench_conv(int dtype, int B, int H, int W, int Cin, int Cout, int KH, int stride, int pad, int with_res, int reps,
                   int flush_mb, float* us_out) {
  return op_guard([&] {
    const int K = KH * KH * Cin, Kpad = dtype == F16X2 ? conv_kpad_split(K) : conv_kpad(K), Npad = conv_npad(Cout);
    const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KH) / stride + 1;
    const size_t es = dtype == BF16 ? 2 : 4;
    const size_t xb = (size_t)B * H * W * Cin * es, yb = (size_t)B * OH * OW * Cout * es, wb = (size_t)Npad * Kpad * (dtype == F32 ? 4 : 2);
    void *x = nullptr, *y = nullptr, *r = nullptr, *w = nullptr, *flush = nullptr; float* bias = nullptr;
    HIP_CHECK(hipMalloc(&x, xb)); HIP_CHECK(hipMalloc(&y, yb)); HIP_CHECK(hipMalloc(&w, wb)); HIP_CHECK(hipMalloc((void**)&bias, Npad * 4));
    HIP_CHECK(hipMemset(x, 0, xb)); HIP_CHECK(hipMemset(w, 0, wb)); HIP_CHECK(hipMemset(bias, 0, Npad * 4));
    if (with_res) { HIP_CHECK(hipMalloc(&r, yb)); HIP_CHECK(hipMemset(r, 0, yb)); }
    if ((g_bench_rewarm & 32) && dtype != F32) {                 // "bench_rewarm" bit 5: random 16-bit operands instead of zeros
      rtd_launch(k_fill_rand16, dim3(1024), dim3(256), 0, nullptr, (uint16_t*)x, xb / 2, 1u);
      rtd_launch(k_fill_rand16, dim3(1024), dim3(256), 0, nullptr, (uint16_t*)w, wb / 2, 2u);
      if (r) rtd_launch(k_fill_rand16, dim3(1024), dim3(256), 0, nullptr, (uint16_t*)r, yb / 2, 3u);
      HIP_CHECK(hipDeviceSynchronize());
    }
    if (flush_mb > 0) HIP_CHECK(hipMalloc(&flush, (size_t)flush_mb << 20));
    ConvArgs a;
    a.x = mk(x, dtype, B, H, W, Cin);
    a.y = mk(y, dtype, B, OH, OW, Cout);
    a.w = w; a.bias = bias; a.KH = KH; a.KW = KH; a.stride = stride; a.pad = pad; a.Kpad = Kpad; a.Npad = Npad;
    a.act = 1; a.res_mode = with_res ? RES_PRE : RES_NONE;
    if (with_res) a.res = mk(r, dtype, B, OH, OW, Cout);
    a.ws.slab_bytes = std::max<size_t>((size_t)4096 * 8 * 8, conv_split_slab_bytes(a));   // block stamps (glds_drop 32) / two-pass split-K
    HIP_CHECK(hipMalloc((void**)&a.ws.slab, a.ws.slab_bytes));
    HIP_CHECK(hipMemset(a.ws.slab, 0, a.ws.slab_bytes));
    hipEvent_t e0, e1;
    HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch_conv(a, nullptr);
    HIP_CHECK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < reps; ++i) launch_conv(a, nullptr);
    HIP_CHECK(hipEventRecord(e1, nullptr));
    HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    us_out[0] = ms * 1e3f / reps;
    us_out[1] = 0.f;
    if (flush) {
      float tot = 0.f;
      for (int i = 0; i < reps; ++i) {
        HIP_CHECK(hipMemsetAsync(flush, i & 0xff, (size_t)flush_mb << 20, nullptr));
        if (g_bench_rewarm & 1) { HIP_CHECK(hipMemsetAsync(x, 0, xb, nullptr)); if (r) HIP_CHECK(hipMemsetAsync(r, 0, yb, nullptr)); }
        if (g_bench_rewarm & 2) HIP_CHECK(hipMemsetAsync(w, 0, wb, nullptr));
        if (g_bench_rewarm & 4) rtd_launch(k_touch_read, dim3(64), dim3(256), 0, nullptr, (const unsigned*)w, wb / 128, (unsigned*)bias);
        if (g_bench_rewarm & 16) {      // in-kernel prefetch path: a small unrelated conv launch carries pf = this filter
          ConvArgs d = a;
          d.x = mk(x, dtype, 1, H, W, Cin); d.y = mk(y, dtype, 1, OH, OW, Cout);
          if (with_res) d.res = mk(r, dtype, 1, OH, OW, Cout);
          d.pf = w; d.pf_bytes = wb;
          launch_conv(d, nullptr);
        }
        if (g_bench_rewarm & 8) rtd_launch(k_touch_read, dim3(64), dim3(256), 0, nullptr, (const unsigned*)x, xb / 128, (unsigned*)bias);
        HIP_CHECK(hipEventRecord(e0, nullptr));
        launch_conv(a, nullptr);
        HIP_CHECK(hipEventRecord(e1, nullptr));
        HIP_CHECK(hipEventSynchronize(e1));
        HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        tot += ms;
      }
      us_out[1] = tot * 1e3f / reps;
    }
    HIP_CHECK(hipDeviceSynchronize());
    if (getenv("RTD_CONV_STAMPS") && atoi(getenv("RTD_CONV_STAMPS")) == 3) {      // A-stationary kernel: per channel tile of blocks 0, 1, 40
      std::vector<long long> st((size_t)64 * 16 * 4);
      HIP_CHECK(hipMemcpy(st.data(), a.ws.slab, st.size() * 8, hipMemcpyDeviceToHost));
      for (int blk : {0, 1, 40}) {
        fprintf(stderr, "block %d: first tile landed %lld | per channel tile: K steps done, slab barrier passed, copy-out issued (shader clocks)\n", blk,
                st[((size_t)blk * 16 + 15) * 4]);
        for (int t = 0; t < 8; ++t) fprintf(stderr, "  tile %d: %7lld %7lld %7lld\n", t, st[((size_t)blk * 16 + t) * 4], st[((size_t)blk * 16 + t) * 4 + 1], st[((size_t)blk * 16 + t) * 4 + 2]);
      }
    } else if (getenv("RTD_CONV_STAMPS") && atoi(getenv("RTD_CONV_STAMPS")) == 2) {      // block-level stamps of the 128-pixel ws kernels
      const int nb = 4096;
      std::vector<long long> st((size_t)nb * 8);
      HIP_CHECK(hipMemcpy(st.data(), a.ws.slab, st.size() * 8, hipMemcpyDeviceToHost));
      long long t0 = -1;
      for (int i = 0; i < nb; ++i) if (st[i * 8 + 6] && (t0 < 0 || st[i * 8 + 6] < t0)) t0 = st[i * 8 + 6];
      double s_land = 0, s_k = 0, s_stage = 0, s_copy = 0, s_ack = 0; int cnt = 0; long long tend = 0;
      for (int i = 0; i < nb; ++i) {
        const long long* q = &st[(size_t)i * 8];
        if (!q[6] || !q[7]) continue;
        s_land += q[0]; s_k += q[1] - q[0]; s_stage += q[2] - q[1]; s_copy += q[3] - q[2]; s_ack += q[4] - q[3]; ++cnt;
        if (q[7] > tend) tend = q[7];
      }
      fprintf(stderr, "ws blocks stamped %d: mean clocks: first tile landed %.0f | K loop %.0f | staging %.0f | copy-out %.0f | store ack %.0f ; kernel wall %.2f us\n",
              cnt, s_land / cnt, s_k / cnt, s_stage / cnt, s_copy / cnt, s_ack / cnt, (tend - t0) * 0.01);
      {   // the core clock the blocks ran at: shader clocks (s_memtime) per 100 MHz tick (s_memrealtime) over a block's life
        double clk = 0; int c2 = 0;
        for (int i = 0; i < nb; ++i) { const long long* q = &st[(size_t)i * 8]; if (q[6] && q[7] > q[6]) { clk += (double)q[4] / ((double)(q[7] - q[6]) * 10.0); ++c2; } }
        if (c2) fprintf(stderr, "  in-kernel core clock %.3f GHz (mean over %d blocks)\n", clk / c2, c2);
      }
      for (int i : {0, 1, 600, 1500, 3000}) {
        const long long* q = &st[(size_t)i * 8];
        if (q[6]) fprintf(stderr, "  block %4d: start %+8.2f us  landed %6lld  kdone %6lld  staged %6lld  stored %6lld  acked %6lld  life %.2f us\n", i,
                          (q[6] - t0) * 0.01, q[0], q[1], q[2], q[3], q[4], (q[7] - q[6]) * 0.01);
      }
    } else if (getenv("RTD_CONV_STAMPS")) {           // with rtd_debug_option("glds_drop", 32) + conv_mode 7: per-K-step stamps of blocks 0..63
      std::vector<long long> st((size_t)64 * 48 * 8);
      HIP_CHECK(hipMemcpy(st.data(), a.ws.slab, st.size() * 8, hipMemcpyDeviceToHost));
      for (int blk : {0, 1, 17}) {
        fprintf(stderr, "block %d: ks | landed  barrier issued | mfma: barrier done   (shader clocks from kernel start)\n", blk);
        for (int ks = 0; ks < 20; ++ks) {
          const long long* q = &st[((size_t)blk * 48 + ks) * 8];
          fprintf(stderr, "  %2d | %7lld %7lld %7lld | %7lld %7lld\n", ks, q[0], q[1], q[2], q[3], q[4]);
        }
        const long long* z = &st[((size_t)blk * 48 + 47) * 8];
        fprintf(stderr, "  K loop done %lld, staged %lld, stores issued %lld, stores complete %lld clocks; wall %lld x10ns\n", z[0], z[1], z[2], z[3], z[6] - z[5]);
      }
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(a.ws.slab);
    (void)hipFree(x); (void)hipFree(y); (void)hipFree(w); (void)hipFree(bias);
    if (r) (void)hipFree(r);
    if (flush) (void)hipFree(flush);
  });
}

// Concurrency micro-benchmark (tools/pair_bench.py): conv A on one stream, conv B on another; us_out = {A alone, B alone,
// A and B issued together} per repetition (`reps` launches of each, events on both streams).
struct BenchConv {
  ConvArgs a;
  void *x = nullptr, *y = nullptr, *w = nullptr; float* bias = nullptr;
};
static void bench_conv_make(BenchConv& c, const int* sh) {   // sh: B, HW, Cin, Cout, K, stride, pad
  const int B = sh[0], H = sh[1], W = sh[1], Cin = sh[2], Cout = sh[3], KH = sh[4], stride = sh[5], pad = sh[6];
  const int K = KH * KH * Cin, Kpad = conv_kpad(K), Npad = conv_npad(Cout);
  const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KH) / stride + 1;
  const size_t xb = (size_t)B * H * W * Cin * 2, yb = (size_t)B * OH * OW * Cout * 2, wb = (size_t)Npad * Kpad * 2;
  HIP_CHECK(hipMalloc(&c.x, xb)); HIP_CHECK(hipMalloc(&c.y, yb)); HIP_CHECK(hipMalloc(&c.w, wb)); HIP_CHECK(hipMalloc((void**)&c.bias, Npad * 4));
  HIP_CHECK(hipMemset(c.x, 0, xb)); HIP_CHECK(hipMemset(c.w, 0, wb)); HIP_CHECK(hipMemset(c.bias, 0, Npad * 4));
  c.a.x = mk(c.x, BF16, B, H, W, Cin);
  c.a.y = mk(c.y, BF16, B, OH, OW, Cout);
  c.a.w = c.w; c.a.bias = c.bias; c.a.KH = KH; c.a.KW = KH; c.a.stride = stride; c.a.pad = pad; c.a.Kpad = Kpad; c.a.Npad = Npad;
  c.a.act = 1; c.a.res_mode = RES_NONE;
}
int rtd_bench_conv_pair(const int* shape_a, const int* shape_b, int reps, float* us_out) {
  return op_guard([&] {
    BenchConv A, B;
    bench_conv_make(A, shape_a);
    bench_conv_make(B, shape_b);
    hipStream_t s1, s2;
    HIP_CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    HIP_CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t e0, e1, f0, f1;
    HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1)); HIP_CHECK(hipEventCreate(&f0)); HIP_CHECK(hipEventCreate(&f1));
    auto run = [&](bool ra, bool rb) {
      for (int i = 0; i < 3; ++i) { if (ra) launch_conv(A.a, s1); if (rb) launch_conv(B.a, s2); }
      HIP_CHECK(hipDeviceSynchronize());
      const auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < reps; ++i) { if (ra) launch_conv(A.a, s1); if (rb) launch_conv(B.a, s2); }
      HIP_CHECK(hipStreamSynchronize(s1));
      HIP_CHECK(hipStreamSynchronize(s2));
      return (float)(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps);
    };
    us_out[0] = run(true, false);
    us_out[1] = run(false, true);
    us_out[2] = run(true, true);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(f0); (void)hipEventDestroy(f1);
    (void)hipStreamDestroy(s1); (void)hipStreamDestroy(s2);
    for (BenchConv* c : {&A, &B}) { (void)hipFree(c->x); (void)hipFree(c->y); (void)hipFree(c->w); (void)hipFree(c->bias); }
  });
}

int rtd_bench_mfma_rate(int random_operands, int ms_target, float* out) {
  // device buffers and events are released on every path (a throwing HIP_CHECK included)
  struct Scratch {
    unsigned* seed = nullptr; long long* stamps = nullptr; float* sink = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ~Scratch() {
      if (e0) (void)hipEventDestroy(e0);
      if (e1) (void)hipEventDestroy(e1);
      if (seed) (void)hipFree(seed);
      if (stamps) (void)hipFree(stamps);
      if (sink) (void)hipFree(sink);
    }
  };
  return op_guard([&] {
    RTD_CHECK(out && ms_target >= 1 && ms_target <= 2000, RTD_E_INVALID, "arguments");
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));                              // the CURRENT device (a rank's own GPU), not device 0
    hipDeviceProp_t prop;
    HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    const int cus = prop.multiProcessorCount;
    std::vector<unsigned> hs(4096, 0u);
    if (random_operands)
      for (int i = 0; i < 4096; ++i) {
        unsigned h = (unsigned)i * 2654435761u + 12345u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        const unsigned lo = (h & 0x8000u) | ((10u + ((h >> 20) & 7u)) << 10) | (h & 0x3ffu);
        unsigned h2 = h * 2654435761u; h2 ^= h2 >> 16;
        const unsigned hi = (h2 & 0x8000u) | ((10u + ((h2 >> 20) & 7u)) << 10) | (h2 & 0x3ffu);
        hs[i] = lo | (hi << 16);
      }
    Scratch sc;
    HIP_CHECK(hipMalloc((void**)&sc.seed, 4096 * 4)); HIP_CHECK(hipMalloc((void**)&sc.stamps, (size_t)cus * 4 * 2 * 8)); HIP_CHECK(hipMalloc((void**)&sc.sink, 16));
    HIP_CHECK(hipMemcpy(sc.seed, hs.data(), 4096 * 4, hipMemcpyHostToDevice));
    // 16 MFMAs of 16 cycles per iteration at <= 2.4 GHz: 9400 iterations per millisecond
    const int iters = ms_target * 9400;
    HIP_CHECK(hipEventCreate(&sc.e0)); HIP_CHECK(hipEventCreate(&sc.e1));
    float ms = 0.f;
    for (int rep = 0; rep < 3; ++rep) {
      HIP_CHECK(hipEventRecord(sc.e0, nullptr));
      rtd_launch(k_mfma_rate, dim3(cus), dim3(256), 0, nullptr, sc.seed, iters, sc.stamps, sc.sink);
      HIP_CHECK(hipEventRecord(sc.e1, nullptr));
      HIP_CHECK(hipEventSynchronize(sc.e1));
      HIP_CHECK(hipEventElapsedTime(&ms, sc.e0, sc.e1));
    }
    std::vector<long long> h((size_t)cus * 4 * 2);
    HIP_CHECK(hipMemcpy(h.data(), sc.stamps, h.size() * 8, hipMemcpyDeviceToHost));
    double cyc = 0, rt = 0;
    for (int w = 0; w < cus * 4; ++w) { cyc += (double)h[2 * w]; rt += (double)h[2 * w + 1]; }
    out[0] = (float)((double)iters * 16.0 * cus * 4.0 * 16384.0 / (ms * 1e-3) / 1e12);
    out[1] = (float)(cyc / (rt * 10.0));
    out[2] = ms;
  });
}

int rtd_op_layernorm(int dtype, const void* x, const void* res, const float* g, const float* b, void* y, int rows, int dim, int out_f32) {
  return op_guard([&] {
    Tensor tx = mk(x, dtype, 1, rows, 1, dim), ty = mk(y, out_f32 ? F32 : dtype, 1, rows, 1, dim), tr = mk(res, dtype, 1, rows, 1, dim);
    launch_layernorm(tx, res ? &tr : nullptr, g, b, ty, 1e-5f, nullptr);
  });
}

int rtd_op_attention(int dtype, const void* qk, const void* v, void* o, int B, int L, int heads, int hd) {
  return op_guard([&] {
    const int D = heads * hd;
    launch_attention(mk(qk, dtype, B, L, 1, 2 * D), mk(v, dtype, B, L, 1, D), mk(o, dtype, B, L, 1, D), heads, nullptr);
  });
}

int rtd_op_msdeform(int dtype, const void* value, const float* offaw, const float* ref, float* out, int B, int Q, int heads, int hd,
                    int n_levels, int n_points, const int32_t* level_hw, int value_ld, float offset_scale) {
  return op_guard([&] {
    RTD_CHECK(n_levels >= 1 && n_levels <= 8, RTD_E_INVALID, "n_levels");
    int32_t lv[24]; int S = 0;
    for (int l = 0; l < n_levels; ++l) { lv[l * 3] = level_hw[2 * l]; lv[l * 3 + 1] = level_hw[2 * l + 1]; lv[l * 3 + 2] = S; S += level_hw[2 * l] * level_hw[2 * l + 1]; }
    int32_t* lvd = nullptr; float* ref8 = nullptr;
    HIP_CHECK(hipMalloc((void**)&lvd, sizeof lv));
    HIP_CHECK(hipMemcpy(lvd, lv, sizeof lv, hipMemcpyHostToDevice));
    HIP_CHECK(hipMalloc((void**)&ref8, (size_t)B * Q * 32));
    HIP_CHECK(hipMemset(ref8, 0, (size_t)B * Q * 32));
    HIP_CHECK(hipMemcpy2D(ref8, 32, ref, 16, 16, (size_t)B * Q, hipMemcpyDeviceToDevice));
    Tensor tv = mk(value, dtype, B, S, 1, heads * hd);
    tv.ld = value_ld; tv.bstride = (int64_t)S * value_ld;
    launch_msdeform(tv, 0, mk(offaw, F32, B, Q, 1, heads * n_levels * n_points * 3), ref8, mk(out, F32, B, Q, 1, heads * hd), heads, hd,
                    n_levels, n_points, lvd, offset_scale, nullptr);
    HIP_CHECK(hipDeviceSynchronize());
    (void)hipFree(lvd); (void)hipFree(ref8);
  });
}


int rtd_op_topk(const float* keys, int B, int N, int K, int32_t* idx_out, float* val_out) {
  return op_guard([&] { launch_topk(keys, B, N, K, idx_out, val_out, nullptr); });
}

int rtd_op_resize(const uint8_t* src, int sh, int sw, void* dst, int dh, int dw, int dtype) {
  return op_guard([&] {
    std::vector<int32_t> hb, hk, vb, vk;
    int hks, vks;
    pil_coeffs(sw, dw, hb, hk, hks);
    pil_coeffs(sh, dh, vb, vk, vks);
    std::vector<void*> tmp;
    auto up = [&](const std::vector<int32_t>& v) {
      void* d = nullptr;
      HIP_CHECK(hipMalloc(&d, v.size() * 4));
      tmp.push_back(d);
      HIP_CHECK(hipMemcpy(d, v.data(), v.size() * 4, hipMemcpyHostToDevice));
      return (const int32_t*)d;
    };
    ResizeCoef c;
    c.hb = up(hb); c.hk = up(hk); c.vb = up(vb); c.vk = up(vk); c.hks = hks; c.vks = vks;
    uint8_t* t = nullptr;
    HIP_CHECK(hipMalloc((void**)&t, (size_t)sh * dw * 3));
    tmp.push_back(t);
    launch_resize_pil(src, sh, sw, t, mk(dst, dtype, 1, dh, dw, 8), 0, c, nullptr);
    HIP_CHECK(hipDeviceSynchronize());
    for (void* p : tmp) (void)hipFree(p);
  });
}

}  // extern "C"
