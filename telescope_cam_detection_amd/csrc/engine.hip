// engine.hip - host side of libmi355rtdetr.so: weight container, execution plan, hipGraph, C ABI.
//
// The engine owns, per handle: the device copy of the (BN-folded, RepVGG-fused) weights in the
// handle's precision, one activation arena + execution plan per batch size, a HIP stream and
// (optionally) one hipGraph per plan, BUILT node by node from the plan (common.h rtd_launch / GraphBuild) - never captured from a stream.  The graph of layers below is the RT-DETRv2 graph of
// HF:rt_detr_v2/modeling_rt_detr_v2.py / rt_detr/modeling_rt_detr_resnet.py (see oracle/rtdetr_oracle.py
// for the line-by-line CPU restatement it is tested against).
#include "engine_internal.h"

using namespace rtd;
using namespace rtd_eng;

namespace rtd {
__thread GraphBuild* t_graph_build = nullptr;
__thread long long t_launches = 0;
}  // namespace rtd

void launch_force_idx(int32_t* dst, const int32_t* src, const int32_t* flag, int n, hipStream_t s);

namespace rtd_eng {
thread_local std::string g_create_error;
PlanOpts g_opts;
}  // namespace rtd_eng

namespace rtd_eng {

// ------------------------------------------------------------------------------------------ blob
void parse_blob(rtd_engine* e) {
  const char* b = e->blob.data();
  const size_t n = e->blob.size();
  RTD_CHECK(n >= 12 && memcmp(b, "RTDW", 4) == 0, RTD_E_WEIGHTS, "weight blob: bad magic");
  uint32_t ver, count;
  memcpy(&ver, b + 4, 4);
  memcpy(&count, b + 8, 4);
  RTD_CHECK(ver == 1, RTD_E_WEIGHTS, "weight blob: unsupported version");
  size_t p = 12;
  for (uint32_t i = 0; i < count; ++i) {
    RTD_CHECK(p + 2 <= n, RTD_E_WEIGHTS, "weight blob: truncated table");
    uint16_t nl;
    memcpy(&nl, b + p, 2); p += 2;
    RTD_CHECK(p + nl + 4 <= n, RTD_E_WEIGHTS, "weight blob: truncated table");
    std::string name(b + p, nl); p += nl;
    uint32_t nd;
    memcpy(&nd, b + p, 4); p += 4;
    RTD_CHECK(nd <= 8 && p + 4 * nd + 16 <= n, RTD_E_WEIGHTS, "weight blob: truncated table");
    HostTensor t;
    for (uint32_t d = 0; d < nd; ++d) {
      uint32_t v;
      memcpy(&v, b + p, 4); p += 4;
      t.shape.push_back(v);
    }
    uint64_t off, nb;
    memcpy(&off, b + p, 8); p += 8;
    memcpy(&nb, b + p, 8); p += 8;
    RTD_CHECK(off % 4 == 0 && off + nb <= n && (int64_t)nb == t.numel() * 4, RTD_E_WEIGHTS, "weight blob: bad tensor extent: " + name);
    t.data = (const float*)(b + off);
    e->host[name] = t;
  }
}

// Real-weights guard, part 1: a checkpoint is refused at load time when a tensor holds a NaN / Inf, or - on the pair
// engine, whose operands are fp16 hi/lo halves that SATURATE at +-65504 (common.h split2, f2h_rne below) - when a folded filter value does
// not fit that range: the result would be finite, plausible and wrong.  (BN folding multiplies a filter row by gamma / sqrt(var + eps):
// a collapsed running_var on a trained checkpoint is how such values arise.)
void check_weight_range(rtd_engine* e) {
  e->max_abs_filter = 0.f;
  e->max_abs_filter_name.clear();
  for (const auto& kv : e->host) {
    const HostTensor& t = kv.second;
    const int64_t n = t.numel();
    float mx = 0.f;
    bool finite = true;
    for (int64_t i = 0; i < n; ++i) {
      const float v = t.data[i];
      if (!std::isfinite(v)) { finite = false; break; }
      mx = std::max(mx, fabsf(v));
    }
    RTD_CHECK(finite, RTD_E_WEIGHTS, "weight blob: tensor " + kv.first + " holds a NaN or an infinity");
    const bool is_filter = kv.first.size() > 2 && kv.first.compare(kv.first.size() - 2, 2, ".w") == 0;
    if (is_filter && mx > e->max_abs_filter) { e->max_abs_filter = mx; e->max_abs_filter_name = kv.first; }
  }
  if (e->cfg.precision == RTD_PREC_F16X3)
    RTD_CHECK(e->max_abs_filter <= 65504.f, RTD_E_WEIGHTS,
              "weight blob: folded filter " + e->max_abs_filter_name + " reaches " + std::to_string(e->max_abs_filter) +
                  ", beyond the fp16 pair format's range (65504): load this checkpoint with precision fp32");
}

const HostTensor& host_tensor(rtd_engine* e, const std::string& name) {
  auto it = e->host.find(name);
  RTD_CHECK(it != e->host.end(), RTD_E_WEIGHTS, "weight blob: missing tensor " + name);
  return it->second;
}

// host [Npad][Kcols] fp32 (zero padded) -> device filter in dt: fp32 as is, bf16 rounded, F16X2 as [32 hi | 32 lo] groups along K
// (Kcols % 32 == 0: a row has 2 * Kcols bf16 elements, the layout of an activation pixel with Kcols channels)
static const char* dt_tag(int dt) { return dt == BF16 ? "#bf16" : (dt == F16X2 ? "#f16x2" : "#f32"); }
static void* upload_filter(rtd_engine* e, const std::vector<float>& pad, int Npad, int Kcols, int dt) {
  float* tmp = nullptr;
  void* out = nullptr;
  HIP_CHECK(hipMalloc((void**)&tmp, pad.size() * 4));
  hipError_t er = hipMemcpy(tmp, pad.data(), pad.size() * 4, hipMemcpyHostToDevice);
  try {
    if (er == hipSuccess) {
      if (dt == F32) {
        out = tmp;
        e->allocs.push_back(tmp);
        tmp = nullptr;
      } else if (dt == F16X2) {
        out = e->dmalloc(pad.size() * 4);
        launch_f32_to_split(tmp, Kcols, out, Kcols, Npad, Kcols, e->stream);
        er = hipStreamSynchronize(e->stream);
      } else {
        out = e->dmalloc(pad.size() * 2);
        launch_f32_to(tmp, out, BF16, (int64_t)pad.size(), e->stream);
        er = hipStreamSynchronize(e->stream);
      }
    }
  } catch (...) {
    if (tmp) (void)hipFree(tmp);
    throw;
  }
  if (tmp) (void)hipFree(tmp);
  HIP_CHECK(er);
  return out;
}

// filter [N][K] fp32 -> device [Npad][Kpad] in dt (zero padded), bias -> fp32 [Npad]
DevWeight get_weight(rtd_engine* e, const std::string& name, int dt, int N, int K) {
  const std::string key = name + dt_tag(dt);
  auto it = e->wcache.find(key);
  if (it != e->wcache.end()) return it->second;
  const HostTensor& w = host_tensor(e, name + ".w");
  const HostTensor& b = host_tensor(e, name + ".b");
  RTD_CHECK(!w.shape.empty() && w.shape[0] == N && w.numel() == (int64_t)N * K, RTD_E_WEIGHTS, "weight shape mismatch: " + name);
  RTD_CHECK(b.numel() == N, RTD_E_WEIGHTS, "bias shape mismatch: " + name);
  DevWeight d;
  d.N = N; d.K = K; d.Kpad = dt == F16X2 ? conv_kpad_split(K) : conv_kpad(K); d.Npad = conv_npad(N); d.dt = dt;
  const int kcols = dt == F16X2 ? d.Kpad / 2 : d.Kpad;
  std::vector<float> pad((size_t)d.Npad * kcols, 0.f);
  for (int r = 0; r < N; ++r) memcpy(&pad[(size_t)r * kcols], w.data + (size_t)r * K, (size_t)K * 4);
  d.w = upload_filter(e, pad, d.Npad, kcols, dt);
  std::vector<float> bp(d.Npad, 0.f);
  memcpy(bp.data(), b.data, (size_t)N * 4);
  d.bias = (float*)e->dmalloc(bp.size() * 4);
  HIP_CHECK(hipMemcpy(d.bias, bp.data(), bp.size() * 4, hipMemcpyHostToDevice));
  e->wcache[key] = d;
  return d;
}

// two filters over the same output channels, concatenated along K: [N][K1 | K2] -> device [Npad][Kpad], bias = b1 + b2
// (a block's last conv with its projection shortcut folded in, ConvArgs::x2)
DevWeight get_weight_cat(rtd_engine* e, const std::string& n1, const std::string& n2, int dt, int N, int K1, int K2) {
  const std::string key = n1 + "+" + n2 + dt_tag(dt);
  auto it = e->wcache.find(key);
  if (it != e->wcache.end()) return it->second;
  const HostTensor& w1 = host_tensor(e, n1 + ".w");
  const HostTensor& w2 = host_tensor(e, n2 + ".w");
  const HostTensor& b1 = host_tensor(e, n1 + ".b");
  const HostTensor& b2 = host_tensor(e, n2 + ".b");
  RTD_CHECK(w1.numel() == (int64_t)N * K1 && w2.numel() == (int64_t)N * K2 && b1.numel() == N && b2.numel() == N, RTD_E_WEIGHTS,
            "weight shape mismatch: " + n1 + " + " + n2);
  DevWeight d;
  d.N = N; d.K = K1 + K2; d.Kpad = dt == F16X2 ? conv_kpad_split(d.K) : conv_kpad(d.K); d.Npad = conv_npad(N); d.dt = dt;
  const int kcols = dt == F16X2 ? d.Kpad / 2 : d.Kpad;
  std::vector<float> pad((size_t)d.Npad * kcols, 0.f);
  for (int r = 0; r < N; ++r) {
    memcpy(&pad[(size_t)r * kcols], w1.data + (size_t)r * K1, (size_t)K1 * 4);
    memcpy(&pad[(size_t)r * kcols + K1], w2.data + (size_t)r * K2, (size_t)K2 * 4);
  }
  d.w = upload_filter(e, pad, d.Npad, kcols, dt);
  std::vector<float> bp(d.Npad, 0.f);
  for (int r = 0; r < N; ++r) bp[r] = b1.data[r] + b2.data[r];
  d.bias = (float*)e->dmalloc(bp.size() * 4);
  HIP_CHECK(hipMemcpy(d.bias, bp.data(), bp.size() * 4, hipMemcpyHostToDevice));
  e->wcache[key] = d;
  return d;
}

// fp32 filter [N][K] -> fragment-major layout of decoder.hip's row_gemm (K padded to Kuse, N to 8 tiles)
// host-side fp16 round-to-nearest-even (subnormals kept, saturating like common.h split2) through the compiler's _Float16
static inline uint16_t f2h_rne(float f) {
  f = f > 65504.f ? 65504.f : (f < -65504.f ? -65504.f : f);
  const _Float16 h = (_Float16)f;
  uint16_t u;
  memcpy(&u, &h, 2);
  return u;
}
static inline float h2f(uint16_t u) {
  _Float16 h;
  memcpy(&h, &u, 2);
  return (float)h;
}

DevWeight get_weight_packed(rtd_engine* e, const std::string& name, int N, int K, int Kuse, bool split = false) {
  const std::string key = name + (split ? "#split" : "#packed");
  auto it = e->wcache.find(key);
  if (it != e->wcache.end()) return it->second;
  const HostTensor& w = host_tensor(e, name + ".w");
  const HostTensor& b = host_tensor(e, name + ".b");
  RTD_CHECK(!w.shape.empty() && w.shape[0] == N && w.numel() == (int64_t)N * K, RTD_E_WEIGHTS, "weight shape mismatch: " + name);
  RTD_CHECK(b.numel() == N && Kuse % 64 == 0 && Kuse >= K, RTD_E_WEIGHTS, "bias / K padding: " + name);
  DevWeight d;
  d.N = N; d.K = Kuse; d.Kpad = Kuse; d.dt = F32;
  const int ntiles = ((N + 15) / 16 + 7) / 8 * 8, kc = Kuse / 16;
  d.Npad = ntiles * 16;
  std::vector<float> pk(split ? 0 : (size_t)ntiles * kc * 256, 0.f);
  for (int t = 0; t < (split ? 0 : ntiles); ++t)
    for (int c = 0; c < kc; ++c)
      for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 4; ++j) {
          const int n = t * 16 + (lane & 15), k = c * 16 + 4 * (lane >> 4) + j;
          if (n < N && k < K) pk[(((size_t)t * kc + c) * 64 + lane) * 4 + j] = w.data[(size_t)n * K + k];
        }
  if (split) {
    // W = hi + lo (two fp16, round-to-nearest-even each): decoder.hip row_gemm_split multiplies both against a hi/lo split of
    // the activations with 3 fp16 MFMAs (hi*hi + hi*lo + lo*hi); the dropped lo*lo term is ~2^-22 relative
    const int kc32 = Kuse / 32;
    std::vector<uint16_t> ps((size_t)ntiles * kc32 * 2 * 512, 0);
    for (int t = 0; t < ntiles; ++t)
      for (int c = 0; c < kc32; ++c)
        for (int lane = 0; lane < 64; ++lane)
          for (int j = 0; j < 8; ++j) {
            const int n = t * 16 + (lane & 15), k = c * 32 + 8 * (lane >> 4) + j;
            if (n < N && k < K) {
              const float v = w.data[(size_t)n * K + k];
              const uint16_t hi = f2h_rne(v);
              const uint16_t lo = f2h_rne(v - h2f(hi));
              const size_t base = ((size_t)t * kc32 + c) * 1024;
              ps[base + lane * 8 + j] = hi;
              ps[base + 512 + lane * 8 + j] = lo;
            }
          }
    d.w = e->dmalloc(ps.size() * 2);
    HIP_CHECK(hipMemcpy(d.w, ps.data(), ps.size() * 2, hipMemcpyHostToDevice));
  } else {
  d.w = e->dmalloc(pk.size() * 4);
  HIP_CHECK(hipMemcpy(d.w, pk.data(), pk.size() * 4, hipMemcpyHostToDevice));
  }
  std::vector<float> bp(d.Npad, 0.f);
  memcpy(bp.data(), b.data, (size_t)N * 4);
  d.bias = (float*)e->dmalloc(bp.size() * 4);
  HIP_CHECK(hipMemcpy(d.bias, bp.data(), bp.size() * 4, hipMemcpyHostToDevice));
  e->wcache[key] = d;
  return d;
}

float* get_vec(rtd_engine* e, const std::string& name, int n) {
  auto it = e->vcache.find(name);
  if (it != e->vcache.end()) return it->second;
  const HostTensor& t = host_tensor(e, name);
  RTD_CHECK(t.numel() == n, RTD_E_WEIGHTS, "vector shape mismatch: " + name);
  float* d = (float*)e->dmalloc((size_t)n * 4);
  HIP_CHECK(hipMemcpy(d, t.data, (size_t)n * 4, hipMemcpyHostToDevice));
  e->vcache[name] = d;
  return d;
}

// ------------------------------------------------------------------------------------------ geometry
int down2(int n) { return (n + 2 - 3) / 2 + 1; }  // k=3, s=2, p=1

// 2D sin-cos position embedding, HF:v2.py:955-1000 : [sin_h | cos_h | sin_w | cos_w], float64 math
std::vector<float> sincos_pos(int h, int w, int dim) {
  const int pd = dim / 4;
  std::vector<float> out((size_t)h * w * dim);
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      float* o = &out[((size_t)y * w + x) * dim];
      for (int i = 0; i < pd; ++i) {
        const double omega = 1.0 / pow(10000.0, (double)i / pd);
        o[i] = (float)sin(y * omega);
        o[pd + i] = (float)cos(y * omega);
        o[2 * pd + i] = (float)sin(x * omega);
        o[3 * pd + i] = (float)cos(x * omega);
      }
    }
  return out;
}

// anchors + valid mask, HF:v2.py:1423-1449 in fp32 arithmetic; invalid -> FLT_MAX
void make_anchors(rtd_engine* e, std::vector<float>& anchors, std::vector<int32_t>& invalid) {
  anchors.assign((size_t)e->S * 4, 0.f);
  invalid.clear();
  for (int l = 0; l < 3; ++l) {
    const int h = e->lvl_h[l], w = e->lvl_w[l];
    const float wh = 0.05f * (float)(1 << l);
    for (int y = 0; y < h; ++y)
      for (int x = 0; x < w; ++x) {
        const int t = e->lvl_start[l] + y * w + x;
        float a[4] = {((float)x + 0.5f) / (float)w, ((float)y + 0.5f) / (float)h, wh, wh};
        bool valid = true;
        for (int k = 0; k < 4; ++k) valid = valid && (a[k] > 1e-2f) && (a[k] < 1.f - 1e-2f);
        for (int k = 0; k < 4; ++k) anchors[(size_t)t * 4 + k] = valid ? logf(a[k] / (1.f - a[k])) : 3.402823466e+38f;
        if (!valid) invalid.push_back(t);
      }
  }
}

// Pillow's precompute_coeffs + normalize_coeffs_8bpc for the bilinear filter (support 1.0),
// box = the whole source; the published algorithm of ImagingResample (Pillow src/libImaging/Resample.c).
void pil_coeffs(int in_size, int out_size, std::vector<int32_t>& bounds, std::vector<int32_t>& kk, int& ksize) {
  const double scale = (double)in_size / out_size;
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = 1.0 * filterscale;
  ksize = (int)ceil(support) * 2 + 1;
  bounds.assign((size_t)out_size * 2, 0);
  kk.assign((size_t)out_size * ksize, 0);
  std::vector<double> k(ksize);
  for (int xx = 0; xx < out_size; ++xx) {
    const double center = (xx + 0.5) * scale;
    const double ss = 1.0 / filterscale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) {
      double t = (x + xmin - center + 0.5) * ss;
      if (t < 0.0) t = -t;
      const double wv = t < 1.0 ? 1.0 - t : 0.0;
      k[x] = wv;
      ww += wv;
    }
    for (int x = 0; x < xmax; ++x) {
      if (ww != 0.0) k[x] /= ww;
      const double v = k[x] * (double)(1 << 22);
      kk[(size_t)xx * ksize + x] = v < 0 ? (int)(-0.5 + v) : (int)(0.5 + v);
    }
    bounds[(size_t)xx * 2] = xmin;
    bounds[(size_t)xx * 2 + 1] = xmax;
  }
}

const ResizeCoef& resize_tables(rtd_engine* e, int sh, int sw) {
  auto key = std::make_pair(sh, sw);
  auto it = e->resize.find(key);
  if (it != e->resize.end()) return it->second.coef;
  std::vector<int32_t> hb, hk, vb, vk;
  int hks, vks;
  pil_coeffs(sw, e->cfg.input_w, hb, hk, hks);
  pil_coeffs(sh, e->cfg.input_h, vb, vk, vks);
  ResizeTables t;
  auto up = [&](const std::vector<int32_t>& v) {
    void* d = e->dmalloc(v.size() * 4);
    HIP_CHECK(hipMemcpy(d, v.data(), v.size() * 4, hipMemcpyHostToDevice));
    return (const int32_t*)d;
  };
  t.coef.hb = up(hb); t.coef.hk = up(hk); t.coef.vb = up(vb); t.coef.vk = up(vk);
  t.coef.hks = hks; t.coef.vks = vks;
  e->resize[key] = t;
  return e->resize[key].coef;
}

// ------------------------------------------------------------------------------------------ plan builder
struct Builder {
  rtd_engine* e;
  Plan* plan;
  bool dry;
  size_t off = 0;

  void* alloc(size_t bytes) {
    off = (off + 255) / 256 * 256;
    void* p = dry ? nullptr : (char*)plan->arena + off;
    off += bytes;
    return p;
  }
  Tensor act(int dt, int n, int h, int w, int c, const std::string& name = "") {
    Tensor t;
    t.dt = dt; t.n = n; t.h = h; t.w = w; t.c = c; t.ld = c; t.bstride = (int64_t)h * w * c;
    const size_t nbytes = (size_t)n * h * w * c * dtype_size(dt);
    t.p = alloc(nbytes);
    if (!dry && dt == F16X2) plan->split_acts.emplace_back(t.p, nbytes);
    if (!name.empty()) plan->named[name] = t;
    return t;
  }
  int lane = 0;              // lane of the ops being pushed
  void push(const std::string& name, const char* kernel, double flops, double bytes, std::function<void(hipStream_t)> f) {
    if (dry) return;
    plan->ops.push_back(Op{name, kernel, flops, bytes, std::move(f), false, lane, 0});
  }
  void marker(int kind) {
    if (dry) return;
    plan->ops.push_back(Op{kind == 1 ? "fork" : "join", "sync", 0.0, 0.0, nullptr, false, 0, kind});
  }
  static double tbytes(const Tensor& t) { return (double)t.pixels() * t.c * dtype_size(t.dt); }

  // conv / linear.  `y` may be a channel-slice view of a wider buffer.
  // `x2` + `name2`: a second input read as an extra 1x1 tap at output resolution, its filter `name2` concatenated along K
  // (ConvArgs::x2: the projection shortcut folded into the block's last conv)
  void conv(const std::string& name, const Tensor& x, const Tensor& y, int k, int stride, int pad, int act,
            const Tensor* res = nullptr, int res_mode = RES_NONE, int real_cin = 0, const Tensor* x2 = nullptr,
            const std::string& name2 = "", int x_up2 = 0, const Tensor* next_y = nullptr, const std::string& next_name = "", int next_act = ACT_NONE) {
    const int K = k * k * x.c + (x2 ? x2->c : 0);
    DevWeight w;
    // name2 empty: `name` is already the filter over [x | x2] (a conv over a concatenation that is read from its two sources)
    if (!dry) w = (x2 && !name2.empty()) ? get_weight_cat(e, name, name2, x.dt, y.c, k * k * x.c, x2->c) : get_weight(e, name, x.dt, y.c, K);
    else { w.Kpad = x.dt == F16X2 ? conv_kpad_split(K) : conv_kpad(K); w.Npad = conv_npad(y.c); }
    ConvArgs a;
    a.x = x; a.y = y; a.w = w.w; a.bias = w.bias;
    if (x2) a.x2 = *x2;
    a.x_up2 = x_up2;
    a.KH = k; a.KW = k; a.stride = stride; a.pad = pad; a.Kpad = w.Kpad; a.Npad = w.Npad;
    a.act = act; a.res_mode = res ? res_mode : RES_NONE;
    if (res) a.res = *res;
    if (next_y) {                                              // the following 1x1 conv rides on this launch (ConvArgs::next_*)
      DevWeight wn;
      if (!dry) wn = get_weight(e, next_name, x.dt, next_y->c, y.c);
      else { wn.Kpad = x.dt == F16X2 ? conv_kpad_split(y.c) : conv_kpad(y.c); wn.Npad = conv_npad(next_y->c); }
      a.next_w = wn.w; a.next_bias = wn.bias; a.next_y = *next_y; a.next_kpad = wn.Kpad; a.next_act = next_act;
    }
    a.opts = &e->conv_opts;
    a.prefer256 = e->cfg.profile == RTD_PROFILE_THROUGHPUT;
    if (avg_pending) { a.avg_y = avg_pending_y; if (dry) a.avg_y.p = nullptr; avg_pending = false; }
    if (y_dead_pending) { a.y_dead = (a.avg_y.c && next_y) ? 1 : 0; y_dead_pending = false; }
    slab_need[lane] = std::max(slab_need[lane], conv_split_slab_bytes(a));
    const double M = (double)y.pixels();
    const double kreal = (double)k * k * (real_cin ? real_cin : x.c) + (x2 ? x2->c : 0);
    const double flops = 2.0 * M * y.c * kreal + (next_y ? 2.0 * M * y.c * next_y->c : 0.0);
    const double bytes = (double)x.pixels() * x.c * dtype_size(x.dt) + tbytes(y) + (double)y.c * K * dtype_size(x.dt) +
                         (res ? tbytes(*res) : 0.0) + (x2 ? tbytes(*x2) : 0.0) + (next_y ? tbytes(*next_y) : 0.0) + (a.avg_y.c ? tbytes(a.avg_y) : 0.0) -
                         (a.y_dead ? tbytes(y) : 0.0);
    auto ap = std::make_shared<ConvArgs>(a);
    if (!dry) {
      // chain: the previous conv launch of the plan prefetches THIS filter while it runs (ConvArgs::pf)
      if (last_conv) { last_conv->pf = w.w; last_conv->pf_bytes = (size_t)w.Npad * w.Kpad * (x.dt == F32 ? 4 : 2); }
      last_conv = ap;
      convs.push_back({ap, lane});
    }
    push(name, "conv_igemm", flops, bytes, [ap](hipStream_t s) { launch_conv(*ap, s); });
  }
  // set by the plan builder right before the conv that should also write the 2 x 2 average of its output (ConvArgs::avg_y); consumed by conv()
  bool avg_pending = false;
  Tensor avg_pending_y;
  bool y_dead_pending = false;   // the next conv's output has no reader outside that launch (see ConvArgs::y_dead)
  std::shared_ptr<ConvArgs> last_conv;
  // two-pass split-K workspace of this plan: sized for the plan's own batch (the slice count depends on per-image extents only, so every
  // batch size up to max_batch runs the same arithmetic), allocated once every conv is known.  ONE SLAB PER LANE: launches of a lane are
  // sequential, but lane 1 (the side stream) runs beside lane 0 - two splitting convs on different lanes must never share partial sums.
  std::vector<std::pair<std::shared_ptr<ConvArgs>, int>> convs;
  size_t slab_need[2] = {0, 0};
  void finish_workspace() {
    ConvWorkspace ws[2];
    for (int l = 0; l < 2; ++l) {
      if (!slab_need[l]) continue;
      ws[l].slab = (float*)alloc(slab_need[l]);
      ws[l].slab_bytes = slab_need[l];
    }
    for (auto& c : convs) c.first->ws = ws[c.second];
  }
  Tensor linear(const std::string& name, const Tensor& x, int N, int odt, int act, const Tensor* res = nullptr,
                const std::string& tname = "") {
    Tensor y = this->act(odt, x.n, x.h, x.w, N, tname);
    conv(name, x, y, 1, 1, 0, act, res, RES_PRE);
    return y;
  }
  // dense fp32 rows -> F16X2 rows (the split engine's trunk type) as its own launch
  Tensor to_split(const std::string& name, const Tensor& x, const std::string& tname = "") {
    Tensor y = act(F16X2, x.n, x.h, x.w, x.c, tname);
    const int64_t rows = x.pixels();
    push(name, "convert", 0.0, tbytes(x) + tbytes(y), [x, y, rows](hipStream_t s) { launch_f32_to_split((const float*)x.p, x.ld, y.p, y.ld, rows, x.c, s); });
    return y;
  }
  Tensor layernorm(const std::string& name, const Tensor& x, int odt, const std::string& tname = "") {
    if (odt == F16X2) return to_split(name + ".split", layernorm(name, x, F32), tname);
    Tensor y = act(odt, x.n, x.h, x.w, x.c, tname);
    if (dry) return y;
    const float* g = get_vec(e, name + ".g", x.c);
    const float* b = get_vec(e, name + ".b", x.c);
    push(name, "layernorm", 8.0 * x.pixels() * x.c, tbytes(x) + tbytes(y),
         [x, y, g, b](hipStream_t s) { launch_layernorm(x, nullptr, g, b, y, 1e-5f, s); });
    return y;
  }
};

void build_graph(rtd_engine* e, Builder& B, int n) {
  const rtd_config& c = e->cfg;
  const int P = e->P;
  const bool SP = P == F16X2;            // rtd_config.precision = RTD_PREC_F16X3: the trunk carries hi/lo fp16 pairs
  const int H = c.input_h, W = c.input_w;
  Plan* plan = B.plan;
  auto nm = [](const char* fmt, int a = 0, int b = 0) {
    char buf[96];
    snprintf(buf, sizeof buf, fmt, a, b);
    return std::string(buf);
  };

  // ---- input + stem (HF:rt_detr_resnet.py:71-114) ----------------------------------------------
  Tensor x = B.act(SP ? F32 : P, n, H, W, 8, "input");   // split engine: fp32 pixels, stem.0 runs on fp32 MFMAs (K = 27) and writes F16X2
  plan->input = x;
  const int eh = c.embedding_size / 2;
  int h = down2(H), w = down2(W);
  Tensor s0 = B.act(P, n, h, w, eh);
  // f16x3: on by default - the generic form there is a 105 MB fp32 NHWC-8 image + an fp32-MFMA stem conv (30 + 180 us at R50 bs 8)
  plan->stem_fused = SP && e->opts.stem_fused_split && eh == 32;
  if (plan->stem_fused) {
    // straight from the uint8 frames (ops.hip stem0_u8_kernel); `x` is only materialised on demand for rtd_debug_tensor("input")
    const uint8_t** table = (const uint8_t**)B.alloc((size_t)c.max_batch * sizeof(void*));
    plan->frame_table = table;
    if (!B.dry) {
      DevWeight w0 = get_weight(e, "backbone.stem.0", P, eh, 9 * 8);
      const void* wp = w0.w; const float* bp = w0.bias; const int kp = w0.Kpad;
      B.push("backbone.stem.0", "conv_igemm", 2.0 * n * h * w * eh * 27.0, (double)n * H * W * 3 + Builder::tbytes(s0),
             [table, n, H, W, wp, kp, bp, s0](hipStream_t st) { launch_stem0_u8(table, n, H, W, wp, kp, bp, s0, ACT_RELU, st); });
    }
  } else {
    B.conv("backbone.stem.0", x, s0, 3, 2, 1, ACT_RELU, nullptr, RES_NONE, 3);
  }
  Tensor s1 = B.act(P, n, h, w, eh);
  B.conv("backbone.stem.1", s0, s1, 3, 1, 1, ACT_RELU);
  Tensor cur;
  {
    // stem.2 + max-pool (HF:rt_detr_resnet.py:100-113).  f16x3: one pass when the direct kernel takes the conv (stem_pool_fuse) - the
    // 320^2 x 64-channel conv output (210 MB at R50 bs 8) is neither written nor read back
    Tensor s2v;                                    // stem.2's output as a shape (allocated only when the pool runs on its own)
    s2v.dt = P; s2v.n = n; s2v.h = h; s2v.w = w; s2v.c = c.embedding_size; s2v.ld = c.embedding_size; s2v.bstride = (int64_t)h * w * c.embedding_size;
    const int ph = down2(h), pw = down2(w);
    Tensor pv = s2v; pv.h = ph; pv.w = pw; pv.bstride = (int64_t)ph * pw * pv.ld;
    ConvArgs probe;
    probe.opts = &e->conv_opts;
    probe.x = s1; probe.x.p = (void*)16; probe.y = s2v; probe.y.p = (void*)16;
    probe.KH = probe.KW = 3; probe.stride = 1; probe.pad = 1; probe.act = ACT_RELU; probe.w = (const void*)16;
    Tensor pvp = pv; pvp.p = (void*)16;
    Tensor probe1 = probe.x; probe1.n = 1;          // asked for ONE image: every plan of a handle makes the same choice
    ConvArgs probe_one = probe; probe_one.x = probe1; probe_one.y.n = 1;
    Tensor pv1 = pvp; pv1.n = 1;
    // (conv -> pool and the fused pass agree bit for bit, so the side buffers' 2 GiB descriptor limit may decide per batch size)
    const long long pool_tiles = (long long)n * ((w + 31) / 32) * ((h + 7) / 8);
    const bool fuse = SP && e->opts.stem_pool_fuse && conv_pool_supported(probe_one, pv1) && pool_tiles * 8192 < (1ll << 31);
    if (fuse) {
      h = ph; w = pw;
      cur = B.act(P, n, h, w, c.embedding_size, "stem");
      ConvArgs a;
      a.x = s1; a.y = s2v; a.y.p = nullptr;
      a.KH = a.KW = 3; a.stride = 1; a.pad = 1; a.act = ACT_RELU; a.opts = &e->conv_opts;
      DevWeight w2;
      if (!B.dry) w2 = get_weight(e, "backbone.stem.2", P, c.embedding_size, 9 * s1.c);
      else { w2.Kpad = conv_kpad_split(9 * s1.c); w2.Npad = conv_npad(c.embedding_size); }
      a.w = w2.w; a.bias = w2.bias; a.Kpad = w2.Kpad; a.Npad = w2.Npad;
      void* side = B.alloc(conv_pool_side_bytes(a));
      const Tensor curv = cur;
      B.push("backbone.stem.2+pool", "conv_igemm", 2.0 * s2v.pixels() * s2v.c * 9.0 * s1.c, Builder::tbytes(s1) + Builder::tbytes(cur),
             [a, curv, side](hipStream_t st) { launch_conv_pool(a, curv, side, st); });
    } else {
      Tensor s2 = B.act(P, n, h, w, c.embedding_size);
      B.conv("backbone.stem.2", s1, s2, 3, 1, 1, ACT_RELU);
      h = ph; w = pw;
      cur = B.act(P, n, h, w, c.embedding_size, "stem");
      const Tensor curv = cur;
      B.push("backbone.pool", "maxpool", 9.0 * cur.pixels() * cur.c, Builder::tbytes(s2) + Builder::tbytes(cur),
             [s2, curv](hipStream_t s) { launch_maxpool3x3s2(s2, curv, s); });
    }
  }

  // the FPN's concat buffers exist before the backbone runs: their projection halves are filled as soon as a stage's map is complete
  const int d = c.enc_dim, hh = c.csp_hidden;
  const int* lh = e->lvl_h; const int* lw = e->lvl_w;
  Tensor cat1 = B.act(P, n, lh[0], lw[0], 2 * d);   // [up(lat1) | proj0]
  Tensor cat0 = B.act(P, n, lh[1], lw[1], 2 * d);   // [up(lat0) | proj1]
  const bool early_enc_proj = (e->opts.side_stream & 4) != 0;
  bool enc_proj_forked = false;

  // ---- residual stages (HF:rt_detr_resnet.py:135-310) ------------------------------------------
  Tensor feats[3];
  int cin = c.embedding_size;
  bool c1_done = false;                                         // this block's c1 already ran inside the previous block's last conv
  Tensor t1_next;
  bool have_prepooled = false;                                  // the 2 x 2 average of the stage input already exists (avg_fuse)
  Tensor prepooled;
  for (int si = 0; si < 4; ++si) {
    const int cout = c.hidden_sizes[si];
    // Buffers are recycled inside a stage (rtd_debug_option "arena_reuse"): the blocks' outputs ping-pong between two buffers
    // (a block's input is dead once its last conv has read it as the residual) and the c1 / c2 temporaries of every block
    // share one buffer each.  Fewer distinct lines means more of a stage lives in L2 + the 256 MB Infinity Cache, and dead
    // activations are overwritten in cache instead of being written back to HBM.
    Tensor pp[2], tb1, tb2;
    if (e->opts.arena_reuse) {
      const int s0 = (si > 0) ? 2 : 1;
      const int oh0 = s0 == 2 ? down2(h) : h, ow0 = s0 == 2 ? down2(w) : w;
      pp[0] = B.act(P, n, oh0, ow0, cout);
      if (c.depths[si] > 1) pp[1] = B.act(P, n, oh0, ow0, cout);
      const int mid0 = c.layer_type == RTD_LAYER_BOTTLENECK ? cout / 4 : cout;
      tb1 = B.act(P, n, c.layer_type == RTD_LAYER_BOTTLENECK ? h : oh0, c.layer_type == RTD_LAYER_BOTTLENECK ? w : ow0, mid0);   // block 0's c1 runs before the stride
      tb2 = B.act(P, n, oh0, ow0, mid0);
    }
    auto view = [&](const Tensor& buf, int hh_, int ww_, int cc_, const std::string& name) {
      Tensor t = buf;
      t.h = hh_; t.w = ww_; t.c = cc_; t.ld = cc_; t.bstride = (int64_t)hh_ * ww_ * cc_;
      if (!name.empty()) B.plan->named[name] = t;
      return t;
    };
    for (int bi = 0; bi < c.depths[si]; ++bi) {
      const int stride = (si > 0 && bi == 0) ? 2 : 1;
      const std::string pfx = nm("backbone.s%d.b%d", si, bi);
      const int oh = stride == 2 ? down2(h) : h, ow = stride == 2 ? down2(w) : w;
      const bool last = bi == c.depths[si] - 1;
      const std::string oname = (last && si >= 1) ? nm("backbone%d", si - 1) : std::string();
      Tensor res = cur;
      bool has_sc;
      if (c.layer_type == RTD_LAYER_BOTTLENECK) has_sc = (cin != cout) || stride != 1;
      else has_sc = (bi == 0);
      // The projection shortcut is a 1x1 conv over the block input whose only use is the pre-activation add of the block's last
      // conv: y = relu(W_last * t + b_last + (W_sc * x_in + b_sc)).  In bf16 plans it is folded into that conv as extra K
      // (ConvArgs::x2): the [B,OH,OW,cout] shortcut tensor is neither written nor read back (R50 bs 8 stage 0: 2 x 105 MB), one
      // launch less, and the sum is rounded once instead of twice.  fp32 plans keep the reference's op sequence.
      const int mid = cout / 4;
      Tensor sc_in = cur;                                    // what the shortcut's 1x1 reads
      bool fold_sc = false;
      if (has_sc) {
        // stride 2: AvgPool2d(2,2,ceil) then 1x1 (HF:rt_detr_resnet.py:199-213); extents are even here
        if (stride == 2) {
          if (have_prepooled) {
            sc_in = prepooled;                                 // written by the previous stage's last conv (ConvArgs::avg_y)
            have_prepooled = false;
          } else {
            Tensor pooled = B.act(P, n, oh, ow, cin);
            B.push(pfx + ".avgpool", "avgpool", 4.0 * pooled.pixels() * cin, Builder::tbytes(cur) + Builder::tbytes(pooled),
                   [cur, pooled](hipStream_t s) { launch_avgpool2(cur, pooled, s); });
            sc_in = pooled;
          }
        }
        if ((P == BF16 || SP) && e->opts.sc_fold) {
          // shapes only: would the kernels take the folded launch?  Asked for ONE image whatever this plan's batch: every plan
          // of an engine must use the same filters (the host copies are dropped after the first plan) and the same arithmetic
          // (batch invariance), and a single image has the smallest grid
          ConvArgs probe;
          probe.opts = &e->conv_opts;
          Tensor yv; yv.dt = P; yv.n = 1; yv.h = oh; yv.w = ow; yv.c = cout; yv.ld = cout; yv.bstride = (int64_t)oh * ow * cout; yv.p = (void*)16;
          Tensor xv = yv; xv.c = xv.ld = (c.layer_type == RTD_LAYER_BOTTLENECK ? mid : cout); xv.bstride = (int64_t)oh * ow * xv.c;
          Tensor x2v = sc_in; x2v.p = (void*)16; x2v.n = 1;
          probe.x = xv; probe.x2 = x2v; probe.y = yv;
          probe.KH = probe.KW = (c.layer_type == RTD_LAYER_BOTTLENECK ? 1 : 3); probe.stride = 1; probe.pad = probe.KH / 2;
          fold_sc = conv_dual_supported(probe);
        }
        if (!fold_sc) {
          res = B.act(P, n, oh, ow, cout);
          B.conv(pfx + ".sc", sc_in, res, 1, 1, 0, ACT_NONE);
        }
      }
      Tensor out = e->opts.arena_reuse ? view(pp[bi & 1], oh, ow, cout, oname) : B.act(P, n, oh, ow, cout, oname);
      if (c.layer_type == RTD_LAYER_BOTTLENECK) {
        // the c1 output of this block: with recycled buffers it is the stage's shared temporary, which the PREVIOUS block's last conv
        // may already have filled (ConvArgs::next_*: the reduce conv fused into the expand conv that produced its input)
        Tensor t1 = c1_done ? t1_next : (e->opts.arena_reuse ? view(tb1, h, w, mid, "") : B.act(P, n, h, w, mid));
        if (!c1_done) B.conv(pfx + ".c1", cur, t1, 1, 1, 0, ACT_RELU);
        c1_done = false;
        Tensor t2 = e->opts.arena_reuse ? view(tb2, oh, ow, mid, "") : B.act(P, n, oh, ow, mid);
        B.conv(pfx + ".c2", t1, t2, 3, stride, 1, ACT_RELU);
        // fuse the NEXT block's c1 (1x1, stride 1, reads `out` at these extents) when the streaming kernel takes this conv: the next block of
        // this stage, or (f16x3 plans) block 0 of the next stage, whose c1 runs before that block's stride
        const Tensor* nx = nullptr;
        std::string nx_name;
        const bool same_stage = bi + 1 < c.depths[si];
        const bool cross_stage = !same_stage && SP && si + 1 < 4;
        if ((P == BF16 || SP) && e->opts.c1_fuse && (same_stage || cross_stage)) {
          const int mid_n = same_stage ? mid : c.hidden_sizes[si + 1] / 4;
          Tensor t1_shape = out; t1_shape.c = t1_shape.ld = mid_n; t1_shape.bstride = (int64_t)oh * ow * mid_n;
          ConvArgs probe;                                        // this plan's shapes (fused and separate launches are bit-identical)
          probe.opts = &e->conv_opts;
          probe.x = t2; probe.x.p = (void*)16;
          probe.y = out; probe.y.p = (void*)16;
          if (fold_sc) { probe.x2 = sc_in; probe.x2.p = (void*)16; }
          else { probe.res = res; probe.res.p = (void*)16; probe.res_mode = RES_PRE; }
          probe.next_y = t1_shape; probe.next_y.p = (void*)16;
          if (conv_next_supported(probe)) {
            t1_next = (same_stage && e->opts.arena_reuse) ? view(tb1, oh, ow, mid_n, "") : B.act(P, n, oh, ow, mid_n);
            nx = &t1_next; nx_name = (same_stage ? nm("backbone.s%d.b%d", si, bi + 1) : nm("backbone.s%d.b0", si + 1)) + ".c1"; c1_done = true;
          }
        }
        if (SP && e->opts.avg_fuse && last && si + 1 < 4 && !fold_sc && (oh & 1) == 0 && (ow & 1) == 0) {
          // the next stage's vd shortcut reads AvgPool2d(2, 2) of `out`: let this launch write it (shapes for ONE image decide, like every fusion)
          ConvArgs probe;
          probe.opts = &e->conv_opts;
          probe.x = t2; probe.x.p = (void*)16; probe.x.n = 1;
          probe.y = out; probe.y.p = (void*)16; probe.y.n = 1;
          probe.res = res; probe.res.p = (void*)16; probe.res.n = 1; probe.res_mode = RES_PRE;
          if (nx) { probe.next_y = *nx; probe.next_y.p = (void*)16; probe.next_y.n = 1; }
          // ... except the 2 GiB descriptor limits, which depend on the batch: beyond them (R50 / R101 at 1280 px from batch 41 on) this
          // plan keeps the separate avg-pool launch - bit-identical results (test_f16x3_fused_vd_shortcut_average_equals_the_avgpool_launch)
          ConvArgs whole = probe;
          whole.x.n = whole.y.n = whole.res.n = n;
          if (nx) whole.next_y.n = n;
          whole.avg_y = out; whole.avg_y.p = (void*)16; whole.avg_y.n = n; whole.avg_y.h = oh / 2; whole.avg_y.w = ow / 2;
          whole.avg_y.ld = cout; whole.avg_y.bstride = (int64_t)(oh / 2) * (ow / 2) * cout;
          if (conv_avg_supported(probe) && conv_sx_batch_fits(whole)) {
            prepooled = B.act(P, n, oh / 2, ow / 2, cout);
            have_prepooled = true;
            B.avg_pending = true; B.avg_pending_y = prepooled;
            // `out` is then read by nobody but this launch's own fused consumers when (a) the next stage's first reduce conv rides on it
            // (nx: stage 1's block 0 reads its c1 input from the tile) and (b) the stage output is not an encoder feature (stage 0):
            // block 0 of the next stage takes its residual from the shortcut conv over the fused average, never from `out`
            if (e->opts.dead_out && nx && si == 0 && oname.empty()) B.y_dead_pending = true;
          }
        }
        if (fold_sc) B.conv(pfx + ".c3", t2, out, 1, 1, 0, ACT_RELU, nullptr, RES_NONE, 0, &sc_in, pfx + ".sc", 0, nx, nx_name, ACT_RELU);
        else B.conv(pfx + ".c3", t2, out, 1, 1, 0, ACT_RELU, &res, RES_PRE, 0, nullptr, "", 0, nx, nx_name, ACT_RELU);
      } else {
        Tensor t1 = e->opts.arena_reuse ? view(tb1, oh, ow, cout, "") : B.act(P, n, oh, ow, cout);
        B.conv(pfx + ".c1", cur, t1, 3, stride, 1, ACT_RELU);
        if (fold_sc) B.conv(pfx + ".c2", t1, out, 3, 1, 1, ACT_RELU, nullptr, RES_NONE, 0, &sc_in, pfx + ".sc");
        else B.conv(pfx + ".c2", t1, out, 3, 1, 1, ACT_RELU, &res, RES_PRE);
      }
      cur = out; h = oh; w = ow; cin = cout;
    }
    if (si >= 1) feats[si - 1] = cur;
    if (early_enc_proj && (si == 1 || si == 2)) {
      // HF:v2.py:1348-1360 encoder input projection of this level, beside the next stage (side stream)
      B.marker(1); B.lane = 1;
      B.conv(si == 1 ? "enc.proj.0" : "enc.proj.1", cur, (si == 1 ? cat1 : cat0).slice_c(d, d), 1, 1, 0, ACT_NONE);
      B.lane = 0;
      enc_proj_forked = true;
    }
  }

  // ---- hybrid encoder (HF:v2.py:1348-1360 input proj, :1041-1095 AIFI, :1183-1209 FPN/PAN) ------
  Tensor pcat0 = B.act(P, n, lh[1], lw[1], 2 * d);  // [down0 | lat1]
  Tensor pcat1 = B.act(P, n, lh[2], lw[2], 2 * d);  // [down1 | lat0]
  if (!early_enc_proj) {
    B.conv("enc.proj.0", feats[0], cat1.slice_c(d, d), 1, 1, 0, ACT_NONE);
    B.conv("enc.proj.1", feats[1], cat0.slice_c(d, d), 1, 1, 0, ACT_NONE);
  }
  const int L = lh[2] * lw[2];
  Tensor t0 = B.act(F32, n, L, 1, d, "aifi_in");
  {
    Tensor t0v = t0; t0v.h = lh[2]; t0v.w = lw[2];
    B.conv("enc.proj.2", feats[2], t0v, 1, 1, 0, ACT_NONE);
  }
  // AIFI in fp32 (0.5 % of the FLOPs; keeps the only global-mixing layer of the encoder exact)
  Tensor pos;
  pos.p = e->pos_dev; pos.dt = F32; pos.n = 1; pos.h = L; pos.w = 1; pos.c = d; pos.ld = d; pos.bstride = (int64_t)L * d;
  Tensor t2;
  const bool aifi_fused = e->opts.dec_fused && d == 256 && c.enc_heads == 8 && c.enc_ffn <= 1024;
  if (aifi_fused) {
    // ---- fused AIFI: 2 launches of decoder.hip's row kernel (modes 3, 4) instead of 9 ----------------------------
    t2 = B.act(P, n, L, 1, d, "aifi_out");
    Tensor qrows = B.act(F32, n, L, 1, d);
    const int tl = (L + 15) / 16;
    const int tlp = (tl + 1) & ~1;                              // the kernels stride the fragment buffers by an even tile count
    float* kf = (float*)B.alloc((size_t)n * 8 * tlp * 512 * 4);
    float* vf = (float*)B.alloc((size_t)n * 8 * tlp * 512 * 4);
    auto elin = [&](const std::string& name, int N, int K) {
      DecLin Lw{};
      if (!B.dry) {
        DevWeight w = get_weight_packed(e, name, N, K, K, e->opts.dec_split && P != F32);
        Lw.w = (const float*)w.w; Lw.b = w.bias; Lw.ldw = w.Kpad; Lw.N = N; Lw.K = w.K;
      }
      return Lw;
    };
    DecArgs a0{};
    a0.split = P != F32 ? e->opts.dec_split : 0;
    a0.attn_split = e->opts.attn_split & 1;                          // bit 0: AIFI, bit 1: decoder
    a0.B = n; a0.Q = L; a0.D = d; a0.heads = 8; a0.S = 0; a0.n_levels = 3; a0.n_points = 4; a0.ffn = c.enc_ffn; a0.C = 4;
    a0.hs_in = (const float*)t0.p; a0.qpos_in = e->pos_dev;
    a0.q_in = (const float*)qrows.p; a0.q_out = (float*)qrows.p;
    a0.kfrag_in = kf; a0.vfrag_in = vf; a0.kfrag_out = kf; a0.vfrag_out = vf;
    DecArgs a3 = a0;
    a3.mode = 3;
    a3.qk = elin("enc.aifi.qk", 2 * d, d); a3.v = elin("enc.aifi.v", d, d);
    B.push("enc.aifi.qkv", "dec_layer", 2.0 * n * L * 3.0 * d * d, (double)n * L * d * 4 * 5, [a3](hipStream_t s) { launch_dec_layer(a3, s); });
    DecArgs a4 = a0;
    a4.mode = 4;
    a4.o = elin("enc.aifi.o", d, d); a4.fc1 = elin("enc.aifi.fc1", c.enc_ffn, d); a4.fc2 = elin("enc.aifi.fc2", d, c.enc_ffn);
    if (!B.dry) {
      a4.ln1.g = get_vec(e, "enc.aifi.ln1.g", d); a4.ln1.b = get_vec(e, "enc.aifi.ln1.b", d);
      a4.ln3.g = get_vec(e, "enc.aifi.ln2.g", d); a4.ln3.b = get_vec(e, "enc.aifi.ln2.b", d);
    }
    Tensor t2f;
    if (SP) { t2f = B.act(F32, n, L, 1, d); a4.hs_out = (float*)t2f.p; }
    else if (P == BF16) a4.out_bf16 = t2.p;
    else a4.hs_out = (float*)t2.p;
    B.push("enc.aifi.layer", "dec_layer", 4.0 * n * (double)L * L * d + 2.0 * n * L * ((double)d * d + 2.0 * d * c.enc_ffn),
           (double)n * L * d * 4 * 4, [a4](hipStream_t s) { launch_dec_layer(a4, s); });
    if (SP) {
      const Tensor src = t2f, dst = t2;
      const int64_t rows = src.pixels();
      B.push("enc.aifi.out_split", "convert", 0.0, 2 * Builder::tbytes(src), [src, dst, rows](hipStream_t s) {
        launch_f32_to_split((const float*)src.p, src.ld, dst.p, dst.ld, rows, src.c, s);
      });
    }
  } else {
  Tensor xp = B.act(F32, n, L, 1, d);
  B.push("enc.aifi.addpos", "add", (double)xp.pixels() * d, 3 * Builder::tbytes(xp), [t0, pos, xp](hipStream_t s) { launch_add(t0, pos, xp, s); });
  // f16x3 plans run the five linears on the pair kernels (inputs converted to hi / lo rows, fp32 rows out; fc1 hands fc2 a pair tensor):
  // on fp32 MFMAs they were 0.33 ms of R101 1280's step.  The attention itself stays exact fp32.
  const bool aifi_pair = SP && e->opts.aifi_pair && d % SPLIT_GROUP == 0 && c.enc_ffn % SPLIT_GROUP == 0;
  auto pin = [&](const char* nm_, const Tensor& x) { return aifi_pair ? B.to_split(std::string(nm_) + ".in_split", x) : x; };
  Tensor qk = B.linear("enc.aifi.qk", pin("enc.aifi.qk", xp), 2 * d, F32, ACT_NONE);
  Tensor vv = B.linear("enc.aifi.v", pin("enc.aifi.v", t0), d, F32, ACT_NONE);
  Tensor att = B.act(F32, n, L, 1, d);
  {
    const int heads = c.enc_heads;
    B.push("enc.aifi.attn", "attention", 4.0 * n * (double)L * L * d, Builder::tbytes(qk) + 2 * Builder::tbytes(vv),
           [qk, vv, att, heads](hipStream_t s) { launch_attention(qk, vv, att, heads, s); });
  }
  Tensor ao = B.linear("enc.aifi.o", pin("enc.aifi.o", att), d, F32, ACT_NONE, &t0);
  Tensor t1 = B.layernorm("enc.aifi.ln1", ao, F32);
  Tensor f1 = B.linear("enc.aifi.fc1", pin("enc.aifi.fc1", t1), c.enc_ffn, aifi_pair ? F16X2 : F32, ACT_GELU);
  Tensor f2 = B.linear("enc.aifi.fc2", f1, d, F32, ACT_NONE, &t1);
  t2 = B.layernorm("enc.aifi.ln2", f2, P, "aifi_out");
  }
  t2.h = lh[2]; t2.w = lw[2];

  // `up_src`: the half-resolution tensor whose 2x nearest upsampling is the first half of `cat`.  In bf16 plans the CSP's first
  // 1x1 conv reads it directly (ConvArgs::x_up2 + x2 = the second half of cat): no upsample launch, no upsampled tensor.
  auto csp = [&](const std::string& pfx, const Tensor& cat, const std::string& oname, const Tensor* up_src = nullptr) {
    Tensor h12 = B.act(P, cat.n, cat.h, cat.w, 2 * hh);
    bool up_fold = false;
    if (up_src && (P == BF16 || SP) && e->opts.up_fold) {
      ConvArgs probe;                                          // shapes for ONE image, like the shortcut fold
      probe.opts = &e->conv_opts;
      probe.x = *up_src; probe.x.p = (void*)16; probe.x.n = 1;
      probe.x2 = cat.slice_c(d, d); probe.x2.p = (void*)16; probe.x2.n = 1;
      probe.y = h12; probe.y.p = (void*)16; probe.y.n = 1;
      probe.x_up2 = 1;
      up_fold = conv_dual_supported(probe);
    }
    if (up_fold) {
      Tensor second = cat.slice_c(d, d);
      B.conv(pfx + ".c12", *up_src, h12, 1, 1, 0, ACT_SILU, nullptr, RES_NONE, 0, &second, "", 1);
    } else {
      if (up_src) {
        const Tensor src = *up_src, dst = cat.slice_c(0, d);
        B.push(pfx + ".up", "upsample2x", 0.0, Builder::tbytes(src) + 4 * Builder::tbytes(src), [src, dst](hipStream_t s) { launch_upsample2x(src, dst, s); });
      }
      B.conv(pfx + ".c12", cat, h12, 1, 1, 0, ACT_SILU);
    }
    Tensor r0 = B.act(P, cat.n, cat.h, cat.w, hh);
    B.conv(pfx + ".rep0", h12.slice_c(0, hh), r0, 3, 1, 1, ACT_SILU);
    Tensor r1 = B.act(P, cat.n, cat.h, cat.w, hh);
    B.conv(pfx + ".rep1", r0, r1, 3, 1, 1, ACT_SILU);
    Tensor h2 = h12.slice_c(hh, hh);
    if (hh == d) {
      Tensor r2 = B.act(P, cat.n, cat.h, cat.w, hh, oname);
      B.conv(pfx + ".rep2", r1, r2, 3, 1, 1, ACT_SILU, &h2, RES_POST);
      return r2;
    }
    Tensor r2 = B.act(P, cat.n, cat.h, cat.w, hh);
    B.conv(pfx + ".rep2", r1, r2, 3, 1, 1, ACT_SILU, &h2, RES_POST);
    Tensor o = B.act(P, cat.n, cat.h, cat.w, d, oname);
    B.conv(pfx + ".c3", r2, o, 1, 1, 0, ACT_SILU);
    return o;
  };
  // FPN top-down
  Tensor lat0 = pcat1.slice_c(d, d);
  B.conv("enc.lat.0", t2, lat0, 1, 1, 0, ACT_SILU);
  if (enc_proj_forked) B.marker(2);                  // the projection halves of cat0 / cat1 are complete
  Tensor F0 = csp("enc.fpn.0", cat0, "", &lat0);
  Tensor lat1 = pcat0.slice_c(d, d);
  B.conv("enc.lat.1", F0, lat1, 1, 1, 0, ACT_SILU);
  Tensor F1 = csp("enc.fpn.1", cat1, "enc0", &lat1);
  // ---- decoder input (HF:v2.py:1533-1623): the projections of the two larger levels only need F1 / P1, so (side_stream bit 1) they run on the
  // side stream beside the PAN path, whose 40^2 and 20^2 grids leave a quarter to two thirds of the CUs idle
  const int dm = c.d_model, S = e->S, Q = c.num_queries, C = c.num_classes, NL = c.dec_layers;
  Tensor mem = B.act(P, n, S, 1, dm, "memory");
  const bool early_proj = (e->opts.side_stream & 2) != 0;
  auto dec_proj = [&](int l, const Tensor& src) {
    Tensor v = mem;
    v.p = B.dry ? nullptr : (char*)mem.p + (size_t)e->lvl_start[l] * dm * dtype_size(P);
    v.h = lh[l]; v.w = lw[l];
    B.conv(nm("dec.proj.%d", l), src, v, 1, 1, 0, ACT_NONE);
  };
  // PAN bottom-up
  if (early_proj) { B.marker(1); B.lane = 1; dec_proj(0, F1); B.lane = 0; }
  B.conv("enc.down.0", F1, pcat0.slice_c(0, d), 3, 2, 1, ACT_SILU);
  Tensor P1 = csp("enc.pan.0", pcat0, "enc1");
  if (early_proj) { B.marker(1); B.lane = 1; dec_proj(1, P1); B.lane = 0; }
  B.conv("enc.down.1", P1, pcat1.slice_c(0, d), 3, 2, 1, ACT_SILU);
  Tensor P2 = csp("enc.pan.1", pcat1, "enc2");
  if (!early_proj) { dec_proj(0, F1); dec_proj(1, P1); }
  dec_proj(2, P2);
  if (early_proj) B.marker(2);
  // The query-selection chain (enc_output -> scores -> top-k -> gather: narrow grids, latency-bound) and the value projection both start
  // from `mem` and meet again in the decoder prologue: the chain runs on the side stream beside the projection (e->opts.side_stream).
  const bool side = (e->opts.side_stream & 1) != 0;
  if (side) { B.marker(1); B.lane = 1; }
  // enc_output on masked memory, fp32 from here on (selection + decoder are exact fp32)
  Tensor eo = B.linear("dec.enc_out.fc", mem, dm, F32, ACT_NONE);
  if (!B.dry && e->n_invalid > 0) {
    const float* bias = get_weight(e, "dec.enc_out.fc", P, dm, dm).bias;
    const int32_t* rows = e->invalid_rows_dev;
    const int nr = e->n_invalid;
    B.push("dec.mask_rows", "set_rows", 0.0, (double)n * nr * dm * 4, [eo, rows, nr, S, bias](hipStream_t s) { launch_set_rows(eo, rows, nr, S, bias, s); });
  }
  const bool fused = e->opts.dec_fused && dm == 256 && c.dec_heads == 8 && c.dec_ffn <= 1024 && C <= 512 && c.n_levels == 3 && c.n_points == 4;
  const bool sel_fused = fused && e->opts.sel_fused && eo.ld == dm;
  float* mx = (float*)B.alloc((size_t)n * S * 4);
  {
    Tensor t; t.p = mx; t.dt = F32; t.n = n; t.h = S; t.w = 1; t.c = 1; t.ld = 1; t.bstride = S;
    plan->named["enc_cls_max"] = t;
  }
  Tensor om;
  DecLN sel_ln{};
  if (sel_fused) {
    // LayerNorm + enc_score_head + class max in one launch: the normalised memory (69 MB fp32 at R50 bs 8) and the logits are
    // never written; the selected rows are normalised again in the gather
    SelArgs sa{};
    if (!B.dry) {
      DevWeight w = get_weight_packed(e, "dec.enc_score", C, dm, dm, false);
      sa.score.w = (const float*)w.w; sa.score.b = w.bias; sa.score.ldw = w.Kpad; sa.score.N = C; sa.score.K = w.K;
      sel_ln.g = get_vec(e, "dec.enc_out.ln.g", dm); sel_ln.b = get_vec(e, "dec.enc_out.ln.b", dm);
      sa.ln = sel_ln;
    }
    sa.x = (const float*)eo.p; sa.ldx = eo.ld; sa.rows = n * S; sa.C = C; sa.rows_per_image = S; sa.mx = mx;
    B.push("dec.select_score", "select_score", (double)n * S * (2.0 * C * dm + 8.0 * dm), (double)n * S * dm * 4 + (double)n * S * 4,
           [sa](hipStream_t s) { launch_select_score(sa, s); });
  } else {
    om = B.layernorm("dec.enc_out.ln", eo, F32, "output_memory");
    Tensor cls = B.linear("dec.enc_score", om, C, F32, ACT_NONE);
    B.push("dec.enc_rowmax", "rowmax", (double)n * S * C, (double)n * S * C * 4, [cls, mx](hipStream_t s) { launch_rowmax(cls, mx, s); });
  }
  int32_t* tk = (int32_t*)B.alloc((size_t)n * Q * 4);
  plan->tk_idx = tk;
  {
    Tensor t; t.p = tk; t.dt = I32; t.n = n; t.h = Q; t.w = 1; t.c = 1; t.ld = 1; t.bstride = Q;
    plan->named["topk"] = t;          // the memory-token ids the decoder ran on (after rtd_debug_force_topk, the forced ones)
  }
  {
    const int32_t* forced = e->forced_idx; const int32_t* flag = e->force_flag;
    B.push("dec.enc_topk", "topk", 0.0, (double)n * S * 4 * 6, [mx, n, S, Q, tk](hipStream_t s) { launch_topk(mx, n, S, Q, tk, nullptr, s); });
    // test hook (rtd_debug_force_topk): overrides the selection with the caller's indices.  Not part of the product graph: the op is
    // skipped - and absent from the built hipGraph - until the hook is used on this handle
    B.push("dec.force_topk", "select", 0.0, 0.0, [tk, forced, flag, n, Q](hipStream_t s) {
      launch_force_idx(tk, forced, flag, n * Q, s);
    });
    if (!B.dry) plan->ops.back().debug_only = true;
  }
  Tensor target = B.act(F32, n, Q, 1, dm, "target");
  if (sel_fused) {
    const float* xp = (const float*)eo.p; const int64_t ldx = eo.ld; float* tp = (float*)target.p; const int64_t ldt = target.ld;
    B.push("dec.gather_target", "gather", 8.0 * n * Q * dm, 2.0 * n * Q * dm * 4, [xp, ldx, S, tk, n, Q, sel_ln, tp, ldt](hipStream_t s) {
      launch_gather_ln(xp, ldx, S, tk, n, Q, sel_ln, tp, ldt, s);
    });
  } else {
    B.push("dec.gather_target", "gather", 0.0, 2.0 * n * Q * dm * 4, [om, tk, S, target](hipStream_t s) { launch_gather_rows(om, tk, S, target, s); });
  }
  B.lane = 0;
  // value_proj of every decoder layer in ONE GEMM (they all read `mem`, HF:v2.py:177)
  Tensor vall = B.linear("dec.vp_all", mem, NL * dm, SP ? F32 : P, ACT_NONE, nullptr, "value_all");   // the samplers read bf16 or fp32 values
  // the join sits after the decoder prologue in the fused plan (the prologue needs the selected rows, not the value maps: it stays on the
  // side stream and the chain's 150 + 45 us run beside the projection's 175 instead of 45 after it)
  bool side_joined = !side;
  const int npts = c.dec_heads * c.n_levels * c.n_points;
  float* ref_unact8 = (float*)B.alloc((size_t)n * Q * 8 * 4);
  float* ref8 = (float*)B.alloc((size_t)n * Q * 8 * 4);
  Tensor ref8t; ref8t.p = ref8; ref8t.dt = F32; ref8t.n = n; ref8t.h = Q; ref8t.w = 1; ref8t.c = 8; ref8t.ld = 8; ref8t.bstride = (int64_t)Q * 8;
  plan->named["ref"] = ref8t;
  {
    Tensor ru = ref8t; ru.p = ref_unact8;
    plan->named["ref_unact"] = ru;
  }
  Tensor hs = target;
  Tensor logits;
  if (!fused && !side_joined) { B.marker(2); side_joined = true; }
  if (fused) {
    // ---- fused decoder: 1 prologue + per layer (self-attention kernel + one fused kernel), decoder.hip -----
    Tensor qpos = B.act(F32, n, Q, 1, dm);
    Tensor qrows = B.act(F32, n, Q, 1, dm);
    const int dtiles = (Q + 15) / 16;
    float* kfrag[2]; float* vfrag[2];
    for (int i = 0; i < 2; ++i) {
      kfrag[i] = (float*)B.alloc((size_t)n * c.dec_heads * ((dtiles + 1) & ~1) * 512 * 4);
      vfrag[i] = (float*)B.alloc((size_t)n * c.dec_heads * ((dtiles + 1) & ~1) * 512 * 4);
    }
    logits = B.act(F32, n, Q, 1, C, "logits");
    auto lin = [&](const std::string& name, int N, int K, int Kuse = 0) {
      DecLin L{};
      if (!B.dry) {
        DevWeight w = get_weight_packed(e, name, N, K, Kuse ? Kuse : K, e->opts.dec_split && P != F32);
        L.w = (const float*)w.w; L.b = w.bias; L.ldw = w.Kpad; L.N = N; L.K = w.K;
      }
      return L;
    };
    auto lnp = [&](const std::string& name) {
      DecLN P{};
      if (!B.dry) { P.g = get_vec(e, name + ".g", dm); P.b = get_vec(e, name + ".b", dm); }
      return P;
    };
    DecArgs base{};
    base.split = P != F32 ? e->opts.dec_split : 0;
    base.attn_split = (e->opts.attn_split >> 1) & 1;
    base.B = n; base.Q = Q; base.D = dm; base.heads = c.dec_heads; base.S = S; base.n_levels = c.n_levels;
    base.n_points = c.n_points; base.ffn = c.dec_ffn; base.C = C; base.offset_scale = c.offset_scale;
    base.ref8 = ref8; base.ref_unact8 = ref_unact8; base.anchors = e->anchors_dev; base.tk_idx = tk;
    base.value = vall.p; base.value_ld = (int)vall.ld; base.value_f32 = vall.dt == F32; base.lvl = e->lvl_dev;
    base.qpos_in = (const float*)qpos.p; base.qpos_out = (float*)qpos.p;
    base.q_in = (const float*)qrows.p; base.q_out = (float*)qrows.p;
    base.logits = (float*)logits.p;
    base.qp0 = lin("dec.qpos.0", 2 * dm, 8, 64);   // K padded to one 64-wide step (zero weights / zero LDS columns)
    base.qp1 = lin("dec.qpos.1", dm, 2 * dm);
    const double row_flops_next = 2.0 * n * Q * ((double)8 * 2 * dm + 2.0 * dm * dm + 2.0 * dm * dm + (double)dm * dm);
    {
      DecArgs a = base;
      a.mode = 0;
      a.hs_in = (const float*)target.p; a.hs_out = nullptr;
      a.bb0 = lin("dec.enc_bbox.0", dm, dm); a.bb1 = lin("dec.enc_bbox.1", dm, dm); a.bb2 = lin("dec.enc_bbox.2", 4, dm);
      a.qk = lin("dec.l0.sa.qk", 2 * dm, dm); a.v = lin("dec.l0.sa.v", dm, dm);
      a.kfrag_out = kfrag[0]; a.vfrag_out = vfrag[0];
      if (!side_joined) B.lane = 1;
      B.push("dec.prologue", "dec_layer", 2.0 * n * Q * (2.0 * dm * dm + 4.0 * dm) + row_flops_next, (double)n * Q * dm * 4 * 6,
             [a](hipStream_t s) { launch_dec_layer(a, s); });
      B.lane = 0;
      if (!side_joined) { B.marker(2); side_joined = true; }
    }
    for (int i = 0; i < NL; ++i) {
      const std::string p = nm("dec.l%d", i);
      Tensor hs_out = B.act(F32, n, Q, 1, dm, nm("dec%d.hs", i));
      DecArgs a = base;
      const bool last = i == NL - 1;
      a.mode = last ? 2 : 1;
      a.hs_in = (const float*)hs.p; a.hs_out = (float*)hs_out.p;
      a.value_coff = i * dm;
      a.kfrag_in = kfrag[i & 1]; a.vfrag_in = vfrag[i & 1];
      a.kfrag_out = kfrag[(i + 1) & 1]; a.vfrag_out = vfrag[(i + 1) & 1];
      a.o = lin(p + ".sa.o", dm, dm); a.ln1 = lnp(p + ".ln1");
      a.offaw = lin(p + ".ca.offaw", 3 * npts, dm); a.op = lin(p + ".ca.op", dm, dm); a.ln2 = lnp(p + ".ln2");
      a.fc1 = lin(p + ".fc1", c.dec_ffn, dm); a.fc2 = lin(p + ".fc2", dm, c.dec_ffn); a.ln3 = lnp(p + ".ln3");
      a.bb0 = lin(nm("dec.bbox.%d.0", i), dm, dm); a.bb1 = lin(nm("dec.bbox.%d.1", i), dm, dm); a.bb2 = lin(nm("dec.bbox.%d.2", i), 4, dm);
      if (last) a.cls = lin("dec.cls", C, dm);
      else { a.qk = lin(nm("dec.l%d.sa.qk", i + 1), 2 * dm, dm); a.v = lin(nm("dec.l%d.sa.v", i + 1), dm, dm); }
      a.probe = e->opts.dec_stamps >> 1;                           // diagnostic (timing only): bits 1 / 2 of "dec_stamps"
      if ((e->opts.dec_stamps & 1) && i == std::min(2, NL - 1)) {
        const int blocks = n * ((Q + 15) / 16);
        Tensor st = B.act(F32, 1, blocks, 1, 16, "dec_stamps");
        a.stamps = (float*)st.p;
      }
      const double fl = 4.0 * n * (double)Q * Q * dm + 2.0 * n * Q * ((double)dm * dm * 2 + 3.0 * npts * dm + 2.0 * dm * c.dec_ffn + 2.0 * dm * dm + 4.0 * dm) +
                        2.0 * n * Q * dm * c.n_levels * c.n_points * 4 + (last ? 2.0 * n * Q * dm * C : row_flops_next);
      B.push(p + ".fused", "dec_layer", fl, (double)n * Q * dm * 4 * 8, [a](hipStream_t s) { launch_dec_layer(a, s); });
      hs = hs_out;
    }
  } else {
  Tensor b0 = B.linear("dec.enc_bbox.0", target, dm, F32, ACT_RELU);
  Tensor b1 = B.linear("dec.enc_bbox.1", b0, dm, F32, ACT_RELU);
  Tensor b2 = B.linear("dec.enc_bbox.2", b1, 4, F32, ACT_NONE);
  {
    const float* anchors = e->anchors_dev;
    B.push("dec.ref_init", "ref_init", 0.0, (double)n * Q * 64, [b2, anchors, tk, S, ref_unact8, ref8](hipStream_t s) { launch_ref_init(b2, anchors, tk, S, ref_unact8, ref8, s); });
  }

  // ---- decoder layers (HF:v2.py:603-661, layer :339-431) ---------------------------------------
  for (int i = 0; i < NL; ++i) {
    const std::string p = nm("dec.l%d", i);
    Tensor qp0 = B.linear("dec.qpos.0", ref8t, 2 * dm, F32, ACT_RELU);
    Tensor qpos = B.linear("dec.qpos.1", qp0, dm, F32, ACT_NONE);
    Tensor hp = B.act(F32, n, Q, 1, dm);
    B.push(p + ".addpos1", "add", (double)n * Q * dm, 3.0 * n * Q * dm * 4, [hs, qpos, hp](hipStream_t s) { launch_add(hs, qpos, hp, s); });
    Tensor sqk = B.linear(p + ".sa.qk", hp, 2 * dm, F32, ACT_NONE);
    Tensor sv = B.linear(p + ".sa.v", hs, dm, F32, ACT_NONE);
    Tensor sa = B.act(F32, n, Q, 1, dm);
    {
      const int heads = c.dec_heads;
      B.push(p + ".sa.attn", "attention", 4.0 * n * (double)Q * Q * dm, Builder::tbytes(sqk) + 2 * Builder::tbytes(sv),
             [sqk, sv, sa, heads](hipStream_t s) { launch_attention(sqk, sv, sa, heads, s); });
    }
    Tensor so = B.linear(p + ".sa.o", sa, dm, F32, ACT_NONE, &hs);
    Tensor hs1 = B.layernorm(p + ".ln1", so, F32);
    Tensor hp2 = B.act(F32, n, Q, 1, dm);
    B.push(p + ".addpos2", "add", (double)n * Q * dm, 3.0 * n * Q * dm * 4, [hs1, qpos, hp2](hipStream_t s) { launch_add(hs1, qpos, hp2, s); });
    Tensor offaw = B.linear(p + ".ca.offaw", hp2, 3 * npts, F32, ACT_NONE);
    Tensor samp = B.act(F32, n, Q, 1, dm);
    {
      const int heads = c.dec_heads, hd = dm / c.dec_heads, nl = c.n_levels, np = c.n_points, coff = i * dm;
      const int32_t* lvl = e->lvl_dev;
      const float osc = c.offset_scale;
      B.push(p + ".ca.sample", "msdeform", 2.0 * n * Q * dm * nl * np * 4, (double)n * Q * heads * nl * np * 4 * hd * dtype_size(P),
             [vall, coff, offaw, ref8, samp, heads, hd, nl, np, lvl, osc](hipStream_t s) {
               launch_msdeform(vall, coff, offaw, ref8, samp, heads, hd, nl, np, lvl, osc, s);
             });
    }
    Tensor co = B.linear(p + ".ca.op", samp, dm, F32, ACT_NONE, &hs1);
    Tensor hs2 = B.layernorm(p + ".ln2", co, F32);
    Tensor g1 = B.linear(p + ".fc1", hs2, c.dec_ffn, F32, ACT_RELU);
    Tensor g2 = B.linear(p + ".fc2", g1, dm, F32, ACT_NONE, &hs2);
    Tensor hs3 = B.layernorm(p + ".ln3", g2, F32, nm("dec%d.hs", i));
    Tensor d0 = B.linear(nm("dec.bbox.%d.0", i), hs3, dm, F32, ACT_RELU);
    Tensor d1 = B.linear(nm("dec.bbox.%d.1", i), d0, dm, F32, ACT_RELU);
    Tensor d2 = B.linear(nm("dec.bbox.%d.2", i), d1, 4, F32, ACT_NONE);
    B.push(p + ".refine", "box_refine", 0.0, (double)n * Q * 48, [d2, ref8](hipStream_t s) { launch_box_refine(d2, ref8, s); });
    hs = hs3;
  }
    logits = B.linear("dec.cls", hs, C, F32, ACT_NONE, nullptr, "logits");
  }
  // ---- heads + post-processor (HF:v2.py:1880-1881; image_processing_rt_detr.py:510-533) ---------
  float* scores = (float*)B.alloc((size_t)n * Q * C * 4);
  float* topv = (float*)B.alloc((size_t)n * Q * 4);
  int32_t* topi = (int32_t*)B.alloc((size_t)n * Q * 4);
  plan->block6 = (float*)B.alloc((size_t)n * Q * 6 * 4);
  plan->scale_wh = (float*)B.alloc((size_t)n * 2 * 4);
  {
    float* block6 = plan->block6; float* scale = plan->scale_wh;
    // one launch when the shape allows (every decoder here: dense [n * Q, C] fp32 logits, Q * C <= 32768), else sigmoid -> top-k -> gather
    const bool fused_ok = e->opts.post_fused && logits.dt == F32 && logits.ld == C && (int64_t)Q * C <= 32768 && Q <= 1024 && logits.pixels() == (int64_t)n * Q;
    if (fused_ok) {
      B.push("post.fused", "topk", (double)n * Q * C, (double)n * Q * C * 4 + (double)n * Q * 64, [logits, ref8, scale, n, Q, block6](hipStream_t s) {
        RTD_CHECK(launch_postprocess_fused(logits, ref8, scale, n, Q, block6, s), 1, "post-processor: fused launch refused a shape the plan accepted");
      });
    } else {
    B.push("post.sigmoid", "postprocess", (double)n * Q * C, 2.0 * n * Q * C * 4, [logits, scores](hipStream_t s) { launch_postprocess_scores(logits, scores, s); });
    B.push("post.topk", "topk", 0.0, (double)n * Q * C * 4 * 6, [scores, n, Q, C, topi, topv](hipStream_t s) { launch_topk(scores, n, Q * C, Q, topi, topv, s); });
    B.push("post.gather", "postprocess", 0.0, (double)n * Q * 64, [topv, topi, ref8, scale, n, Q, C, block6](hipStream_t s) {
      launch_postprocess_gather(topv, topi, ref8, scale, n, Q, C, block6, s);
    });
    }
  }
  B.finish_workspace();
}

Plan* get_plan(rtd_engine* e, int n) {
  auto it = e->plans.find(n);
  if (it != e->plans.end()) return it->second.get();
  std::unique_ptr<Plan> plan(new Plan());
  plan->n = n;
  Builder dry{e, plan.get(), true};
  build_graph(e, dry, n);
  plan->arena_bytes = dry.off + 4096;
  HIP_CHECK(hipMalloc(&plan->arena, plan->arena_bytes));
  e->allocs.push_back(plan->arena);
  HIP_CHECK(hipMemsetAsync(plan->arena, 0, plan->arena_bytes, e->stream));
  plan->named.clear();
  Builder real{e, plan.get(), false};
  build_graph(e, real, n);
  HIP_CHECK(hipStreamSynchronize(e->stream));
  Plan* p = plan.get();
  e->plans[n] = std::move(plan);
  e->st_plans++;
  return p;
}

// one op of the plan on its lane's stream (eager pass); fork / join markers become event edges between the handle's two streams
void run_op(rtd_engine* e, Op& op) {
  if (op.debug_only && !e->force_used) return;
  if (op.kind == 1 || op.kind == 2) {
    hipEvent_t ev = op.kind == 1 ? e->ev_fork : e->ev_join;
    hipStream_t from = op.kind == 1 ? e->stream : e->side, to = op.kind == 1 ? e->side : e->stream;
    HIP_CHECK(hipEventRecord(ev, from));
    HIP_CHECK(hipStreamWaitEvent(to, ev, 0));
  } else {
    op.run(op.lane == 1 ? e->side : e->stream);
  }
}

// The plan's hipGraph, built node by node: every launcher runs once with a GraphBuild active on this thread, so its rtd_launch calls add
// kernel nodes (parameters copied at that moment) instead of enqueuing work; a lane's nodes form a chain, fork / join markers add the
// cross-lane edges.  Host-only: no stream is touched, nothing runs, no stream capture and no event is involved - rounds 2 / 3 captured
// this graph from the handle's streams and met hipErrorCapturedEvent / hipErrorStreamCaptureUnsupported in torch calls of other
// threads of the process; with no capture anywhere in the library those states cannot exist.
void build_exec(rtd_engine* e, Plan* p, long long eager_launches) {
  GraphBuild gb;
  HIP_CHECK(hipGraphCreate(&gb.graph, 0));
  gb.lane_stream[0] = e->stream;
  gb.lane_stream[1] = e->side;
  t_graph_build = &gb;
  try {
    for (auto& op : p->ops) {
      if (op.debug_only && !e->force_used) continue;
      if (op.kind == 1) gb.fork();
      else if (op.kind == 2) gb.join();
      else op.run(op.lane == 1 ? e->side : e->stream);
    }
  } catch (...) {
    t_graph_build = nullptr;
    (void)hipGraphDestroy(gb.graph);
    throw;
  }
  t_graph_build = nullptr;
  hipGraphExec_t exec = nullptr;
  hipError_t er = gb.nodes == eager_launches ? hipGraphInstantiate(&exec, gb.graph, nullptr, nullptr, 0) : hipErrorInvalidValue;
  if (er != hipSuccess) {
    (void)hipGraphDestroy(gb.graph);
    RTD_CHECK(gb.nodes == eager_launches, RTD_E_HIP, "graph build: " + std::to_string(gb.nodes) + " nodes for " + std::to_string(eager_launches) +
                                                          " launches of the eager pass (a launcher bypasses rtd_launch)");
    HIP_CHECK(er);
  }
  p->graph = gb.graph;
  p->exec = exec;
  e->st_graphs++;
  e->st_graph_nodes += gb.nodes;
}

// eager pass of a plan that has no graph yet (faults and shape errors surface here, with RTD_TRACE_OPS=1: name + sync per op), then its graph
void warm_and_build(rtd_engine* e, Plan* p) {
  const bool trace = getenv("RTD_TRACE_OPS") != nullptr;
  const long long l0 = t_launches;
  for (auto& op : p->ops) {
    if (op.debug_only && !e->force_used) continue;
    if (trace) { fprintf(stderr, "[rtd] %s (%s)\n", op.name.c_str(), op.kernel); fflush(stderr); }
    run_op(e, op);
    if (trace) { HIP_CHECK(hipStreamSynchronize(e->side)); HIP_CHECK(hipStreamSynchronize(e->stream)); }
  }
  const long long launches = t_launches - l0;
  HIP_CHECK(hipStreamSynchronize(e->side));
  HIP_CHECK(hipStreamSynchronize(e->stream));
  e->st_eager++;
  build_exec(e, p, launches);
}

void run_plan(rtd_engine* e, Plan* p) {
  if (e->cfg.use_graph) {
    if (!p->exec) warm_and_build(e, p);      // (the eager pass above already produced this call's results once; the replay repeats them)
    HIP_CHECK(hipGraphLaunch(p->exec, e->stream));
    e->st_graph_launches++;
  } else {
    for (auto& op : p->ops) run_op(e, op);
    e->st_eager++;
  }
}

// preprocess n frames into plan->input and set the post-processor's (w,h) scale
void enqueue_frames(rtd_engine* e, Plan* p, int n, const uint8_t* const* frames, const int32_t* hw, bool on_device, bool via_pinned = false) {
  const int H = e->cfg.input_h, W = e->cfg.input_w;
  size_t total = 0, max_tmp = 0;
  bool any_resize = false;
  for (int i = 0; i < n; ++i) {
    RTD_CHECK(frames[i] != nullptr && hw[2 * i] > 0 && hw[2 * i + 1] > 0 && hw[2 * i] <= 16384 && hw[2 * i + 1] <= 16384,
              RTD_E_INVALID, "frame pointer / size");
    total += ((size_t)hw[2 * i] * hw[2 * i + 1] * 3 + 255) / 256 * 256;
    if (hw[2 * i] != H || hw[2 * i + 1] != W) any_resize = true;
  }
  // one frame of another size sends EVERY frame of the batch through the resampler (identity-sized ones with 1-tap
  // coefficients): the horizontal pass' intermediate [src_h][W][3] must fit the largest of them, not only the resized ones
  if (any_resize)
    for (int i = 0; i < n; ++i) max_tmp = std::max(max_tmp, (size_t)hw[2 * i] * W * 3);   // (the fused uint8 stem resamples only the odd-sized ones)
  if (!on_device && total > e->frame_stage_bytes) {
    HIP_CHECK(hipStreamSynchronize(e->stream));
    if (e->frame_stage) (void)hipFree(e->frame_stage);
    e->frame_stage = nullptr; e->frame_stage_bytes = 0;
    HIP_CHECK(hipMalloc((void**)&e->frame_stage, total));
    e->frame_stage_bytes = total;
  }
  if (!on_device && via_pinned && total > e->pin_stage_bytes) {
    HIP_CHECK(hipStreamSynchronize(e->stream));
    if (e->pin_stage) (void)hipHostFree(e->pin_stage);
    e->pin_stage = nullptr; e->pin_stage_bytes = 0;
    HIP_CHECK(hipHostMalloc((void**)&e->pin_stage, total, hipHostMallocDefault));
    e->pin_stage_bytes = total;
  }
  if (max_tmp > e->resize_tmp_bytes) {
    HIP_CHECK(hipStreamSynchronize(e->stream));
    if (e->resize_tmp) (void)hipFree(e->resize_tmp);
    e->resize_tmp = nullptr; e->resize_tmp_bytes = 0;
    HIP_CHECK(hipMalloc((void**)&e->resize_tmp, max_tmp));
    e->resize_tmp_bytes = max_tmp;
  }
  size_t off = 0;
  bool all_identity = true;
  FrameArgs fa;
  memset(&fa, 0, sizeof fa);
  fa.n = n;
  for (int i = 0; i < n; ++i) {
    const size_t bytes = (size_t)hw[2 * i] * hw[2 * i + 1] * 3;
    const uint8_t* dev = frames[i];
    if (!on_device) {
      if (via_pinned) memcpy(e->pin_stage + off, frames[i], bytes);   // the caller's buffer is free again when the call returns; one DMA below
      else HIP_CHECK(hipMemcpyAsync(e->frame_stage + off, frames[i], bytes, hipMemcpyHostToDevice, e->stream));
      dev = e->frame_stage + off;
      off += (bytes + 255) / 256 * 256;
    }
    fa.ptr[i] = dev;
    fa.scale_wh[2 * i] = (float)hw[2 * i + 1];      // orig_size = [w, h]  (src/rtdetr_detector.py:234)
    fa.scale_wh[2 * i + 1] = (float)hw[2 * i];
    if (hw[2 * i] != H || hw[2 * i + 1] != W) all_identity = false;
  }
  if (!on_device && via_pinned) HIP_CHECK(hipMemcpyAsync(e->frame_stage, e->pin_stage, off, hipMemcpyHostToDevice, e->stream));
  if (p->stem_fused) {
    // frames of the network's size are read in place; the others are resampled (PIL-exact, uint8) into the staging slots
    if (!all_identity && !e->u8_stage) HIP_CHECK(hipMalloc((void**)&e->u8_stage, (size_t)e->cfg.max_batch * H * W * 3));
    for (int i = 0; i < n; ++i) {
      if (hw[2 * i] == H && hw[2 * i + 1] == W) continue;
      const ResizeCoef& rc = resize_tables(e, hw[2 * i], hw[2 * i + 1]);
      uint8_t* dst = e->u8_stage + (size_t)i * H * W * 3;
      launch_resize_pil_u8(fa.ptr[i], hw[2 * i], hw[2 * i + 1], e->resize_tmp, dst, H, W, rc, e->stream);
      fa.ptr[i] = dst;
    }
    launch_set_frame_table(fa, p->frame_table, p->scale_wh, e->stream);
    e->last_fa = fa;
    return;
  }
  if (all_identity) {
    launch_preprocess_identity(fa, H, W, p->input, p->scale_wh, e->stream);
  } else {
    launch_set_scale(fa, p->scale_wh, e->stream);
    for (int i = 0; i < n; ++i) {
      // the same resampler handles an identity-sized frame exactly (1-tap coefficients of 1.0)
      const ResizeCoef& rc = resize_tables(e, hw[2 * i], hw[2 * i + 1]);
      launch_resize_pil(fa.ptr[i], hw[2 * i], hw[2 * i + 1], e->resize_tmp, p->input, i, rc, e->stream);
    }
  }
}

void check_n(rtd_engine* e, int n) {
  RTD_CHECK(e->loaded, RTD_E_STATE, "rtd_load_weights has not succeeded on this handle");
  RTD_CHECK(n >= 1 && n <= e->cfg.max_batch, RTD_E_INVALID, "batch size out of range");
}

// Nobody else may have left the handle's streams in capture mode (the library itself never captures): if some other component of the
// process did, say so instead of failing somewhere inside with a bare HIP code.
void check_streams_live(rtd_engine* e) {
  for (hipStream_t st : {e->stream, e->side}) {
    if (!st) continue;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    const hipError_t er = hipStreamIsCapturing(st, &cs);
    if (er != hipSuccess) (void)hipGetLastError();
    RTD_CHECK(er == hipSuccess && cs == hipStreamCaptureStatusNone, RTD_E_STATE,
              std::string("the handle's ") + (st == e->stream ? "main" : "side") + " stream is in capture state " + std::to_string((int)cs) + " (hipStreamIsCapturing: " +
                  hipGetErrorString(er) + "); this library never captures - another component of the process put it there");
  }
}

void forward(rtd_engine* e, int n, const uint8_t* const* frames, const int32_t* hw, bool on_device, bool via_pinned = false) {
  check_n(e, n);
  HIP_CHECK(hipSetDevice(e->cfg.device));
  check_streams_live(e);
  if (via_pinned && e->in_flight) { HIP_CHECK(hipStreamSynchronize(e->stream)); e->in_flight = false; }   // the staging buffers hold one batch
  Plan* p = get_plan(e, n);
  enqueue_frames(e, p, n, frames, hw, on_device, via_pinned);
  run_plan(e, p);
  e->last_n = n;
}

// zero-filled staging frames behind the fused stem's frame table (a plan that has not seen a real call yet: rtd_prepare, rtd_profile)
void point_at_blank_frames(rtd_engine* h, Plan* p, int n) {
  if (!p->stem_fused) return;
  const size_t fb = (size_t)h->cfg.input_h * h->cfg.input_w * 3;
  if (!h->u8_stage) HIP_CHECK(hipMalloc((void**)&h->u8_stage, (size_t)h->cfg.max_batch * fb));
  HIP_CHECK(hipMemsetAsync(h->u8_stage, 0, (size_t)h->cfg.max_batch * fb, h->stream));
  FrameArgs fa;
  memset(&fa, 0, sizeof fa);
  fa.n = n;
  for (int i = 0; i < n; ++i) { fa.ptr[i] = h->u8_stage + (size_t)i * fb; fa.scale_wh[2 * i] = (float)h->cfg.input_w; fa.scale_wh[2 * i + 1] = (float)h->cfg.input_h; }
  launch_set_frame_table(fa, p->frame_table, p->scale_wh, h->stream);
  h->last_fa = fa;
}


}  // namespace rtd_eng

namespace rtd {
__global__ void k_force_idx(int32_t* dst, const int32_t* src, const int32_t* flag, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && *flag) dst[i] = src[i];
}
}  // namespace rtd
void launch_force_idx(int32_t* dst, const int32_t* src, const int32_t* flag, int n, hipStream_t s) {
  rtd_launch(rtd::k_force_idx, dim3((n + 255) / 256), dim3(256), 0, s, dst, src, flag, n);
}



// =========================================================================================== C ABI
extern "C" {

const char* rtd_version(void) { return "mi355-rtdetr 0.1 (gfx950)"; }

const char* rtd_last_error(rtd_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int rtd_create(const rtd_config* cfg, rtd_handle* out) {
  if (!cfg || !out) { g_create_error = "null argument"; return RTD_E_INVALID; }
  try {
    RTD_CHECK(cfg->struct_size == (int32_t)sizeof(rtd_config), RTD_E_INVALID, "rtd_config.struct_size mismatch");
    RTD_CHECK(cfg->precision == RTD_PREC_BF16 || cfg->precision == RTD_PREC_FP32 || cfg->precision == RTD_PREC_F16X3, RTD_E_INVALID, "precision");
    RTD_CHECK(cfg->max_batch >= 1 && cfg->max_batch <= 64, RTD_E_INVALID, "max_batch must be in [1,64]");
    RTD_CHECK(cfg->profile == RTD_PROFILE_LATENCY || cfg->profile == RTD_PROFILE_THROUGHPUT, RTD_E_INVALID, "profile");
    // the FPN concatenates a 2x-upsampled map with the next level: every level must halve exactly
    RTD_CHECK(cfg->input_h >= 64 && cfg->input_w >= 64 && cfg->input_h % 32 == 0 && cfg->input_w % 32 == 0 &&
                  cfg->input_h <= 4096 && cfg->input_w <= 4096, RTD_E_INVALID, "input size must be a multiple of 32 in [64,4096]");
    RTD_CHECK(cfg->n_levels == 3 && cfg->n_points >= 1 && cfg->n_points <= 8, RTD_E_INVALID, "n_levels must be 3");
    RTD_CHECK(cfg->d_model % cfg->dec_heads == 0 && cfg->d_model / cfg->dec_heads == 32, RTD_E_INVALID, "decoder head dim must be 32");
    RTD_CHECK(cfg->enc_dim % cfg->enc_heads == 0 && cfg->enc_dim % 8 == 0 && cfg->csp_hidden % 8 == 0, RTD_E_INVALID, "encoder dims");
    RTD_CHECK(cfg->embedding_size % 16 == 0, RTD_E_INVALID, "embedding_size must be a multiple of 16");
    for (int i = 0; i < 4; ++i) RTD_CHECK(cfg->depths[i] >= 1 && cfg->hidden_sizes[i] % 32 == 0, RTD_E_INVALID, "stage config");
    RTD_CHECK(cfg->num_queries >= 1 && cfg->num_queries <= 1024 && cfg->num_classes % 4 == 0, RTD_E_INVALID, "num_queries <= 1024, num_classes % 4 == 0");
    RTD_CHECK((cfg->dec_heads * cfg->n_levels * cfg->n_points * 3) % 4 == 0, RTD_E_INVALID, "sampling head width");
    rtd_engine* e = new rtd_engine();
    e->cfg = *cfg;
    e->opts = g_opts;
    e->conv_opts = conv_opts_template();
    e->P = cfg->precision == RTD_PREC_BF16 ? BF16 : (cfg->precision == RTD_PREC_F16X3 ? F16X2 : F32);
    if (e->P == F16X2) {
      // hi/lo pairs travel in 32-channel groups (common.h F16X2): every trunk width must be whole groups
      bool ok = (cfg->embedding_size / 2) % SPLIT_GROUP == 0 && cfg->enc_dim % SPLIT_GROUP == 0 && cfg->csp_hidden % SPLIT_GROUP == 0 && cfg->d_model % SPLIT_GROUP == 0;
      for (int i = 0; i < 4; ++i) ok = ok && cfg->hidden_sizes[i] % SPLIT_GROUP == 0 && (cfg->layer_type != RTD_LAYER_BOTTLENECK || (cfg->hidden_sizes[i] / 4) % SPLIT_GROUP == 0);
      if (!ok) { delete e; RTD_CHECK(false, RTD_E_INVALID, "precision f16x3 needs every trunk channel count to be a multiple of 32 (embedding_size of 64)"); }
    }
    *out = e;
    return RTD_OK;
  } catch (const Error& er) {
    g_create_error = er.what();
    return er.code;
  } catch (const std::exception& ex) {
    g_create_error = ex.what();
    return RTD_E_INVALID;
  }
}

int rtd_load_weights(rtd_handle h, const void* blob, size_t nbytes) {
  return guarded(h, [&] {
    rtd_engine* e = h;
    RTD_CHECK(!e->loaded, RTD_E_STATE, "weights already loaded on this handle");
    RTD_CHECK(blob && nbytes >= 12, RTD_E_WEIGHTS, "empty weight blob");
    int ndev = 0;
    HIP_CHECK(hipGetDeviceCount(&ndev));
    RTD_CHECK(e->cfg.device >= 0 && e->cfg.device < ndev, RTD_E_INVALID, "device ordinal out of range");
    HIP_CHECK(hipSetDevice(e->cfg.device));
    if (!e->stream) HIP_CHECK(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    if (!e->side) HIP_CHECK(hipStreamCreateWithFlags(&e->side, hipStreamNonBlocking));
    if (!e->ev_fork) HIP_CHECK(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
    if (!e->ev_join) HIP_CHECK(hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming));
    if (!e->ev_xs) HIP_CHECK(hipEventCreateWithFlags(&e->ev_xs, hipEventDisableTiming));
    e->blob.assign((const char*)blob, (const char*)blob + nbytes);
    parse_blob(e);
    check_weight_range(e);
    const rtd_config& c = e->cfg;
    // level geometry (strides 8/16/32)
    int hh = down2(down2(c.input_h)), ww = down2(down2(c.input_w));
    e->S = 0;
    for (int l = 0; l < 3; ++l) {
      hh = down2(hh); ww = down2(ww);
      e->lvl_h[l] = hh; e->lvl_w[l] = ww; e->lvl_start[l] = e->S;
      e->S += hh * ww;
    }
    RTD_CHECK(e->lvl_h[0] == 2 * e->lvl_h[1] && e->lvl_h[1] == 2 * e->lvl_h[2] && e->lvl_w[0] == 2 * e->lvl_w[1] && e->lvl_w[1] == 2 * e->lvl_w[2],
              RTD_E_INVALID, "feature pyramid does not halve exactly");
    RTD_CHECK(c.num_queries <= e->S, RTD_E_INVALID, "num_queries exceeds the number of memory tokens");
    std::vector<float> anchors; std::vector<int32_t> invalid;
    make_anchors(e, anchors, invalid);
    e->anchors_dev = (float*)e->dmalloc(anchors.size() * 4);
    HIP_CHECK(hipMemcpy(e->anchors_dev, anchors.data(), anchors.size() * 4, hipMemcpyHostToDevice));
    e->n_invalid = (int)invalid.size();
    e->invalid_rows_dev = (int32_t*)e->dmalloc(invalid.size() * 4 + 16);
    if (!invalid.empty()) HIP_CHECK(hipMemcpy(e->invalid_rows_dev, invalid.data(), invalid.size() * 4, hipMemcpyHostToDevice));
    int32_t lv[9];
    for (int l = 0; l < 3; ++l) { lv[l * 3] = e->lvl_h[l]; lv[l * 3 + 1] = e->lvl_w[l]; lv[l * 3 + 2] = e->lvl_start[l]; }
    e->lvl_dev = (int32_t*)e->dmalloc(sizeof lv);
    HIP_CHECK(hipMemcpy(e->lvl_dev, lv, sizeof lv, hipMemcpyHostToDevice));
    std::vector<float> pos = sincos_pos(e->lvl_h[2], e->lvl_w[2], c.enc_dim);
    e->pos_dev = (float*)e->dmalloc(pos.size() * 4);
    HIP_CHECK(hipMemcpy(e->pos_dev, pos.data(), pos.size() * 4, hipMemcpyHostToDevice));
    e->forced_idx = (int32_t*)e->dmalloc((size_t)c.max_batch * c.num_queries * 4);
    e->force_flag = (int32_t*)e->dmalloc(16);
    HIP_CHECK(hipMemset(e->force_flag, 0, 16));
    HIP_CHECK(hipHostMalloc((void**)&e->block_host, (size_t)c.max_batch * c.num_queries * 6 * 4, hipHostMallocDefault));
    e->loaded = true;
    // build the bs=1 plan now: validates every tensor name / shape of the blob against the graph
    try {
      get_plan(e, 1);
    } catch (...) {
      e->loaded = false;
      throw;
    }
    e->blob.clear(); e->blob.shrink_to_fit(); e->host.clear();
  });
}

static void copy_block(rtd_engine* e, Plan* p, int n) {
  HIP_CHECK(hipMemcpyAsync(e->block_host, p->block6, (size_t)n * e->cfg.num_queries * 24, hipMemcpyDeviceToHost, e->stream));
  HIP_CHECK(hipStreamSynchronize(e->stream));
}

static void filter_rows(rtd_engine* h, int n, float conf, int32_t wildlife_only, rtd_det* out, int32_t* counts) {
  const int Q = h->cfg.num_queries;
  for (int i = 0; i < n; ++i) {
    int cnt = 0;
    for (int q = 0; q < Q; ++q) {
      const float* r = h->block_host + ((size_t)i * Q + q) * 6;
      const float score = r[1];
      if (score < conf) continue;                       // src/rtdetr_detector.py:271
      const int cid = (int)r[0];
      if (wildlife_only && !(cid == 0 || cid == 14 || cid == 15 || cid == 16 || cid == 21)) continue;  // :277
      rtd_det& d = out[(size_t)i * Q + cnt++];
      d.class_id = cid; d.score = score; d.x1 = r[2]; d.y1 = r[3]; d.x2 = r[4]; d.y2 = r[5];
    }
    counts[i] = cnt;
  }
}

int rtd_infer(rtd_handle h, int32_t n, const uint8_t* const* frames, const int32_t* hw, int32_t on_device,
              float conf, int32_t wildlife_only, rtd_det* out, int32_t* counts) {
  return guarded(h, [&] {
    RTD_CHECK(frames && hw && out && counts, RTD_E_INVALID, "null argument");
    forward(h, n, frames, hw, on_device != 0);
    copy_block(h, h->plans[n].get(), n);
    h->in_flight = false;
    filter_rows(h, n, conf, wildlife_only, out, counts);
  });
}

int rtd_infer_raw(rtd_handle h, int32_t n, const uint8_t* const* frames, const int32_t* hw, int32_t on_device,
                  int32_t* labels, float* boxes, float* scores) {
  return guarded(h, [&] {
    RTD_CHECK(frames && hw && labels && boxes && scores, RTD_E_INVALID, "null argument");
    forward(h, n, frames, hw, on_device != 0);
    copy_block(h, h->plans[n].get(), n);
    const int Q = h->cfg.num_queries;
    for (size_t t = 0; t < (size_t)n * Q; ++t) {
      const float* r = h->block_host + t * 6;
      labels[t] = (int32_t)r[0];
      scores[t] = r[1];
      memcpy(boxes + t * 4, r + 2, 16);
    }
  });
}

int rtd_infer_async(rtd_handle h, int32_t n, const uint8_t* const* frames, const int32_t* hw, int32_t frames_on_device) {
  return guarded(h, [&] {
    RTD_CHECK(frames && hw, RTD_E_INVALID, "null argument");
    try {
      forward(h, n, frames, hw, frames_on_device != 0, /*via_pinned=*/frames_on_device == 0);
    } catch (...) {
      // whatever was enqueued before the failure (the pinned -> HBM DMA) must not outlive the call: the next submit writes the staging buffer
      if (h->stream) (void)hipStreamSynchronize(h->stream);
      h->in_flight = false;
      throw;
    }
    h->in_flight = true;
    h->st_submits++;
  });
}

int rtd_collect(rtd_handle h, float conf, int32_t wildlife_only, rtd_det* out, int32_t* counts) {
  return guarded(h, [&] {
    RTD_CHECK(out && counts, RTD_E_INVALID, "null argument");
    RTD_CHECK(h->last_n > 0, RTD_E_STATE, "rtd_collect: no batch was submitted on this handle");
    HIP_CHECK(hipSetDevice(h->cfg.device));
    const int n = h->last_n;
    copy_block(h, h->plans[n].get(), n);
    h->in_flight = false;
    h->st_collects++;
    filter_rows(h, n, conf, wildlife_only, out, counts);
  });
}

int rtd_preprocess(rtd_handle h, const uint8_t* frame, int32_t fh, int32_t fw, int32_t frame_on_device, float* out_chw_dev) {
  return guarded(h, [&] {
    rtd_engine* e = h;
    RTD_CHECK(e->loaded, RTD_E_STATE, "weights not loaded");
    RTD_CHECK(frame && out_chw_dev && fh > 0 && fw > 0 && fh <= 16384 && fw <= 16384, RTD_E_INVALID, "frame pointer / size");
    HIP_CHECK(hipSetDevice(e->cfg.device));
    check_streams_live(e);
    const int H = e->cfg.input_h, W = e->cfg.input_w;
    if (e->in_flight) { HIP_CHECK(hipStreamSynchronize(e->stream)); }           // a submitted batch may still read the staging buffers used below
    const size_t bytes = (size_t)fh * fw * 3;
    const uint8_t* src = frame;
    if (!frame_on_device) {
      if (bytes > e->frame_stage_bytes) {
        HIP_CHECK(hipStreamSynchronize(e->stream));
        if (e->frame_stage) (void)hipFree(e->frame_stage);
        e->frame_stage = nullptr; e->frame_stage_bytes = 0;
        HIP_CHECK(hipMalloc((void**)&e->frame_stage, bytes));
        e->frame_stage_bytes = bytes;
      }
      HIP_CHECK(hipMemcpyAsync(e->frame_stage, frame, bytes, hipMemcpyHostToDevice, e->stream));
      src = e->frame_stage;
    }
    if (fh != H || fw != W) {                                                    // PIL-exact antialiased stretch, uint8 in and out (T.Resize on the PIL image)
      const size_t tmp = (size_t)fh * W * 3;
      if (tmp > e->resize_tmp_bytes) {
        HIP_CHECK(hipStreamSynchronize(e->stream));
        if (e->resize_tmp) (void)hipFree(e->resize_tmp);
        e->resize_tmp = nullptr; e->resize_tmp_bytes = 0;
        HIP_CHECK(hipMalloc((void**)&e->resize_tmp, tmp));
        e->resize_tmp_bytes = tmp;
      }
      if (!e->u8_stage) HIP_CHECK(hipMalloc((void**)&e->u8_stage, (size_t)e->cfg.max_batch * H * W * 3));
      const ResizeCoef& rc = resize_tables(e, fh, fw);
      launch_resize_pil_u8(src, fh, fw, e->resize_tmp, e->u8_stage, H, W, rc, e->stream);
      src = e->u8_stage;
    }
    launch_u8_hwc_to_chw_f32(src, H, W, out_chw_dev, e->stream);
    HIP_CHECK(hipStreamSynchronize(e->stream));
  });
}

int rtd_prepare(rtd_handle h, int32_t n) {
  return guarded(h, [&] {
    check_n(h, n);
    HIP_CHECK(hipSetDevice(h->cfg.device));
    check_streams_live(h);
    HIP_CHECK(hipStreamSynchronize(h->stream));
    h->in_flight = false;
    Plan* p = get_plan(h, n);
    if (!h->cfg.use_graph || p->exec) return;
    point_at_blank_frames(h, p, n);
    if (!p->stem_fused) {
      // the stand-alone preprocess writes plan->input per call; the arena is zero-filled, which is a valid (black) input
    }
    warm_and_build(h, p);
    HIP_CHECK(hipStreamSynchronize(h->stream));
  });
}

int rtd_result_block(rtd_handle h, float** dev_ptr, int64_t* n_floats) {
  return guarded(h, [&] {
    RTD_CHECK(dev_ptr && n_floats && h->last_n > 0, RTD_E_STATE, "no forward has run");
    *dev_ptr = h->plans[h->last_n]->block6;
    *n_floats = (int64_t)h->last_n * h->cfg.num_queries * 6;
  });
}

int rtd_sync(rtd_handle h) {
  return guarded(h, [&] {
    if (h->stream) { HIP_CHECK(hipSetDevice(h->cfg.device)); HIP_CHECK(hipStreamSynchronize(h->stream)); }
    h->in_flight = false;
  });
}

// Ordering against streams the CALLER owns (torch's current stream, the stream RCCL runs on).  Both directions use an event that belongs
// to the handle and is recorded / waited on with plain HIP calls: no torch- or RCCL-created event ever touches the handle's streams, and
// the handle's streams are never handed to torch (no ExternalStream) - VERDICT r3 item 1(a).
int rtd_wait_stream(rtd_handle h, void* producer_stream) {
  return guarded(h, [&] {
    RTD_CHECK(h->loaded, RTD_E_STATE, "weights not loaded");
    HIP_CHECK(hipSetDevice(h->cfg.device));
    HIP_CHECK(hipEventRecord(h->ev_xs, (hipStream_t)producer_stream));
    HIP_CHECK(hipStreamWaitEvent(h->stream, h->ev_xs, 0));
  });
}

int rtd_signal_stream(rtd_handle h, void* consumer_stream) {
  return guarded(h, [&] {
    RTD_CHECK(h->loaded, RTD_E_STATE, "weights not loaded");
    HIP_CHECK(hipSetDevice(h->cfg.device));
    HIP_CHECK(hipEventRecord(h->ev_xs, h->stream));
    HIP_CHECK(hipStreamWaitEvent((hipStream_t)consumer_stream, h->ev_xs, 0));
  });
}

// Real-weights guard, part 2: the handle's arithmetic against the library's own exact fp32 engine, on ONE built-in frame, with THESE weights.
// Two temporary bs-1 handles are built from the handle's blob: the fp32 engine (the reference arithmetic, ~2 ms per frame) and an engine of
// the handle's precision whose activations all keep their own buffer (arena_reuse off), so that after the forward every F16X2 activation of
// the network can be scanned for the saturation value.  Rows are matched like the parity tests match them (same label, |dscore| <= 1e-3,
// max |dbox| <= 1e-2 px, greedy) - near-ties at the two top-k cuts can cost single rows on any engine, a checkpoint outside the format's
// range costs most of them.
static void check_frame(int H, int W, std::vector<uint8_t>& f) {
  f.resize((size_t)H * W * 3);
  uint32_t lcg = 12345u;
  auto rnd = [&]() { lcg = lcg * 1664525u + 1013904223u; return lcg >> 8; };
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x)
      for (int c = 0; c < 3; ++c) f[((size_t)y * W + x) * 3 + c] = (uint8_t)(70 + 40 * c + (x * (60 + 20 * c)) / W + (y * (50 - 10 * c)) / H);
  for (int o = 0; o < 14; ++o) {
    const int cx = (int)(rnd() % (unsigned)W), cy = (int)(rnd() % (unsigned)H);
    const int sx = std::max(4, (int)(rnd() % (unsigned)std::max(1, W / 4))), sy = std::max(4, (int)(rnd() % (unsigned)std::max(1, H / 4)));
    const uint8_t col[3] = {(uint8_t)(rnd() & 255), (uint8_t)(rnd() & 255), (uint8_t)(rnd() & 255)};
    const bool ell = rnd() & 1;
    for (int y = std::max(0, cy - sy); y < std::min(H, cy + sy); ++y)
      for (int x = std::max(0, cx - sx); x < std::min(W, cx + sx); ++x) {
        if (ell) {
          const float dx = (float)(x - cx) / sx, dy = (float)(y - cy) / sy;
          if (dx * dx + dy * dy >= 1.f) continue;
        }
        for (int c = 0; c < 3; ++c) f[((size_t)y * W + x) * 3 + c] = col[c];
      }
  }
  for (auto& v : f) v = (uint8_t)std::min(255, std::max(0, (int)v + (int)(rnd() % 13u) - 6));
}

int rtd_self_check(rtd_handle h, const void* blob, size_t nbytes, rtd_check_report* out) {
  return guarded(h, [&] {
    RTD_CHECK(out && out->struct_size == (int32_t)sizeof(rtd_check_report), RTD_E_INVALID, "rtd_check_report.struct_size mismatch");
    RTD_CHECK(h->loaded, RTD_E_STATE, "weights not loaded");
    RTD_CHECK(blob && nbytes >= 12, RTD_E_WEIGHTS, "self check: the weight blob given to rtd_load_weights is needed again (the handle does not keep its host copy)");
    HIP_CHECK(hipSetDevice(h->cfg.device));
    const int H = h->cfg.input_h, W = h->cfg.input_w, Q = h->cfg.num_queries;
    std::vector<uint8_t> frame;
    check_frame(H, W, frame);
    const uint8_t* fp = frame.data();
    const int32_t hw[2] = {H, W};
    struct Rows { std::vector<int32_t> l; std::vector<float> b, s; };
    int64_t saturated = 0;
    auto run = [&](int precision, bool scan, Rows& r) {
      rtd_config c = h->cfg;
      c.precision = precision; c.max_batch = 1; c.use_graph = 0;
      rtd_handle t = nullptr;
      RTD_CHECK(rtd_create(&c, &t) == RTD_OK, RTD_E_HIP, "self check: rtd_create failed: " + std::string(rtd_last_error(nullptr)));
      t->opts.arena_reuse = scan ? 0 : t->opts.arena_reuse;      // every activation keeps its own buffer until the scan
      t->opts.side_stream = 0;
      int rc = rtd_load_weights(t, blob, nbytes);
      r.l.resize(Q); r.b.resize((size_t)Q * 4); r.s.resize(Q);
      if (rc == RTD_OK) rc = rtd_infer_raw(t, 1, &fp, hw, 0, r.l.data(), r.b.data(), r.s.data());
      std::string msg = rc == RTD_OK ? "" : std::string(rtd_last_error(t));
      if (rc == RTD_OK && scan) {
        unsigned long long* cnt = nullptr;
        hipError_t er = hipMalloc((void**)&cnt, 8);
        if (er == hipSuccess) er = hipMemsetAsync(cnt, 0, 8, t->stream);
        if (er == hipSuccess) {
          for (const auto& a : t->plans[1]->split_acts) launch_count_saturated(a.first, (int64_t)(a.second / 2), cnt, t->stream);
          unsigned long long v = 0;
          er = hipMemcpyAsync(&v, cnt, 8, hipMemcpyDeviceToHost, t->stream);
          if (er == hipSuccess) er = hipStreamSynchronize(t->stream);
          saturated = (int64_t)v;
        }
        if (cnt) (void)hipFree(cnt);
        if (er != hipSuccess) { rc = RTD_E_HIP; msg = hipGetErrorString(er); }
      }
      rtd_destroy(t);
      RTD_CHECK(rc == RTD_OK, rc, "self check (" + std::string(precision == RTD_PREC_FP32 ? "fp32" : "subject") + " engine): " + msg);
    };
    Rows ref, got;
    run(RTD_PREC_FP32, false, ref);
    run(h->cfg.precision, h->cfg.precision == RTD_PREC_F16X3, got);
    HIP_CHECK(hipSetDevice(h->cfg.device));
    const float stol = 1e-3f, btol = 1e-2f;
    std::vector<char> used(Q, 0);
    int matched = 0;
    float ws = 0.f, wb = 0.f;
    for (int i = 0; i < Q; ++i) {
      int best = -1;
      float bd = 0.f;
      for (int j = 0; j < Q; ++j) {
        if (used[j] || got.l[j] != ref.l[i] || fabsf(got.s[j] - ref.s[i]) > stol) continue;
        float d = 0.f;
        for (int k = 0; k < 4; ++k) d = std::max(d, fabsf(got.b[(size_t)j * 4 + k] - ref.b[(size_t)i * 4 + k]));
        if (best < 0 || d < bd) { best = j; bd = d; }
      }
      if (best >= 0 && bd <= btol) {
        used[best] = 1; ++matched;
        ws = std::max(ws, fabsf(got.s[best] - ref.s[i])); wb = std::max(wb, bd);
      }
    }
    bool finite = true;
    for (int i = 0; i < Q; ++i) finite = finite && std::isfinite(got.s[i]) && std::isfinite(got.b[(size_t)i * 4]);
    memset(out, 0, sizeof *out);
    out->struct_size = (int32_t)sizeof(rtd_check_report);
    out->rows = Q; out->rows_matched = finite ? matched : 0;
    out->worst_score_err = ws; out->worst_box_err_px = wb;
    out->score_tol = stol; out->box_tol_px = btol;
    out->saturated_values = saturated;
    out->max_abs_filter = h->max_abs_filter;
    strncpy(out->max_abs_filter_name, h->max_abs_filter_name.c_str(), sizeof(out->max_abs_filter_name) - 1);
    h->st_saturated = saturated;
  });
}

int rtd_get_stats(rtd_handle h, rtd_stats* out) {
  if (!h || !out) return RTD_E_INVALID;
  std::lock_guard<std::mutex> lk(h->mu);
  memset(out, 0, sizeof *out);
  out->struct_size = (int32_t)sizeof(rtd_stats);
  out->plans = h->st_plans; out->graphs = h->st_graphs; out->graph_nodes = h->st_graph_nodes; out->graph_launches = h->st_graph_launches;
  out->eager_passes = h->st_eager; out->submits = h->st_submits; out->collects = h->st_collects; out->failed_calls = h->st_failed;
  out->last_error_code = h->st_last_code;
  out->in_flight = h->in_flight ? 1 : 0;
  out->saturated_values = h->st_saturated;
  out->max_abs_filter = h->max_abs_filter;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (h->stream && hipStreamIsCapturing(h->stream, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusInvalidated; }
  out->stream_capture_status = (int32_t)cs;
  return RTD_OK;
}

void* rtd_stream(rtd_handle h) { return h ? (void*)h->stream : nullptr; }

int64_t rtd_arena_bytes(rtd_handle h) {
  if (!h) return 0;
  std::lock_guard<std::mutex> lk(h->mu);
  int64_t t = 0;
  for (auto& kv : h->plans) t += (int64_t)kv.second->arena_bytes;
  return t;
}

void rtd_destroy(rtd_handle h) {
  if (!h) return;
  {
    std::lock_guard<std::mutex> lk(h->mu);
    (void)hipSetDevice(h->cfg.device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->side) (void)hipStreamSynchronize(h->side);
    for (auto& kv : h->plans) {
      if (kv.second->exec) (void)hipGraphExecDestroy(kv.second->exec);
      if (kv.second->graph) (void)hipGraphDestroy(kv.second->graph);
    }
    for (void* p : h->allocs) (void)hipFree(p);
    if (h->frame_stage) (void)hipFree(h->frame_stage);
    if (h->resize_tmp) (void)hipFree(h->resize_tmp);
    if (h->u8_stage) (void)hipFree(h->u8_stage);
    if (h->block_host) (void)hipHostFree(h->block_host);
    if (h->pin_stage) (void)hipHostFree(h->pin_stage);
    if (h->side) (void)hipStreamSynchronize(h->side);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);             // only ever recorded on live streams: nothing in this library captures
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    if (h->ev_xs) (void)hipEventDestroy(h->ev_xs);
    if (h->side) (void)hipStreamDestroy(h->side);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    (void)hipGetLastError();                         // nothing a teardown call reported may stay behind as this thread's sticky error
  }
  delete h;
}

int rtd_crop_resize_batch(int32_t n, const uint8_t* const* frames_dev, const int32_t* frame_hw, const int32_t* rects, int32_t out_size,
                          const float* mean3, const float* std3, float* out_dev, void* stream) {
  // asynchronous: validated, copied into the launch and enqueued on the caller's stream - no device synchronisation (the header)
  try {
    [&] {
    RTD_CHECK(n >= 0 && frames_dev && frame_hw && rects && mean3 && std3 && out_dev, RTD_E_INVALID, "null argument");
    for (int base = 0; base < n; base += 64) {
      const int m = std::min(64, n - base);
      CropBatch cb;
      memset(&cb, 0, sizeof cb);
      for (int i = 0; i < m; ++i) {
        cb.frame[i] = frames_dev[base + i];
        cb.fh[i] = frame_hw[2 * (base + i)]; cb.fw[i] = frame_hw[2 * (base + i) + 1];
        cb.x1[i] = rects[4 * (base + i)]; cb.y1[i] = rects[4 * (base + i) + 1];
        cb.x2[i] = rects[4 * (base + i) + 2]; cb.y2[i] = rects[4 * (base + i) + 3];
      }
      launch_crop_resize(cb, m, out_size, mean3, std3, out_dev + (size_t)base * 3 * out_size * out_size, (hipStream_t)stream);
    }
    }();
    return RTD_OK;
  } catch (const Error& er) {
    g_create_error = er.what();
    return er.code;
  } catch (const std::exception& ex) {
    g_create_error = ex.what();
    return RTD_E_HIP;
  }
}

}  // extern "C"
