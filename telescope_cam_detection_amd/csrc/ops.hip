// ops.hip - every non-GEMM kernel of the RT-DETRv2 path for gfx950 (wave = 64 lanes).
//
// These are HBM/L2-bound byte-moving or small fp32 kernels: coalesced 16-byte accesses along the
// NHWC channel dimension, fp32 arithmetic regardless of the storage type, no MFMA reshaping.
#include <float.h>

#include <algorithm>

#include "common.h"

namespace rtd {

// ------------------------------------------------------------------------------------------ helpers
template <typename T> __device__ __forceinline__ float ldf(const T* p) { return (float)(*p); }
template <typename T> __device__ __forceinline__ void stf(T* p, float v) { *p = (T)v; }

__device__ __forceinline__ float wave_sum(float v) { return wave_sum64(v); }

static inline unsigned blocks_for(int64_t n, int per_block) { return (unsigned)((n + per_block - 1) / per_block); }

#define DISPATCH_T(dt, ...)                     \
  do {                                          \
    if ((dt) == BF16) { typedef bf16 T; __VA_ARGS__; } \
    else { typedef float T; __VA_ARGS__; }      \
  } while (0)

// ------------------------------------------------------------------------------------------ dtype conversion
template <typename T>
__global__ void k_f32_to(const float* __restrict__ src, T* __restrict__ dst, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = (T)src[i];
}
template <typename T>
__global__ void k_to_f32(const T* __restrict__ src, float* __restrict__ dst, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = (float)src[i];
}
void launch_f32_to(const float* src, void* dst, int dt, int64_t n, hipStream_t s) {
  if (n == 0) return;
  DISPATCH_T(dt, rtd_launch(k_f32_to<T>, dim3(blocks_for(n, 256)), dim3(256), 0, s, src, (T*)dst, n));
  HIP_CHECK(hipGetLastError());
}
void launch_to_f32(const void* src, int dt, float* dst, int64_t n, hipStream_t s) {
  if (n == 0) return;
  DISPATCH_T(dt, rtd_launch(k_to_f32<T>, dim3(blocks_for(n, 256)), dim3(256), 0, s, (const T*)src, dst, n));
  HIP_CHECK(hipGetLastError());
}

// fp32 rows <-> F16X2 rows (common.h: groups of 32 channels, [32 hi | 32 lo]); a thread moves 8 channels
__global__ void k_f32_to_split(const float* __restrict__ src, int64_t lds_, sp16* __restrict__ dst, int64_t ldd, int64_t rows, int C) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int c8 = C / 8;
  if (i >= rows * c8) return;
  const int64_t r = i / c8;
  const int c = (int)(i - r * c8) * 8;
  const f32x4 a = *(const f32x4*)(src + r * lds_ + c), b = *(const f32x4*)(src + r * lds_ + c + 4);
  const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  split_store8(dst, r * ldd, c, v);
}
__global__ void k_split_to_f32(const sp16* __restrict__ src, int64_t lds_, float* __restrict__ dst, int64_t ldd, int64_t rows, int C) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int c8 = C / 8;
  if (i >= rows * c8) return;
  const int64_t r = i / c8;
  const int c = (int)(i - r * c8) * 8;
  float v[8];
  split_load8(src, r * lds_, c, v);
  *(f32x4*)(dst + r * ldd + c) = f32x4{v[0], v[1], v[2], v[3]};
  *(f32x4*)(dst + r * ldd + c + 4) = f32x4{v[4], v[5], v[6], v[7]};
}
void launch_f32_to_split(const float* src, int64_t lds_, void* dst, int64_t ldd, int64_t rows, int C, hipStream_t s) {
  if (rows == 0) return;
  RTD_CHECK(C % SPLIT_GROUP == 0 && lds_ % 4 == 0 && ldd % SPLIT_GROUP == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0, 1, "f32 -> split: 32-channel groups, 16-byte rows");
  rtd_launch(k_f32_to_split, dim3(blocks_for(rows * (C / 8), 256)), dim3(256), 0, s, src, lds_, (sp16*)dst, ldd, rows, C);
  HIP_CHECK(hipGetLastError());
}
void launch_split_to_f32(const void* src, int64_t lds_, float* dst, int64_t ldd, int64_t rows, int C, hipStream_t s) {
  if (rows == 0) return;
  RTD_CHECK(C % SPLIT_GROUP == 0 && lds_ % SPLIT_GROUP == 0 && ldd % 4 == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0, 1, "split -> f32: 32-channel groups, 16-byte rows");
  rtd_launch(k_split_to_f32, dim3(blocks_for(rows * (C / 8), 256)), dim3(256), 0, s, (const sp16*)src, lds_, dst, ldd, rows, C);
  HIP_CHECK(hipGetLastError());
}

// Real-weights guard (rtd_self_check): how many hi halves of an F16X2 buffer sit at the format's saturation value +-65504 (0x7BFF: split2
// clamps there instead of producing inf).  A legitimate activation of exactly 65504 is as good as impossible, so a non-zero count says the
// checkpoint drives this engine outside its range.  `n16` 16-bit words, groups of [32 hi | 32 lo].
__global__ void k_count_saturated(const unsigned short* __restrict__ p, int64_t n16, unsigned long long* __restrict__ count) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  unsigned c = 0;
  for (; i < n16; i += step)
    if ((i & 63) < 32 && (p[i] & 0x7FFFu) == 0x7BFFu) ++c;
  if (c) atomicAdd(count, (unsigned long long)c);
}
void launch_count_saturated(const void* split_buf, int64_t n16, unsigned long long* count_dev, hipStream_t s) {
  if (n16 <= 0) return;
  rtd_launch(k_count_saturated, dim3(std::min<unsigned>(blocks_for(n16, 256), 4096u)), dim3(256), 0, s, (const unsigned short*)split_buf, n16, count_dev);
  HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------ layer norm
// y = LN(x (+ res)) * g + b over the last dim; one wave per row.  torch.nn.functional.layer_norm
// semantics (biased variance, eps inside the sqrt).  HF:v2.py:861,886 (post-norm residual blocks).
template <typename TX, typename TR, typename TY>
__global__ __launch_bounds__(256) void k_layernorm(const TX* __restrict__ x, int64_t ldx, const TR* __restrict__ r,
                                                    int64_t ldr, const float* __restrict__ g,
                                                    const float* __restrict__ bta, TY* __restrict__ y, int64_t ldy,
                                                    int rows, int dim, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  constexpr int MAXE = 16;  // dim <= 1024
  float v[MAXE];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < MAXE; ++i) {
    const int c = lane + i * 64;
    float t = 0.f;
    if (c < dim) {
      t = ldf(x + (int64_t)row * ldx + c);
      if (r) t += ldf(r + (int64_t)row * ldr + c);
    }
    v[i] = t;
    sum += t;
  }
  const float mean = wave_sum(sum) / (float)dim;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < MAXE; ++i) {
    const int c = lane + i * 64;
    const float d = (c < dim) ? (v[i] - mean) : 0.f;
    sq += d * d;
  }
  const float rstd = rsqrtf(wave_sum(sq) / (float)dim + eps);
#pragma unroll
  for (int i = 0; i < MAXE; ++i) {
    const int c = lane + i * 64;
    if (c < dim) stf(y + (int64_t)row * ldy + c, (v[i] - mean) * rstd * g[c] + bta[c]);
  }
}

void launch_layernorm(const Tensor& x, const Tensor* res, const float* g, const float* b, const Tensor& y, float eps,
                      hipStream_t s) {
  const int rows = (int)x.pixels(), dim = x.c;
  RTD_CHECK(dim <= 1024 && y.c == dim && y.pixels() == rows, 1, "layernorm: shape");
  RTD_CHECK(x.bstride == (int64_t)x.h * x.w * x.ld && y.bstride == (int64_t)y.h * y.w * y.ld, 1, "layernorm: dense rows");
  if (res) RTD_CHECK(res->c == dim && res->pixels() == rows && res->bstride == (int64_t)res->h * res->w * res->ld, 1, "layernorm: residual");
  const dim3 grid((rows + 3) / 4), blk(256);
#define LN_GO(TX, TR, TY)                                                                                   \
  rtd_launch((k_layernorm<TX, TR, TY>), grid, blk, 0, s, (const TX*)x.p, x.ld,                       \
                     (const TR*)(res ? res->p : nullptr), res ? res->ld : 0, g, b, (TY*)y.p, y.ld, rows, dim, eps)
  const int rdt = res ? res->dt : x.dt;
  const int key = (x.dt == F32) * 4 + (rdt == F32) * 2 + (y.dt == F32);
  switch (key) {
    case 0: LN_GO(bf16, bf16, bf16); break;
    case 1: LN_GO(bf16, bf16, float); break;
    case 2: LN_GO(bf16, float, bf16); break;
    case 3: LN_GO(bf16, float, float); break;
    case 4: LN_GO(float, bf16, bf16); break;
    case 5: LN_GO(float, bf16, float); break;
    case 6: LN_GO(float, float, bf16); break;
    default: LN_GO(float, float, float); break;
  }
#undef LN_GO
  HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------ add (x + pos)
// y[b,p,c] = a[b,p,c] + pos[(b or 0),p,c]   - the "(hidden + position_embeddings)" of HF:v2.py:319
template <typename TA, typename TB, typename TY>
__global__ void k_add(const TA* __restrict__ a, const TB* __restrict__ b, TY* __restrict__ y, int64_t per_image,
                      int64_t total, int b_broadcast) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int64_t j = b_broadcast ? (i % per_image) : i;
  y[i] = (TY)((float)a[i] + (float)b[j]);
}
void launch_add(const Tensor& a, const Tensor& b, const Tensor& y, hipStream_t s) {
  RTD_CHECK(a.ld == a.c && b.ld == b.c && y.ld == y.c && a.c == b.c && a.c == y.c, 1, "add: dense tensors");
  const int64_t per = (int64_t)a.h * a.w * a.c, total = per * a.n;
  RTD_CHECK((b.n == 1 || b.n == a.n) && (int64_t)b.h * b.w * b.c == per && y.pixels() == a.pixels(), 1, "add: shape");
  const dim3 grid(blocks_for(total, 256)), blk(256);
  const int bb = (b.n == 1 && a.n != 1);
#define ADD_GO(TA, TB, TY) rtd_launch((k_add<TA, TB, TY>), grid, blk, 0, s, (const TA*)a.p, (const TB*)b.p, (TY*)y.p, per, total, bb)
  const int key = (a.dt == F32) * 4 + (b.dt == F32) * 2 + (y.dt == F32);
  switch (key) {
    case 0: ADD_GO(bf16, bf16, bf16); break;
    case 1: ADD_GO(bf16, bf16, float); break;
    case 2: ADD_GO(bf16, float, bf16); break;
    case 3: ADD_GO(bf16, float, float); break;
    case 4: ADD_GO(float, bf16, bf16); break;
    case 5: ADD_GO(float, bf16, float); break;
    case 6: ADD_GO(float, float, bf16); break;
    default: ADD_GO(float, float, float); break;
  }
#undef ADD_GO
  HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------ max-pool 3x3 s2 p1
// HF:rt_detr_resnet.py:103 nn.MaxPool2d(3, 2, 1); padding counts as -inf.  4 channels per thread.
template <typename T>
__global__ void k_maxpool(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int C, int64_t ldx, int OH,
                          int OW, int64_t ldy) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int c4 = C / 4;
  const int64_t total = (int64_t)B * OH * OW * c4;
  if (i >= total) return;
  const int cc = (int)(i % c4) * 4;
  int64_t p = i / c4;
  const int ox = (int)(p % OW); p /= OW;
  const int oy = (int)(p % OH);
  const int b = (int)(p / OH);
  float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  for (int dy = 0; dy < 3; ++dy) {
    const int iy = oy * 2 - 1 + dy;
    if ((unsigned)iy >= (unsigned)H) continue;
    for (int dx = 0; dx < 3; ++dx) {
      const int ix = ox * 2 - 1 + dx;
      if ((unsigned)ix >= (unsigned)W) continue;
      const T* q = x + (((int64_t)b * H + iy) * W + ix) * ldx + cc;
#pragma unroll
      for (int k = 0; k < 4; ++k) m[k] = fmaxf(m[k], (float)q[k]);
    }
  }
  T* o = y + (((int64_t)b * OH + oy) * OW + ox) * ldy + cc;
#pragma unroll
  for (int k = 0; k < 4; ++k) o[k] = (T)m[k];
}
// bf16, C % 8 == 0, even output extents: a thread owns 8 channels of a 2 x 2 output patch and reads its 5 x 5 input window
// once (25 x 16 bytes for 4 outputs instead of 36 x 8; the taps a patch shares between its outputs never leave registers,
// and the lanes of 8 consecutive threads cover one pixel's 128-byte line)
__global__ __launch_bounds__(256) void k_maxpool_bf16_2x2(const bf16* __restrict__ x, bf16* __restrict__ y, int B, int H, int W, int C, int64_t ldx,
                                                          int OH, int OW, int64_t ldy) {
  typedef __bf16 v8 __attribute__((ext_vector_type(8)));
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int c8 = C / 8, PW = OW / 2, PH = OH / 2;
  const int64_t total = (int64_t)B * PH * PW * c8;
  if (i >= total) return;
  const int cc = (int)(i % c8) * 8;
  int64_t p = i / c8;
  const int px = (int)(p % PW); p /= PW;
  const int py = (int)(p % PH);
  const int b = (int)(p / PH);
  const int iy0 = 4 * py - 1, ix0 = 4 * px - 1;                  // window rows iy0 .. iy0+4, cols ix0 .. ix0+4
  float m[4][8];
#pragma unroll
  for (int o = 0; o < 4; ++o)
#pragma unroll
    for (int k = 0; k < 8; ++k) m[o][k] = -INFINITY;
#pragma unroll
  for (int dy = 0; dy < 5; ++dy) {
    const int iy = iy0 + dy;
    if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
    for (int dx = 0; dx < 5; ++dx) {
      const int ix = ix0 + dx;
      if ((unsigned)ix >= (unsigned)W) continue;
      const v8 v = *(const v8*)(x + (((int64_t)b * H + iy) * W + ix) * ldx + cc);
#pragma unroll
      for (int oy = 0; oy < 2; ++oy)
#pragma unroll
        for (int ox = 0; ox < 2; ++ox)
          if (dy >= 2 * oy && dy <= 2 * oy + 2 && dx >= 2 * ox && dx <= 2 * ox + 2) {     // compile-time after unrolling
#pragma unroll
            for (int k = 0; k < 8; ++k) m[oy * 2 + ox][k] = fmaxf(m[oy * 2 + ox][k], (float)v[k]);
          }
    }
  }
#pragma unroll
  for (int oy = 0; oy < 2; ++oy)
#pragma unroll
    for (int ox = 0; ox < 2; ++ox) {
      v8 o;
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = (bf16)m[oy * 2 + ox][k];
      *(v8*)(y + (((int64_t)b * OH + 2 * py + oy) * OW + 2 * px + ox) * ldy + cc) = o;
    }
}

// F16X2: 8 channels (one 16-byte hi chunk + one 16-byte lo chunk) of one output per thread; hi + lo is exact in fp32, so the max
// is the max of the represented values and re-splitting it reproduces the winning tap's (hi, lo) pair bit for bit
__global__ __launch_bounds__(256) void k_maxpool_split(const sp16* __restrict__ x, sp16* __restrict__ y, int B, int H, int W, int C, int64_t ldx,
                                                       int OH, int OW, int64_t ldy) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int c8 = C / 8;
  const int64_t total = (int64_t)B * OH * OW * c8;
  if (i >= total) return;
  const int cc = (int)(i % c8) * 8;
  int64_t p = i / c8;
  const int ox = (int)(p % OW); p /= OW;
  const int oy = (int)(p % OH);
  const int b = (int)(p / OH);
  float m[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) m[k] = -INFINITY;
#pragma unroll
  for (int dy = 0; dy < 3; ++dy) {
    const int iy = oy * 2 - 1 + dy;
    if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int ix = ox * 2 - 1 + dx;
      if ((unsigned)ix >= (unsigned)W) continue;
      float v[8];
      split_load8(x, (((int64_t)b * H + iy) * W + ix) * ldx, cc, v);
#pragma unroll
      for (int k = 0; k < 8; ++k) m[k] = fmaxf(m[k], v[k]);
    }
  }
  split_store8(y, (((int64_t)b * OH + oy) * OW + ox) * ldy, cc, m);
}

// ... on a 2 x 2 output patch per thread (even output extents): the 5 x 5 input window is read once - 25 x 32 bytes for 4 outputs instead
// of 36 x 32 - and the taps the patch's outputs share stay in registers (k_maxpool_bf16_2x2's scheme)
__global__ __launch_bounds__(256) void k_maxpool_split_2x2(const sp16* __restrict__ x, sp16* __restrict__ y, int B, int H, int W, int C, int64_t ldx,
                                                           int OH, int OW, int64_t ldy) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int c8 = C / 8, PW = OW / 2, PH = OH / 2;
  const int64_t total = (int64_t)B * PH * PW * c8;
  if (i >= total) return;
  const int cc = (int)(i % c8) * 8;
  int64_t p = i / c8;
  const int px = (int)(p % PW); p /= PW;
  const int py = (int)(p % PH);
  const int b = (int)(p / PH);
  const int iy0 = 4 * py - 1, ix0 = 4 * px - 1;
  float m[4][8];
#pragma unroll
  for (int o = 0; o < 4; ++o)
#pragma unroll
    for (int k = 0; k < 8; ++k) m[o][k] = -INFINITY;
#pragma unroll
  for (int dy = 0; dy < 5; ++dy) {
    const int iy = iy0 + dy;
    if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
    for (int dx = 0; dx < 5; ++dx) {
      const int ix = ix0 + dx;
      if ((unsigned)ix >= (unsigned)W) continue;
      float v[8];
      split_load8(x, (((int64_t)b * H + iy) * W + ix) * ldx, cc, v);
#pragma unroll
      for (int oy = 0; oy < 2; ++oy)
#pragma unroll
        for (int ox = 0; ox < 2; ++ox)
          if (dy >= 2 * oy && dy <= 2 * oy + 2 && dx >= 2 * ox && dx <= 2 * ox + 2) {     // compile-time after unrolling
#pragma unroll
            for (int k = 0; k < 8; ++k) m[oy * 2 + ox][k] = fmaxf(m[oy * 2 + ox][k], v[k]);
          }
    }
  }
#pragma unroll
  for (int oy = 0; oy < 2; ++oy)
#pragma unroll
    for (int ox = 0; ox < 2; ++ox) split_store8(y, (((int64_t)b * OH + 2 * py + oy) * OW + 2 * px + ox) * ldy, cc, m[oy * 2 + ox]);
}

void launch_maxpool3x3s2(const Tensor& x, const Tensor& y, hipStream_t s) {
  RTD_CHECK(x.dt == y.dt && x.c == y.c && x.c % 4 == 0 && x.n == y.n, 1, "maxpool: dtype/channels");
  RTD_CHECK(y.h == (x.h + 2 - 3) / 2 + 1 && y.w == (x.w + 2 - 3) / 2 + 1, 1, "maxpool: shape");
  RTD_CHECK(x.bstride == (int64_t)x.h * x.w * x.ld && y.bstride == (int64_t)y.h * y.w * y.ld, 1, "maxpool: dense images");
  if (x.dt == F16X2) {
    RTD_CHECK(x.c % SPLIT_GROUP == 0 && x.ld % SPLIT_GROUP == 0 && y.ld % SPLIT_GROUP == 0 && (((uintptr_t)x.p | (uintptr_t)y.p) & 15) == 0, 1, "maxpool: split layout");
    if (y.h % 2 == 0 && y.w % 2 == 0) {
      const int64_t total2 = (int64_t)y.n * (y.h / 2) * (y.w / 2) * (y.c / 8);
      rtd_launch(k_maxpool_split_2x2, dim3(blocks_for(total2, 256)), dim3(256), 0, s, (const sp16*)x.p, (sp16*)y.p, x.n, x.h, x.w, x.c, x.ld, y.h, y.w, y.ld);
      HIP_CHECK(hipGetLastError());
      return;
    }
    const int64_t total = (int64_t)y.n * y.h * y.w * (y.c / 8);
    rtd_launch(k_maxpool_split, dim3(blocks_for(total, 256)), dim3(256), 0, s, (const sp16*)x.p, (sp16*)y.p, x.n, x.h, x.w, x.c, x.ld, y.h, y.w, y.ld);
    HIP_CHECK(hipGetLastError());
    return;
  }
  if (x.dt == BF16 && x.c % 8 == 0 && x.ld % 8 == 0 && y.ld % 8 == 0 && y.h % 2 == 0 && y.w % 2 == 0 && (((uintptr_t)x.p | (uintptr_t)y.p) & 15) == 0) {
    const int64_t total2 = (int64_t)y.n * (y.h / 2) * (y.w / 2) * (y.c / 8);
    rtd_launch(k_maxpool_bf16_2x2, dim3(blocks_for(total2, 256)), dim3(256), 0, s, (const bf16*)x.p, (bf16*)y.p, x.n, x.h, x.w, x.c,
                       x.ld, y.h, y.w, y.ld);
    HIP_CHECK(hipGetLastError());
    return;
  }
  const int64_t total = (int64_t)y.n * y.h * y.w * (y.c / 4);
  DISPATCH_T(x.dt, rtd_launch(k_maxpool<T>, dim3(blocks_for(total, 256)), dim3(256), 0, s, (const T*)x.p, (T*)y.p,
                                      x.n, x.h, x.w, x.c, x.ld, y.h, y.w, y.ld));
  HIP_CHECK(hipGetLastError());
}


// ------------------------------------------------------------------------------------------ avg-pool 2x2 s2
// nn.AvgPool2d(2, 2, 0, ceil_mode=True) of the ResNet-vd shortcut (HF:rt_detr_resnet.py:199-205); extents are
// even here (input sizes are multiples of 32), so every window is a full 2x2.  8 channels per thread.
template <typename T>
__global__ void k_avgpool2(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int C, int64_t ldx, int64_t ldy) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  constexpr int V = 16 / (int)sizeof(T);
  const int cv = C / V;
  const int OH = H / 2, OW = W / 2;
  const int64_t total = (int64_t)B * OH * OW * cv;
  if (i >= total) return;
  const int cc = (int)(i % cv) * V;
  int64_t p = i / cv;
  const int ox = (int)(p % OW); p /= OW;
  const int oy = (int)(p % OH);
  const int b = (int)(p / OH);
  const T* q = x + (((int64_t)b * H + 2 * oy) * W + 2 * ox) * ldx + cc;
  typedef T VT __attribute__((ext_vector_type(V)));
  const VT a0 = *(const VT*)q, a1 = *(const VT*)(q + ldx), a2 = *(const VT*)(q + (int64_t)W * ldx), a3 = *(const VT*)(q + (int64_t)(W + 1) * ldx);
  VT o;
#pragma unroll
  for (int k = 0; k < V; ++k) o[k] = (T)((((float)a0[k] + (float)a1[k]) + ((float)a2[k] + (float)a3[k])) * 0.25f);
  *(VT*)(y + (((int64_t)b * OH + oy) * OW + ox) * ldy + cc) = o;
}
__global__ void k_avgpool2_split(const sp16* __restrict__ x, sp16* __restrict__ y, int B, int H, int W, int C, int64_t ldx, int64_t ldy) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int cv = C / 8;
  const int OH = H / 2, OW = W / 2;
  const int64_t total = (int64_t)B * OH * OW * cv;
  if (i >= total) return;
  const int cc = (int)(i % cv) * 8;
  int64_t p = i / cv;
  const int ox = (int)(p % OW); p /= OW;
  const int oy = (int)(p % OH);
  const int b = (int)(p / OH);
  const int64_t q = (((int64_t)b * H + 2 * oy) * W + 2 * ox) * ldx;
  float a0[8], a1[8], a2[8], a3[8], o[8];
  split_load8(x, q, cc, a0); split_load8(x, q + ldx, cc, a1); split_load8(x, q + (int64_t)W * ldx, cc, a2); split_load8(x, q + (int64_t)(W + 1) * ldx, cc, a3);
#pragma unroll
  for (int k = 0; k < 8; ++k) o[k] = ((a0[k] + a1[k]) + (a2[k] + a3[k])) * 0.25f;
  split_store8(y, (((int64_t)b * OH + oy) * OW + ox) * ldy, cc, o);
}
void launch_avgpool2(const Tensor& x, const Tensor& y, hipStream_t s) {
  if (x.dt == F16X2) {
    RTD_CHECK(y.dt == F16X2 && x.c == y.c && x.c % SPLIT_GROUP == 0 && x.ld % SPLIT_GROUP == 0 && y.ld % SPLIT_GROUP == 0 && x.n == y.n, 1, "avgpool: split layout");
    RTD_CHECK(x.h % 2 == 0 && x.w % 2 == 0 && y.h == x.h / 2 && y.w == x.w / 2, 1, "avgpool: even extents only");
    RTD_CHECK(x.bstride == (int64_t)x.h * x.w * x.ld && y.bstride == (int64_t)y.h * y.w * y.ld && (((uintptr_t)x.p | (uintptr_t)y.p) & 15) == 0, 1, "avgpool: dense images");
    const int64_t total = (int64_t)y.n * y.h * y.w * (y.c / 8);
    rtd_launch(k_avgpool2_split, dim3(blocks_for(total, 256)), dim3(256), 0, s, (const sp16*)x.p, (sp16*)y.p, x.n, x.h, x.w, x.c, x.ld, y.ld);
    HIP_CHECK(hipGetLastError());
    return;
  }
  const int V = x.dt == BF16 ? 8 : 4;
  RTD_CHECK(x.dt == y.dt && x.c == y.c && x.c % V == 0 && x.ld % V == 0 && y.ld % V == 0 && x.n == y.n, 1, "avgpool: dtype/channels");
  RTD_CHECK(x.h % 2 == 0 && x.w % 2 == 0 && y.h == x.h / 2 && y.w == x.w / 2, 1, "avgpool: even extents only");
  RTD_CHECK(x.bstride == (int64_t)x.h * x.w * x.ld && y.bstride == (int64_t)y.h * y.w * y.ld, 1, "avgpool: dense images");
  RTD_CHECK((((uintptr_t)x.p | (uintptr_t)y.p) & 15) == 0, 1, "avgpool: alignment");
  const int64_t total = (int64_t)y.n * y.h * y.w * (y.c / V);
  DISPATCH_T(x.dt, rtd_launch(k_avgpool2<T>, dim3(blocks_for(total, 256)), dim3(256), 0, s, (const T*)x.p, (T*)y.p, x.n,
                                      x.h, x.w, x.c, x.ld, y.ld));
  HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------ nearest 2x upsample
// F.interpolate(scale_factor=2, mode="nearest") (HF:v2.py:1191), written straight into the first
// channel half of the FPN concat buffer (y is a channel-slice view).
template <typename T>
__global__ void k_upsample2x(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int C, int64_t ldx,
                             int64_t ldy) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int c4 = C / 4;
  const int OH = 2 * H, OW = 2 * W;
  const int64_t total = (int64_t)B * OH * OW * c4;
  if (i >= total) return;
  const int cc = (int)(i % c4) * 4;
  int64_t p = i / c4;
  const int ox = (int)(p % OW); p /= OW;
  const int oy = (int)(p % OH);
  const int b = (int)(p / OH);
  const T* q = x + (((int64_t)b * H + (oy >> 1)) * W + (ox >> 1)) * ldx + cc;
  T* o = y + (((int64_t)b * OH + oy) * OW + ox) * ldy + cc;
#pragma unroll
  for (int k = 0; k < 4; ++k) o[k] = q[k];
}
void launch_upsample2x(const Tensor& x, const Tensor& y, hipStream_t s) {
  RTD_CHECK(x.dt == y.dt && x.c == y.c && x.c % 4 == 0 && y.h == 2 * x.h && y.w == 2 * x.w && x.n == y.n, 1, "upsample: shape");
  RTD_CHECK(x.bstride == (int64_t)x.h * x.w * x.ld && y.bstride == (int64_t)y.h * y.w * y.ld, 1, "upsample: dense images");
  const int64_t total = (int64_t)y.n * y.h * y.w * (y.c / 4);
  DISPATCH_T(x.dt, rtd_launch(k_upsample2x<T>, dim3(blocks_for(total, 256)), dim3(256), 0, s, (const T*)x.p,
                                      (T*)y.p, x.n, x.h, x.w, x.c, x.ld, y.ld));
  HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------ multi-head attention
// HF:v2.py:246-270 eager attention: softmax(q k^T * hd^-0.5) v, no mask.  One thread owns one query
// row (q and the output accumulator live in registers, fp32), keys/values are staged through LDS in
// tiles of 64 and read as wave-wide broadcasts; online softmax.  L is 300..1600 here, so the whole
// problem is a few GFLOP - fp32 VALU keeps the decoder in exact-fp32 territory.
template <typename T, int HD>
__global__ __launch_bounds__(256) void k_attention(const T* __restrict__ qk, int64_t ldqk, const T* __restrict__ v,
                                                    int64_t ldv, T* __restrict__ o, int64_t ldo, int L, int D) {
  // 4 lanes share one query row: lane part p takes keys p, p+4, ... of every staged tile and the
  // four partial (max, sum, acc) states are merged with two xor-shuffles at the end.
  constexpr int KT = 64, KS = 4, QPB = 256 / KS;
  constexpr int LDK = HD + 4;                     // padded rows: the 4 parts read 4 different rows per instruction
  __shared__ __attribute__((aligned(16))) float Ks[KT * LDK];
  __shared__ __attribute__((aligned(16))) float Vs[KT * LDK];
  const int b = blockIdx.z, head = blockIdx.y;
  const int part = threadIdx.x & (KS - 1);
  const int qi = blockIdx.x * QPB + (threadIdx.x >> 2);
  const bool active = qi < L;
  const float scale = rsqrtf((float)HD);
  float q[HD], acc[HD];
  const T* qrow = qk + ((int64_t)b * L + (active ? qi : 0)) * ldqk + head * HD;
#pragma unroll
  for (int d = 0; d < HD; ++d) {
    q[d] = (float)qrow[d] * scale;
    acc[d] = 0.f;
  }
  float m = -INFINITY, l = 0.f;
  for (int k0 = 0; k0 < L; k0 += KT) {
    __syncthreads();
    for (int e = threadIdx.x; e < KT * HD; e += 256) {
      const int j = e / HD, d = e - j * HD;
      const int kj = k0 + j;
      float kv = 0.f, vv = 0.f;
      if (kj < L) {
        kv = (float)qk[((int64_t)b * L + kj) * ldqk + D + head * HD + d];
        vv = (float)v[((int64_t)b * L + kj) * ldv + head * HD + d];
      }
      Ks[j * LDK + d] = kv;
      Vs[j * LDK + d] = vv;
    }
    __syncthreads();
    const int jn = min(KT, L - k0);
    for (int j = part; j < jn; j += KS) {
      const float* kr = &Ks[j * LDK];
      const float* vr = &Vs[j * LDK];
      float sc = 0.f;
#pragma unroll
      for (int d = 0; d < HD; ++d) sc = fmaf(q[d], kr[d], sc);
      if (sc > m) {
        const float f = __expf(m - sc);
        l *= f;
#pragma unroll
        for (int d = 0; d < HD; ++d) acc[d] *= f;
        m = sc;
      }
      const float p = __expf(sc - m);
      l += p;
#pragma unroll
      for (int d = 0; d < HD; ++d) acc[d] = fmaf(p, vr[d], acc[d]);
    }
  }
  // merge the 4 partial softmax states (a part that saw no key has m = -inf, l = 0)
#pragma unroll
  for (int off = 1; off < KS; off <<= 1) {
    const float mo = __shfl_xor(m, off, 64), lo = __shfl_xor(l, off, 64);
    const float mn = fmaxf(m, mo);
    const float fa = (m == -INFINITY) ? 0.f : __expf(m - mn);
    const float fb = (mo == -INFINITY) ? 0.f : __expf(mo - mn);
    l = l * fa + lo * fb;
#pragma unroll
    for (int d = 0; d < HD; ++d) acc[d] = acc[d] * fa + __shfl_xor(acc[d], off, 64) * fb;
    m = mn;
  }
  if (active) {
    const float inv = 1.f / l;
    T* orow = o + ((int64_t)b * L + qi) * ldo + head * HD;
    // each of the 4 lanes writes a quarter of the row
#pragma unroll
    for (int d = 0; d < HD; ++d)
      if (d / (HD / KS) == part) orow[d] = (T)(acc[d] * inv);    // static register index, predicated store
  }
}

// The same attention on the matrix cores, exact fp32 (v_mfma_f32_16x16x4_f32 = an fp32 fma chain): the long-sequence form (AIFI of the
// wide encoders: 1600 tokens x 384 channels at 1280 px ran 0.67 ms per step on the VALU kernel above, 23 TFLOP/s).  A block = 4 waves = 64
// queries of one (image, head); keys / values are staged 64 at a time in LDS (coalesced 16-byte loads) and every wave runs decoder.hip's
// transposed scheme on its 16 queries: S^T = K Q^T per 16-key tile (keys on the accumulator rows, the lane's query on the column), the
// softmax reductions over keys are 4 registers + one row-swap reduction, and the accumulator tile of P^T is already the B operand of
// O^T += V^T P^T.
template <int HD>
__global__ __launch_bounds__(256) void k_attention_mfma_f32(const float* __restrict__ qk, int64_t ldqk, const float* __restrict__ v, int64_t ldv,
                                                             float* __restrict__ o, int64_t ldo, int L, int D) {
  constexpr int KT = 64, LDK = HD + 4, NC = HD / 16;
  static_assert(HD % 16 == 0, "head dim");
  __shared__ __attribute__((aligned(16))) float Ks[KT * LDK];
  __shared__ __attribute__((aligned(16))) float Vs[KT * LDK];
  const int b = blockIdx.z, head = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, q = lane >> 4;
  const int qi = blockIdx.x * 64 + wave * 16 + r16;
  const float scale = rsqrtf((float)HD);
  f32x4 qa[NC];
  {
    const float* qrow = qk + ((int64_t)b * L + min(qi, L - 1)) * ldqk + head * HD + 4 * q;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      qa[c] = *(const f32x4*)(qrow + 16 * c);
#pragma unroll
      for (int u = 0; u < 4; ++u) qa[c][u] *= scale;
    }
  }
  float m = -INFINITY, l = 0.f;
  f32x4 O[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) O[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* kbase = qk + (int64_t)b * L * ldqk + D + head * HD;
  const float* vbase = v + (int64_t)b * L * ldv + head * HD;
  for (int k0 = 0; k0 < L; k0 += KT) {
    __syncthreads();
    for (int e = tid; e < KT * (HD / 4); e += 256) {
      const int j = e / (HD / 4), d4 = (e - j * (HD / 4)) * 4;
      const int kj = k0 + j;
      f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = kv;
      if (kj < L) { kv = *(const f32x4*)(kbase + (int64_t)kj * ldqk + d4); vv = *(const f32x4*)(vbase + (int64_t)kj * ldv + d4); }
      *(f32x4*)&Ks[j * LDK + d4] = kv;
      *(f32x4*)&Vs[j * LDK + d4] = vv;
    }
    __syncthreads();
    const int nsub = (min(KT, L - k0) + 15) >> 4;
    for (int sub = 0; sub < nsub; ++sub) {
      f32x4 S = {0.f, 0.f, 0.f, 0.f};                           // S^T[key 4 q + r][query r16]
      const float* kr = &Ks[(sub * 16 + r16) * LDK + 4 * q];
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const f32x4 kf = *(const f32x4*)(kr + 16 * c);
#pragma unroll
        for (int u = 0; u < 4; ++u) S = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[u], qa[c][u], S, 0, 0, 0);
      }
      float sv[4];
      float mx = -INFINITY;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        sv[r] = (k0 + sub * 16 + 4 * q + r < L) ? S[r] : -INFINITY;
        mx = fmaxf(mx, sv[r]);
      }
      mx = rows4_max(mx);
      const float mn = fmaxf(m, mx);
      const float alpha = __expf(m - mn);
      f32x4 pr;
      float rs = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) { pr[r] = __expf(sv[r] - mn); rs += pr[r]; }
      rs = rows4_sum(rs);
      l = l * alpha + rs;
      m = mn;
      const float* vr = &Vs[(sub * 16 + 4 * q) * LDK + r16];
#pragma unroll
      for (int c = 0; c < NC; ++c) {
#pragma unroll
        for (int r = 0; r < 4; ++r) O[c][r] *= alpha;
#pragma unroll
        for (int u = 0; u < 4; ++u) O[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(vr[u * LDK + 16 * c], pr[u], O[c], 0, 0, 0);
      }
    }
  }
  if (qi < L) {
    const float inv = 1.f / l;
    float* orow = o + ((int64_t)b * L + qi) * ldo + head * HD + 4 * q;
#pragma unroll
    for (int c = 0; c < NC; ++c) *(f32x4*)(orow + 16 * c) = f32x4{O[c][0] * inv, O[c][1] * inv, O[c][2] * inv, O[c][3] * inv};
  }
}

void launch_attention(const Tensor& qk, const Tensor& v, const Tensor& o, int heads, hipStream_t s) {
  const int B = qk.n, L = qk.h * qk.w, D = v.c;
  RTD_CHECK(qk.c == 2 * D && o.c == D && v.n == B && o.n == B && v.h * v.w == L && o.h * o.w == L, 1, "attention: shape");
  RTD_CHECK(qk.dt == v.dt && qk.dt == o.dt, 1, "attention: dtype");
  RTD_CHECK(D % heads == 0, 1, "attention: heads");
  const int hd = D / heads;
  const dim3 grid((L + 63) / 64, heads, B), blk(256);
  if (qk.dt == F32 && L >= 64 && (hd == 32 || hd == 48 || hd == 64) && qk.ld % 4 == 0 && v.ld % 4 == 0 && o.ld % 4 == 0 &&
      (((uintptr_t)qk.p | (uintptr_t)v.p | (uintptr_t)o.p) & 15) == 0) {
    if (hd == 32) rtd_launch(k_attention_mfma_f32<32>, grid, blk, 0, s, (const float*)qk.p, qk.ld, (const float*)v.p, v.ld, (float*)o.p, o.ld, L, D);
    else if (hd == 48) rtd_launch(k_attention_mfma_f32<48>, grid, blk, 0, s, (const float*)qk.p, qk.ld, (const float*)v.p, v.ld, (float*)o.p, o.ld, L, D);
    else rtd_launch(k_attention_mfma_f32<64>, grid, blk, 0, s, (const float*)qk.p, qk.ld, (const float*)v.p, v.ld, (float*)o.p, o.ld, L, D);
    HIP_CHECK(hipGetLastError());
    return;
  }
#define ATT_GO(HD) DISPATCH_T(qk.dt, rtd_launch((k_attention<T, HD>), grid, blk, 0, s, (const T*)qk.p, qk.ld, (const T*)v.p, v.ld, (T*)o.p, o.ld, L, D))
  if (hd == 32) ATT_GO(32);
  else if (hd == 48) ATT_GO(48);
  else if (hd == 64) ATT_GO(64);
  else RTD_CHECK(false, 1, "attention: head dim must be 32, 48 or 64");
#undef ATT_GO
  HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------ masked-anchor rows
// memory = valid_mask * src  (HF:v2.py:1583): a masked row makes enc_output's Linear return its bias.
template <typename T>
__global__ void k_set_rows(T* __restrict__ y, int64_t ld, int C, const int32_t* __restrict__ rows, int nrows,
                           int rows_per_image, const float* __restrict__ vec, int B) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)B * nrows * C;
  if (i >= total) return;
  const int c = (int)(i % C);
  const int64_t t = i / C;
  const int r = rows[t % nrows];
  const int b = (int)(t / nrows);
  y[((int64_t)b * rows_per_image + r) * ld + c] = (T)vec[c];
}
void launch_set_rows(const Tensor& y, const int32_t* rows, int nrows, int rows_per_image, const float* vec, hipStream_t s) {
  if (nrows == 0) return;
  const int64_t total = (int64_t)y.n * nrows * y.c;
  DISPATCH_T(y.dt, rtd_launch(k_set_rows<T>, dim3(blocks_for(total, 256)), dim3(256), 0, s, (T*)y.p, y.ld, y.c, rows,
                                      nrows, rows_per_image, vec, y.n));
  HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------ row max
// enc_outputs_class.max(-1).values (HF:v2.py:1590); 16 lanes per row.
__global__ void k_rowmax(const float* __restrict__ x, int64_t ld, int C, int64_t rows, float* __restrict__ out) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t row = gid >> 4;
  const int j = (int)(gid & 15);
  float m = -INFINITY;
  if (row < rows)
    for (int c = j; c < C; c += 16) m = fmaxf(m, x[row * ld + c]);
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if (row < rows && j == 0) out[row] = m;
}
void launch_rowmax(const Tensor& x, float* out, hipStream_t s) {
  RTD_CHECK(x.dt == F32, 1, "rowmax: fp32 logits expected");
  const int64_t rows = x.pixels();
  rtd_launch(k_rowmax, dim3(blocks_for(rows * 16, 256)), dim3(256), 0, s, (const float*)x.p, x.ld, x.c, rows, out);
  HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------ exact top-k
// torch.topk(keys, K, dim=1) for K <= 1024 (HF:v2.py:1590 and image_processing_rt_detr.py:527):
// one workgroup per image; 4-pass 8-bit radix select of the K-th largest key, deterministic
// compaction (ties: lowest index first), bitonic sort of the K survivors by (key desc, index asc).
__device__ __forceinline__ unsigned f2key(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k) {
  const unsigned u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __uint_as_float(u);
}

// REG = true: every thread keeps its <= 32 keys in registers (N <= 32768), so the four radix passes and the
// compaction touch global memory once; REG = false re-reads the keys from global (any N).
// POST (REG only): the whole post-processor of HF:rt_detr/image_processing_rt_detr.py:510-533 in this launch - the keys are
// sigmoid(logits[q][c]) made while they are loaded (index = q * C + c) and a winner is written as its finished row [label, score, x1, y1,
// x2, y2] (label = index % C, query = index / C, cxcywh -> xyxy, scaled to the original frame): the score tensor, the index / value
// vectors and two launches (k_pp_scores, k_pp_gather) disappear.  Same keys, same tie rule, same arithmetic per row as the three-launch form.
struct TopkPost {
  const float* ref8;       // [B][Q][8] cxcywh boxes of the last decoder layer
  const float* scale_wh;   // [B][2] original frame (w, h)
  float* block6;           // [B][K][6]
  int C, Q;
};
__device__ __forceinline__ float topk_sigmoid(float x) { return 1.f / (1.f + __expf(-x)); }   // == sigmoidf_ below (k_pp_scores)
template <bool REG, bool POST = false>
__global__ __launch_bounds__(1024) void k_topk(const float* __restrict__ keys, int N, int K, int32_t* __restrict__ idx_out,
                                                float* __restrict__ val_out, const TopkPost pa) {
  static_assert(!POST || REG, "the fused post-processor keeps its keys in registers");
  auto emit = [&](unsigned r, unsigned long long e) {            // rank r of this image: composite (key << 32 | ~index)
    const int idx = (int)(0xffffffffu - (unsigned)(e & 0xffffffffu));
    const float val = key2f((unsigned)(e >> 32));
    if (POST) {
      const int bimg = blockIdx.x;
      const int label = idx % pa.C;
      int q = idx / pa.C;
      q = min(max(q, 0), pa.Q - 1);
      const float* rr = pa.ref8 + ((int64_t)bimg * pa.Q + q) * 8;
      const float cx = rr[0], cy = rr[1], w = rr[2], h = rr[3];
      const float sw = pa.scale_wh[bimg * 2 + 0], sh = pa.scale_wh[bimg * 2 + 1];
      float* o = pa.block6 + ((int64_t)bimg * K + r) * 6;
      o[0] = (float)label;
      o[1] = val;
      o[2] = (cx - 0.5f * w) * sw;
      o[3] = (cy - 0.5f * h) * sh;
      o[4] = (cx + 0.5f * w) * sw;
      o[5] = (cy + 0.5f * h) * sh;
    } else {
      idx_out[(int64_t)blockIdx.x * K + r] = idx;
      if (val_out) val_out[(int64_t)blockIdx.x * K + r] = val;
    }
  };
  constexpr int MAXPT = 32;
  // one histogram per wave: scores of one image share their high key bits, so a block-wide histogram serialises ~N LDS atomics on a
  // handful of addresses per pass (24000 keys: ~10 us per pass); per wave the same-address adds of one instruction cost ~64 cycles
  __shared__ unsigned whist[16][256];
  __shared__ unsigned hist[256];
  __shared__ unsigned long long sel[1024];
  __shared__ unsigned s_prefix, s_krem, s_cnt_gt, s_cnt_eq, s_cnt_T;
  __shared__ unsigned wave_cnt[16];
  const int tid = threadIdx.x;
  const float* kb = keys + (int64_t)blockIdx.x * N;
  unsigned kreg[MAXPT];
  const int npt = REG ? (N + 1023) / 1024 : 0;
  if (REG) {
#pragma unroll
    for (int j = 0; j < MAXPT; ++j) {
      const int i = tid + j * 1024;
      kreg[j] = (j < npt && i < N) ? f2key(POST ? topk_sigmoid(kb[i]) : kb[i]) : 0u;         // slots beyond N are never counted (guarded by i < N)
    }
  }
  // ---- fast path (round 5): bound the candidates before any pass over all the keys ----------------------------------------------
  // The keys are thrown into 1024 BAGS by a multiplicative hash of their index; the K-th largest of the bag maxima, L, is a lower bound
  // of the K-th largest key (the K best bags each hold a different key >= L), so the answer lies among the keys >= L - about 1.1 K of
  // them (24 000 keys, K = 300: ~340) whatever structure the index carries (per-thread strides meet the post-processor's class-major
  // layout: every 16th thread holds a class; contiguous groups meet the encoder's spatially clustered scores; a hash meets neither).
  // L costs one LDS atomic-max per key on scattered addresses and a radix select over ONE value per thread (4 passes of 1024 LDS
  // atomics instead of 24 576: a pass over all the keys of one image serialises 64-way on the handful of exponent buckets that scores
  // share, ~12 us for the first pass alone); the candidates are compacted with one atomic per wave and key slot and ordered by
  // counting, for every candidate, the candidates above it (the 64-bit (key, ~index) composites are distinct: ranks are exact, ties fall
  // lowest-index-first as before).  More than 1024 candidates (heavy ties at the cut, e.g. masked anchors sharing one score) fall
  // through to the general path below.
  {
    unsigned* bag = (unsigned*)sel;                                // the first 4 KB of sel (re-zeroed below)
    bag[tid] = 0u;
    __syncthreads();
    auto bag_of = [](int i) -> unsigned { return ((unsigned)i * 2654435761u) >> 22; };
    if (REG) {
#pragma unroll
      for (int j = 0; j < MAXPT; ++j) {
        const int i = tid + j * 1024;
        if (j < npt && i < N) atomicMax(&bag[bag_of(i)], kreg[j]);
      }
    } else {
      for (int i = tid; i < N; i += 1024) atomicMax(&bag[bag_of(i)], f2key(kb[i]));
    }
    __syncthreads();
    const unsigned tmax = bag[tid];
    __syncthreads();
    if (tid == 0) { s_prefix = 0; s_krem = (unsigned)K; s_cnt_gt = 0; }
    sel[tid] = 0ull;
    __syncthreads();
    for (int pass = 0; pass < 4; ++pass) {
      const int shift = 24 - 8 * pass;
      unsigned* mine = whist[tid >> 6];
      for (int b = tid & 63; b < 256; b += 64) mine[b] = 0;
      __builtin_amdgcn_wave_barrier();
      const unsigned prefix = s_prefix;
      const unsigned mask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
      if ((tmax & mask) == prefix) atomicAdd(&mine[(tmax >> shift) & 255u], 1u);
      __syncthreads();
      if (tid < 256) {
        unsigned t = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += whist[w][tid];
        hist[tid] = t;
      }
      __syncthreads();
      if (tid < 64) {
        const int l = tid;
        const unsigned c0 = hist[255 - 4 * l], c1 = hist[254 - 4 * l], c2 = hist[253 - 4 * l], c3 = hist[252 - 4 * l];
        const unsigned tot = c0 + c1 + c2 + c3;
        unsigned incl = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const unsigned t = __shfl_up(incl, o, 64);
          if (l >= o) incl += t;
        }
        const unsigned excl = incl - tot;
        const unsigned rem = s_krem;
        if (excl < rem && rem <= incl) {
          unsigned r = rem - excl, d;
          if (r <= c0) d = 255 - 4 * l;
          else if ((r -= c0) <= c1) d = 254 - 4 * l;
          else if ((r -= c1) <= c2) d = 253 - 4 * l;
          else { r -= c2; d = 252 - 4 * l; }
          s_krem = r;
          s_prefix = prefix | (d << shift);
        }
      }
      __syncthreads();
    }
    const unsigned Lb = s_prefix;                                  // K-th largest thread maximum
    const int lane_ = tid & 63;
    auto offer = [&](unsigned k, int i, bool has) {                // wave-aggregated compaction of the keys >= Lb
      const bool take = has && k >= Lb;
      const unsigned long long bal = __ballot(take);
      if (bal) {
        unsigned base = 0;
        if (lane_ == (int)__builtin_ctzll(bal)) base = atomicAdd(&s_cnt_gt, (unsigned)__popcll(bal));
        base = __shfl(base, (int)__builtin_ctzll(bal), 64);
        const unsigned pos = base + (unsigned)__popcll(bal & ((1ull << lane_) - 1ull));
        if (take && pos < 1024) sel[pos] = ((unsigned long long)k << 32) | (unsigned)(0xffffffffu - (unsigned)i);
      }
    };
    if (REG) {
#pragma unroll
      for (int j = 0; j < MAXPT; ++j) {
        const int i = tid + j * 1024;
        if (j < npt) offer(kreg[j], i, i < N);                      // (j < npt is block-uniform: every lane of a wave takes part in the ballot)
      }
    } else {
      for (int base_i = 0; base_i < N; base_i += 1024) {
        const int i = base_i + tid;
        offer(i < N ? f2key(kb[i]) : 0u, i, i < N);
      }
    }
    __syncthreads();
    const unsigned C = s_cnt_gt;
    if (C <= 1024u) {                                               // block-uniform
      if ((unsigned)tid < C) {
        const unsigned long long my = sel[tid];
        unsigned r = 0;
        for (unsigned j = 0; j < C; ++j) r += sel[j] > my ? 1u : 0u;
        if (r < (unsigned)K) emit(r, my);
      }
      return;
    }
    __syncthreads();                                                // every wave has read C before the general path resets the counters
  }
  if (tid == 0) { s_prefix = 0; s_krem = (unsigned)K; s_cnt_gt = 0; s_cnt_eq = 0; }
  sel[tid] = 0ull;
  __syncthreads();
  // ---- radix select: after the loop s_prefix is the key of the K-th largest element ------------
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    unsigned* mine = whist[tid >> 6];
    for (int b = tid & 63; b < 256; b += 64) mine[b] = 0;
    __builtin_amdgcn_wave_barrier();
    const unsigned prefix = s_prefix;
    const unsigned mask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
    if (REG) {
#pragma unroll
      for (int j = 0; j < MAXPT; ++j) {
        const int i = tid + j * 1024;
        if (j < npt && i < N && (kreg[j] & mask) == prefix) atomicAdd(&mine[(kreg[j] >> shift) & 255u], 1u);
      }
    } else {
      for (int i = tid; i < N; i += 1024) {
        const unsigned k = f2key(kb[i]);
        if ((k & mask) == prefix) atomicAdd(&mine[(k >> shift) & 255u], 1u);
      }
    }
    __syncthreads();
    if (tid < 256) {
      unsigned t = 0;
#pragma unroll
      for (int w = 0; w < 16; ++w) t += whist[w][tid];
      hist[tid] = t;
    }
    __syncthreads();
    if (tid < 64) {
      // wave 0 scans the 256 buckets from the top: lane l owns buckets 255-4l .. 252-4l
      const int l = tid;
      const unsigned c0 = hist[255 - 4 * l], c1 = hist[254 - 4 * l], c2 = hist[253 - 4 * l], c3 = hist[252 - 4 * l];
      const unsigned tot = c0 + c1 + c2 + c3;
      unsigned incl = tot;                                       // inclusive prefix over lanes (descending buckets)
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_up(incl, o, 64);
        if (l >= o) incl += t;
      }
      const unsigned excl = incl - tot;
      const unsigned rem = s_krem;
      if (excl < rem && rem <= incl) {                           // exactly one lane holds the target bucket
        unsigned r = rem - excl, d;
        if (r <= c0) d = 255 - 4 * l;
        else if ((r -= c0) <= c1) d = 254 - 4 * l;
        else if ((r -= c1) <= c2) d = 253 - 4 * l;
        else { r -= c2; d = 252 - 4 * l; }
        s_krem = r;
        s_prefix = prefix | (d << shift);
        if (pass == 3) s_cnt_T = hist[d];                          // how many keys equal the K-th largest exactly
      }
    }
    __syncthreads();
  }
  const unsigned T = s_prefix;
  const unsigned need_eq = s_krem;      // how many elements equal to T are taken (lowest indices)
  // ---- compaction: strictly greater (any order), then equal in index order ---------------------
  if (REG) {
#pragma unroll
    for (int j = 0; j < MAXPT; ++j) {
      const int i = tid + j * 1024;
      if (j < npt && i < N && kreg[j] > T) {
        const unsigned pos = atomicAdd(&s_cnt_gt, 1u);
        if (pos < 1024) sel[pos] = ((unsigned long long)kreg[j] << 32) | (unsigned)(0xffffffffu - (unsigned)i);
      }
    }
  } else {
    for (int i = tid; i < N; i += 1024) {
      const unsigned k = f2key(kb[i]);
      if (k > T) {
        const unsigned pos = atomicAdd(&s_cnt_gt, 1u);
        if (pos < 1024) sel[pos] = ((unsigned long long)k << 32) | (unsigned)(0xffffffffu - (unsigned)i);
      }
    }
  }
  __syncthreads();
  const unsigned n_gt = s_cnt_gt;       // == K - need_eq
  const int lane = tid & 63, wv = tid >> 6;
  if (s_cnt_T == need_eq) {
    // the usual case - no tie straddles the cut: every key equal to T is taken, in any order (the sort below orders by (key, index))
    if (REG) {
#pragma unroll
      for (int j = 0; j < MAXPT; ++j) {
        const int i = tid + j * 1024;
        if (j < npt && i < N && kreg[j] == T) {
          const unsigned pos = n_gt + atomicAdd(&s_cnt_eq, 1u);
          if (pos < 1024) sel[pos] = ((unsigned long long)T << 32) | (unsigned)(0xffffffffu - (unsigned)i);
        }
      }
    } else {
      for (int i = tid; i < N; i += 1024) {
        if (f2key(kb[i]) == T) {
          const unsigned pos = n_gt + atomicAdd(&s_cnt_eq, 1u);
          if (pos < 1024) sel[pos] = ((unsigned long long)T << 32) | (unsigned)(0xffffffffu - (unsigned)i);
        }
      }
    }
  } else
  for (int base = 0, j = 0; base < N; base += 1024, ++j) {
    const int i = base + tid;
    bool eq = false;
    if (i < N) {
      if (REG) {
        unsigned kv = 0;
#pragma unroll
        for (int jj = 0; jj < MAXPT; ++jj) kv = (jj == j) ? kreg[jj] : kv;   // static register indices
        eq = kv == T;
      } else {
        eq = f2key(kb[i]) == T;
      }
    }
    const unsigned long long bal = __ballot(eq);
    if (lane == 0) wave_cnt[wv] = (unsigned)__popcll(bal);
    __syncthreads();
    unsigned before = s_cnt_eq;
    for (int w = 0; w < wv; ++w) before += wave_cnt[w];
    const unsigned my = before + (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
    if (eq && my < need_eq && n_gt + my < 1024)
      sel[n_gt + my] = ((unsigned long long)T << 32) | (unsigned)(0xffffffffu - (unsigned)i);
    __syncthreads();
    if (tid == 0) {
      unsigned tot = 0;
      for (int w = 0; w < 16; ++w) tot += wave_cnt[w];
      s_cnt_eq += tot;
    }
    __syncthreads();
    if (s_cnt_eq >= need_eq) break;
  }
  __syncthreads();
  // ---- bitonic sort, descending (unused slots are 0 and sink to the end): 512 slots when K fits, else 1024.  Exchanges at distance
  // j < 64 stay inside one wave's 64 slots (LDS executes a wave's accesses in order): only the j >= 64 steps need the block barrier -
  // 6 (512 slots) or 10 (1024) instead of 45 / 55
  const int SN = K <= 512 ? 512 : 1024;
  for (int k = 2; k <= SN; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j >= 64) __syncthreads();                                // the partner slot was last written by another wave
      const int ixj = tid ^ j;
      if (tid < SN && ixj > tid) {
        const unsigned long long a = sel[tid], b = sel[ixj];
        const bool desc = (tid & k) == 0;
        if (desc ? (a < b) : (a > b)) { sel[tid] = b; sel[ixj] = a; }
      }
      if (j >= 64) __syncthreads();
      else __builtin_amdgcn_wave_barrier();
    }
  }
  __syncthreads();
  if (tid < K) emit((unsigned)tid, sel[tid]);
}
void launch_topk(const float* keys, int B, int N, int K, int32_t* idx, float* vals, hipStream_t s) {
  RTD_CHECK(K >= 1 && K <= 1024 && K <= N, 1, "topk: K must be in [1, min(1024, N)]");
  const TopkPost none = {nullptr, nullptr, nullptr, 0, 0};
  if (N <= 32768) rtd_launch((k_topk<true, false>), dim3(B), dim3(1024), 0, s, keys, N, K, idx, vals, none);
  else rtd_launch((k_topk<false, false>), dim3(B), dim3(1024), 0, s, keys, N, K, idx, vals, none);
  HIP_CHECK(hipGetLastError());
}
// the post-processor in one launch (TopkPost); false when the shape needs the three-launch form (dense logits rows, Q * C keys in registers)
bool launch_postprocess_fused(const Tensor& logits, const float* ref8, const float* scale_wh, int B, int Q, float* block6, hipStream_t s) {
  const int C = logits.c;
  if (logits.dt != F32 || logits.ld != C || (int64_t)Q * C > 32768 || Q > 1024 || logits.pixels() != (int64_t)B * Q) return false;
  const TopkPost pa = {ref8, scale_wh, block6, C, Q};
  rtd_launch((k_topk<true, true>), dim3(B), dim3(1024), 0, s, (const float*)logits.p, Q * C, Q, (int32_t*)nullptr, (float*)nullptr, pa);
  HIP_CHECK(hipGetLastError());
  return true;
}

// ------------------------------------------------------------------------------------------ gather rows
// target = output_memory.gather(topk_ind) (HF:v2.py:1609)
template <typename TS, typename TD>
__global__ void k_gather_rows(const TS* __restrict__ src, int64_t lds_, int rows_per_image, const int32_t* __restrict__ idx,
                              int Q, int C, TD* __restrict__ dst, int64_t ldd, int B) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)B * Q * C;
  if (i >= total) return;
  const int c = (int)(i % C);
  const int64_t t = i / C;
  const int b = (int)(t / Q);
  int r = idx[t];
  r = min(max(r, 0), rows_per_image - 1);
  dst[t * ldd + c] = (TD)(float)src[((int64_t)b * rows_per_image + r) * lds_ + c];
}
void launch_gather_rows(const Tensor& src, const int32_t* idx, int rows_per_image, const Tensor& dst, hipStream_t s) {
  const int B = dst.n, Q = dst.h * dst.w, C = dst.c;
  RTD_CHECK(src.c == C && src.n == B && src.h * src.w == rows_per_image, 1, "gather: shape");
  const int64_t total = (int64_t)B * Q * C;
  const dim3 grid(blocks_for(total, 256)), blk(256);
#define G_GO(TS, TD) rtd_launch((k_gather_rows<TS, TD>), grid, blk, 0, s, (const TS*)src.p, src.ld, rows_per_image, idx, Q, C, (TD*)dst.p, dst.ld, B)
  if (src.dt == BF16 && dst.dt == BF16) G_GO(bf16, bf16);
  else if (src.dt == BF16) G_GO(bf16, float);
  else if (dst.dt == BF16) G_GO(float, bf16);
  else G_GO(float, float);
#undef G_GO
  HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------ reference boxes
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }
__device__ __forceinline__ float inv_sigmoid(float x) {   // HF:v2.py:548-552, eps 1e-5
  x = fminf(fmaxf(x, 0.f), 1.f);
  const float x1 = fmaxf(x, 1e-5f), x2 = fmaxf(1.f - x, 1e-5f);
  return __logf(x1 / x2);
}
// reference_points_unact = enc_bbox_head(target) + anchors[topk] ; reference = sigmoid(.)  (HF:v2.py:1588-1599,609)
__global__ void k_ref_init(const float* __restrict__ delta, int64_t ldd, const float* __restrict__ anchors,
                           const int32_t* __restrict__ idx, int S, float* __restrict__ ref_unact8, float* __restrict__ ref8,
                           int64_t total) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  int r = idx[t];
  r = min(max(r, 0), S - 1);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float u = delta[t * ldd + k] + anchors[(int64_t)r * 4 + k];
    ref_unact8[t * 8 + k] = u;
    ref8[t * 8 + k] = sigmoidf_(u);
    ref_unact8[t * 8 + 4 + k] = 0.f;
    ref8[t * 8 + 4 + k] = 0.f;
  }
}
void launch_ref_init(const Tensor& boxdelta, const float* anchors, const int32_t* idx, int S, float* ref_unact8, float* ref8,
                     hipStream_t s) {
  RTD_CHECK(boxdelta.dt == F32 && boxdelta.c == 4, 1, "ref_init: fp32 [.,4] deltas expected");
  const int64_t total = boxdelta.pixels();
  rtd_launch(k_ref_init, dim3(blocks_for(total, 256)), dim3(256), 0, s, (const float*)boxdelta.p, boxdelta.ld, anchors,
                     idx, S, ref_unact8, ref8, total);
  HIP_CHECK(hipGetLastError());
}
// new_reference = sigmoid(bbox_embed(hs) + inverse_sigmoid(reference))   (HF:v2.py:636-639)
__global__ void k_box_refine(const float* __restrict__ delta, int64_t ldd, float* __restrict__ ref8, int64_t total) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
#pragma unroll
  for (int k = 0; k < 4; ++k) ref8[t * 8 + k] = sigmoidf_(delta[t * ldd + k] + inv_sigmoid(ref8[t * 8 + k]));
}
void launch_box_refine(const Tensor& delta, float* ref8, hipStream_t s) {
  RTD_CHECK(delta.dt == F32 && delta.c == 4, 1, "box_refine: fp32 [.,4] deltas expected");
  const int64_t total = delta.pixels();
  rtd_launch(k_box_refine, dim3(blocks_for(total, 256)), dim3(256), 0, s, (const float*)delta.p, delta.ld, ref8, total);
  HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------ MS-deformable sampling
// HF:v2.py:44-115,186-221 (method "default"): for each (b, query, head): softmax over the
// n_levels*n_points attention logits, sampling location = ref_xy + off/n_points * ref_wh * offset_scale,
// grid_sample(bilinear, padding zeros, align_corners=False) of that head's 32-channel value rows.
// 32 lanes = the 32 channels of one (b, q, head): each bilinear tap is one coalesced 64/128-byte row read.
template <typename TV, typename TO>
__global__ __launch_bounds__(256) void k_msdeform(const TV* __restrict__ value, int64_t ldv, int64_t v_bstride,
                                                   const float* __restrict__ offaw, int64_t ldoa,
                                                   const float* __restrict__ ref8, TO* __restrict__ out, int64_t ldo,
                                                   int Q, int heads, int n_levels, int n_points,
                                                   const int32_t* __restrict__ lvl, float offset_scale, int64_t items) {
  const int64_t item = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);   // (b, q, head)
  const int ch = threadIdx.x & 31;
  if (item >= items) return;
  const int head = (int)(item % heads);
  const int64_t bq = item / heads;
  const int b = (int)(bq / Q);
  const int LP = n_levels * n_points;
  const float* oa = offaw + bq * ldoa;
  const float* offs = oa + (int64_t)head * LP * 2;
  const float* awl = oa + (int64_t)heads * LP * 2 + (int64_t)head * LP;
  float mx = -INFINITY;
  for (int i = 0; i < LP; ++i) mx = fmaxf(mx, awl[i]);
  float den = 0.f;
  for (int i = 0; i < LP; ++i) den += __expf(awl[i] - mx);
  const float inv_den = 1.f / den;
  const float rx = ref8[bq * 8 + 0], ry = ref8[bq * 8 + 1], rw = ref8[bq * 8 + 2], rh = ref8[bq * 8 + 3];
  const float pscale = 1.f / (float)n_points;
  const TV* vb = value + (int64_t)b * v_bstride + head * 32 + ch;
  float acc = 0.f;
  for (int l = 0; l < n_levels; ++l) {
    const int H = lvl[l * 3 + 0], W = lvl[l * 3 + 1], start = lvl[l * 3 + 2];
    for (int p = 0; p < n_points; ++p) {
      const int i = l * n_points + p;
      const float aw = __expf(awl[i] - mx) * inv_den;
      const float lx = rx + offs[i * 2 + 0] * pscale * rw * offset_scale;
      const float ly = ry + offs[i * 2 + 1] * pscale * rh * offset_scale;
      const float gx = 2.f * lx - 1.f, gy = 2.f * ly - 1.f;                 // sampling_grids = 2*loc - 1
      const float ix = ((gx + 1.f) * (float)W - 1.f) * 0.5f;                // align_corners=False unnormalise
      const float iy = ((gy + 1.f) * (float)H - 1.f) * 0.5f;
      const float fx = floorf(ix), fy = floorf(iy);
      const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
      const float wx1 = ix - fx, wy1 = iy - fy, wx0 = (fx + 1.f) - ix, wy0 = (fy + 1.f) - iy;
      float sv = 0.f;
      const bool okx0 = (unsigned)x0 < (unsigned)W, okx1 = (unsigned)x1 < (unsigned)W;
      const bool oky0 = (unsigned)y0 < (unsigned)H, oky1 = (unsigned)y1 < (unsigned)H;
      if (oky0 && okx0) sv += (float)vb[(int64_t)(start + y0 * W + x0) * ldv] * (wx0 * wy0);
      if (oky0 && okx1) sv += (float)vb[(int64_t)(start + y0 * W + x1) * ldv] * (wx1 * wy0);
      if (oky1 && okx0) sv += (float)vb[(int64_t)(start + y1 * W + x0) * ldv] * (wx0 * wy1);
      if (oky1 && okx1) sv += (float)vb[(int64_t)(start + y1 * W + x1) * ldv] * (wx1 * wy1);
      acc += sv * aw;
    }
  }
  out[bq * ldo + head * 32 + ch] = (TO)acc;
}
void launch_msdeform(const Tensor& value, int value_coff, const Tensor& offaw, const float* ref8, const Tensor& out, int heads,
                     int hd, int n_levels, int n_points, const int32_t* level_hw_start, float offset_scale, hipStream_t s) {
  RTD_CHECK(hd == 32, 1, "msdeform: head dim must be 32");
  RTD_CHECK(offaw.dt == F32 && offaw.c == heads * n_levels * n_points * 3, 1, "msdeform: offsets|weights layout");
  RTD_CHECK(out.c == heads * hd && out.pixels() == offaw.pixels(), 1, "msdeform: output shape");
  const int B = out.n, Q = out.h * out.w;
  const int64_t items = (int64_t)B * Q * heads;
  const dim3 grid(blocks_for(items, 8)), blk(256);
  const char* vp = (const char*)value.p + (size_t)value_coff * dtype_size(value.dt);
#define MS_GO(TV, TO) rtd_launch((k_msdeform<TV, TO>), grid, blk, 0, s, (const TV*)vp, value.ld, value.bstride, (const float*)offaw.p, offaw.ld, ref8, (TO*)out.p, out.ld, Q, heads, n_levels, n_points, level_hw_start, offset_scale, items)
  if (value.dt == BF16 && out.dt == BF16) MS_GO(bf16, bf16);
  else if (value.dt == BF16) MS_GO(bf16, float);
  else if (out.dt == BF16) MS_GO(float, bf16);
  else MS_GO(float, float);
#undef MS_GO
  HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------ post-process
// RTDETRPostProcessor (deploy) == HF:rt_detr/image_processing_rt_detr.py:510-533
__global__ void k_pp_scores(const float* __restrict__ logits, int64_t ld, int C, int64_t rows, float* __restrict__ scores) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * C) return;
  const int64_t r = i / C;
  const int c = (int)(i - r * C);
  scores[i] = sigmoidf_(logits[r * ld + c]);
}
void launch_postprocess_scores(const Tensor& logits, float* scores, hipStream_t s) {
  RTD_CHECK(logits.dt == F32, 1, "postprocess: fp32 logits expected");
  const int64_t rows = logits.pixels();
  rtd_launch(k_pp_scores, dim3(blocks_for(rows * logits.c, 256)), dim3(256), 0, s, (const float*)logits.p, logits.ld,
                     logits.c, rows, scores);
  HIP_CHECK(hipGetLastError());
}
// labels = index % C ; query = index // C ; boxes = cxcywh->xyxy * (w,h,w,h) of the ORIGINAL frame
__global__ void k_pp_gather(const float* __restrict__ topv, const int32_t* __restrict__ topi, const float* __restrict__ ref8,
                            const float* __restrict__ scale_wh, int Q, int C, float* __restrict__ block6, int64_t total) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int b = (int)(t / Q);
  const int idx = topi[t];
  const int label = idx % C;
  int q = idx / C;
  q = min(max(q, 0), Q - 1);
  const float* r = ref8 + ((int64_t)b * Q + q) * 8;
  const float cx = r[0], cy = r[1], w = r[2], h = r[3];
  const float sw = scale_wh[b * 2 + 0], sh = scale_wh[b * 2 + 1];
  float* o = block6 + t * 6;
  o[0] = (float)label;
  o[1] = topv[t];
  o[2] = (cx - 0.5f * w) * sw;
  o[3] = (cy - 0.5f * h) * sh;
  o[4] = (cx + 0.5f * w) * sw;
  o[5] = (cy + 0.5f * h) * sh;
}
void launch_postprocess_gather(const float* topv, const int32_t* topi, const float* ref8, const float* scale_wh, int B, int Q,
                               int C, float* block6, hipStream_t s) {
  const int64_t total = (int64_t)B * Q;
  rtd_launch(k_pp_gather, dim3(blocks_for(total, 256)), dim3(256), 0, s, topv, topi, ref8, scale_wh, Q, C, block6, total);
  HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------ pre-process
// src/rtdetr_detector.py:224-231 for a frame that already has the network's input size: BGR->RGB,
// ToTensor (uint8 / 255), written as NHWC with the 3 channels padded to 8 (one 16-byte bf16 chunk).
template <typename T>
__global__ void k_preprocess_identity(const FrameArgs fa, int H, int W, T* __restrict__ y, float* __restrict__ scale_wh, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 2 * fa.n) scale_wh[i] = fa.scale_wh[i];               // the post-processor's orig_target_sizes
  if (i >= total) return;
  const int64_t hw = (int64_t)H * W;
  const int b = (int)(i / hw);
  const int64_t p = i - (int64_t)b * hw;
  const uint8_t* f = fa.ptr[b] + p * 3;
  const float bl = (float)f[0], g = (float)f[1], r = (float)f[2];
  T* o = y + i * 8;
  o[0] = (T)(r / 255.0f);
  o[1] = (T)(g / 255.0f);
  o[2] = (T)(bl / 255.0f);
#pragma unroll
  for (int k = 3; k < 8; ++k) o[k] = (T)0.f;
}
void launch_preprocess_identity(const FrameArgs& fa, int H, int W, const Tensor& y, float* scale_wh_dev, hipStream_t s) {
  RTD_CHECK(y.c == 8 && y.ld == 8 && y.h == H && y.w == W && y.n >= fa.n, 1, "preprocess: output must be [n,H,W,8]");
  const int64_t total = (int64_t)fa.n * H * W;
  DISPATCH_T(y.dt, rtd_launch(k_preprocess_identity<T>, dim3(blocks_for(total, 256)), dim3(256), 0, s, fa, H, W, (T*)y.p, scale_wh_dev, total));
  HIP_CHECK(hipGetLastError());
}
__global__ void k_set_scale(const FrameArgs fa, float* __restrict__ scale_wh) {
  const int i = threadIdx.x;
  if (i < 2 * fa.n) scale_wh[i] = fa.scale_wh[i];
}
void launch_set_scale(const FrameArgs& fa, float* scale_wh_dev, hipStream_t s) {
  rtd_launch(k_set_scale, dim3(1), dim3(2 * RTD_MAX_BATCH), 0, s, fa, scale_wh_dev);
  HIP_CHECK(hipGetLastError());
}

// PIL's antialiased bilinear stretch-resize (ImagingResample, 8 bits per channel): two separable
// passes with fixed-point coefficients (22 fractional bits) and a uint8 intermediate - the exact
// arithmetic of `T.Resize` on a PIL image (src/rtdetr_detector.py:176-180).  The coefficient
// tables are computed on the host (engine.hip: pil_coeffs) the way Pillow's precompute_coeffs does.
__device__ __forceinline__ int clip8(int v) {
  v >>= 22;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}
__global__ void k_resize_h(const uint8_t* __restrict__ src, int sh, int sw, uint8_t* __restrict__ tmp, int dw, ResizeCoef c) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)sh * dw) return;
  const int y = (int)(i / dw), x = (int)(i - (int64_t)y * dw);
  const int xmin = c.hb[x * 2], cnt = c.hb[x * 2 + 1];
  const int32_t* k = c.hk + (int64_t)x * c.hks;
  int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
  const uint8_t* row = src + ((int64_t)y * sw + xmin) * 3;
  for (int j = 0; j < cnt; ++j) {
    s0 += (int)row[j * 3 + 0] * k[j];
    s1 += (int)row[j * 3 + 1] * k[j];
    s2 += (int)row[j * 3 + 2] * k[j];
  }
  uint8_t* o = tmp + i * 3;
  o[0] = (uint8_t)clip8(s0); o[1] = (uint8_t)clip8(s1); o[2] = (uint8_t)clip8(s2);
}
template <typename T>
__global__ void k_resize_v(const uint8_t* __restrict__ tmp, int sh, int dw, T* __restrict__ y, int dh, ResizeCoef c) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)dh * dw) return;
  const int yy = (int)(i / dw), x = (int)(i - (int64_t)yy * dw);
  const int ymin = c.vb[yy * 2], cnt = c.vb[yy * 2 + 1];
  const int32_t* k = c.vk + (int64_t)yy * c.vks;
  int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
  for (int j = 0; j < cnt; ++j) {
    const uint8_t* p = tmp + ((int64_t)(ymin + j) * dw + x) * 3;
    s0 += (int)p[0] * k[j];
    s1 += (int)p[1] * k[j];
    s2 += (int)p[2] * k[j];
  }
  T* o = y + i * 8;
  o[0] = (T)((float)clip8(s2) / 255.0f);   // BGR -> RGB
  o[1] = (T)((float)clip8(s1) / 255.0f);
  o[2] = (T)((float)clip8(s0) / 255.0f);
#pragma unroll
  for (int q = 3; q < 8; ++q) o[q] = (T)0.f;
}
void launch_resize_pil(const uint8_t* src, int sh, int sw, uint8_t* tmp, const Tensor& y, int image, const ResizeCoef& c,
                       hipStream_t s) {
  const int dh = y.h, dw = y.w;
  RTD_CHECK(y.c == 8 && y.ld == 8 && image < y.n, 1, "resize: output must be [n,H,W,8]");
  rtd_launch(k_resize_h, dim3(blocks_for((int64_t)sh * dw, 256)), dim3(256), 0, s, src, sh, sw, tmp, dw, c);
  char* yp = (char*)y.p + (size_t)image * y.bstride * dtype_size(y.dt);
  DISPATCH_T(y.dt, rtd_launch(k_resize_v<T>, dim3(blocks_for((int64_t)dh * dw, 256)), dim3(256), 0, s, tmp, sh, dw, (T*)yp, dh, c));
  HIP_CHECK(hipGetLastError());
}


__global__ void k_set_frame_table(const FrameArgs fa, const uint8_t** __restrict__ table, float* __restrict__ scale_wh) {
  const int i = threadIdx.x;
  if (i < 2 * fa.n) scale_wh[i] = fa.scale_wh[i];
  if (i < fa.n) table[i] = fa.ptr[i];
}
void launch_set_frame_table(const FrameArgs& fa, const uint8_t** table_dev, float* scale_wh_dev, hipStream_t s) {
  rtd_launch(k_set_frame_table, dim3(1), dim3(2 * RTD_MAX_BATCH), 0, s, fa, table_dev, scale_wh_dev);
  HIP_CHECK(hipGetLastError());
}

// vertical pass of the PIL resampler with a uint8 HWC result in the SOURCE channel order (the fused stem does BGR->RGB itself)
__global__ void k_resize_v_u8(const uint8_t* __restrict__ tmp, int sh, int dw, uint8_t* __restrict__ y, int dh, ResizeCoef c) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)dh * dw) return;
  const int yy = (int)(i / dw), x = (int)(i - (int64_t)yy * dw);
  const int ymin = c.vb[yy * 2], cnt = c.vb[yy * 2 + 1];
  const int32_t* k = c.vk + (int64_t)yy * c.vks;
  int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
  for (int j = 0; j < cnt; ++j) {
    const uint8_t* p = tmp + ((int64_t)(ymin + j) * dw + x) * 3;
    s0 += (int)p[0] * k[j];
    s1 += (int)p[1] * k[j];
    s2 += (int)p[2] * k[j];
  }
  uint8_t* o = y + i * 3;
  o[0] = (uint8_t)clip8(s0); o[1] = (uint8_t)clip8(s1); o[2] = (uint8_t)clip8(s2);
}
void launch_resize_pil_u8(const uint8_t* src, int sh, int sw, uint8_t* tmp, uint8_t* dst, int dh, int dw, const ResizeCoef& c, hipStream_t s) {
  rtd_launch(k_resize_h, dim3(blocks_for((int64_t)sh * dw, 256)), dim3(256), 0, s, src, sh, sw, tmp, dw, c);
  rtd_launch(k_resize_v_u8, dim3(blocks_for((int64_t)dh * dw, 256)), dim3(256), 0, s, tmp, sh, dw, dst, dh, c);
  HIP_CHECK(hipGetLastError());
}

// RTDETRDetector.preprocess (src/rtdetr_detector.py:206-236) as a value: uint8 HWC BGR at the network's size -> [3][H][W] fp32 RGB in [0, 1]
// (ToTensor: v / 255.0f, the division the engines' own input paths perform).  One thread per pixel.
__global__ void k_u8_hwc_to_chw_f32(const uint8_t* __restrict__ src, int H, int W, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t hw = (int64_t)H * W;
  if (i >= hw) return;
  const uint8_t* px = src + i * 3;
  out[i] = (float)px[2] / 255.0f;
  out[hw + i] = (float)px[1] / 255.0f;
  out[2 * hw + i] = (float)px[0] / 255.0f;
}
void launch_u8_hwc_to_chw_f32(const uint8_t* src, int H, int W, float* out, hipStream_t s) {
  rtd_launch(k_u8_hwc_to_chw_f32, dim3(blocks_for((int64_t)H * W, 256)), dim3(256), 0, s, src, H, W, out);
  HIP_CHECK(hipGetLastError());
}

// ---- backbone.stem.0 from uint8 frames -------------------------------------------------------------------------------------
// The generic path writes every frame as bf16 NHWC with 3 -> 8 channel padding (16 bytes per pixel: 52 MB at 640^2 bs 8) and the
// stem conv reads it back with K = 72 of which 27 taps are real.  Here a block stages the (2*8+1) x (2*32+1) pixel uint8 patch of
// its 8 x 32 output tile (3.3 KB), every lane gathers the 27 real taps of its output pixel as bytes, converts them exactly like
// the preprocess kernel ((T)(v / 255.0f)) and feeds two v_mfma_f32_32x32x16_bf16 steps (K = 32).  HBM: 3 bytes per input pixel.
typedef float f32x16_s __attribute__((ext_vector_type(16)));
typedef float f32x4_s __attribute__((ext_vector_type(4)));
// SPLIT (f16x3 engine): the normalised pixel v / 255.0f is kept as a hi / lo fp16 pair (two patches), the filter row `wq` is a pair row
// (K = 72 real taps x channels in 32-element groups [32 hi | 32 lo]), every MFMA step runs hi*hi + hi*lo + lo*hi and the output rows are
// F16X2 pixels ([32 hi | 32 lo], y_bstride / ldy in channels).
__global__ __launch_bounds__(256) void stem0_u8_kernel(const uint8_t* const* __restrict__ table, int H, int W, const sp16* __restrict__ wq, int Kpad,
                                                        const float* __restrict__ bias, sp16* __restrict__ y, long long y_bstride, long long ldy,
                                                        int OH, int OW, int tiles_x, int tiles_y, int act) {
  constexpr bool SPLIT = true;    // (the kernel began as a template over the bf16 engine's plain form, measured neutral there and removed)
  typedef sp16 E;
  typedef E Ex8 __attribute__((ext_vector_type(8)));
  typedef E Ex4 __attribute__((ext_vector_type(4)));
  constexpr int TH = 8, TW = 32, PR = 2 * TH + 1, PC = 2 * TW + 1, ROWE = 200;        // 65 px * 3 = 195 elements per patch row -> 200
  constexpr int NDW = (PC * 3 + 3 + 3) / 4;                                            // aligned dwords that cover one patch row
  constexpr int ROWO = (SPLIT ? 128 : 64) + 16;                                        // store slab row: 32 channels + skew
  __shared__ __attribute__((aligned(16))) E patch[PR * ROWE];                        // already normalised: (E)(v / 255.0f)
  __shared__ __attribute__((aligned(16))) E patch_lo[SPLIT ? PR * ROWE : 8];         // SPLIT: E(v / 255.0f - hi)
  __shared__ __attribute__((aligned(16))) char stage[4][32 * ROWO];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, h = lane >> 5;
  int t = blockIdx.x;
  const int tx = t % tiles_x; t /= tiles_x;
  const int ty = t % tiles_y;
  const int b = t / tiles_y;
  const int x0 = tx * TW, y0 = ty * TH;
  const uint8_t* __restrict__ f = table[b];
  const long long fbytes = (long long)H * W * 3;
  const bool aligned4 = (((uintptr_t)f) & 3) == 0;
  // patch: input rows 2*y0-1 .. 2*y0+15, cols 2*x0-1 .. 2*x0+63; zero outside the frame (the conv's zero padding of the
  // normalised image).  One aligned dword of the frame per thread and step, each byte converted ONCE.
  for (int e = tid; e < PR * NDW; e += 256) {
    const int r = e / NDW, d = e - r * NDW;
    const int iy = 2 * y0 - 1 + r;
    const long long g0 = ((long long)iy * W + (2 * x0 - 1)) * 3;                       // frame byte of patch element (r, 0); may be < 0
    const long long a0 = (g0 & ~3ll) + 4ll * d;                                         // this thread's aligned dword
    unsigned dw = 0;
    const bool row_ok = (unsigned)iy < (unsigned)H;
    if (row_ok) {
      if (aligned4 && a0 >= 0 && a0 + 4 <= fbytes) dw = *(const unsigned*)(f + a0);
      else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (a0 + q >= 0 && a0 + q < fbytes) dw |= (unsigned)f[a0 + q] << (8 * q);
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int el = (int)(a0 + q - g0);                                                // element index inside the patch row
      if (el < 0 || el >= PC * 3) continue;
      const int ix = 2 * x0 - 1 + el / 3;
      const float v = (row_ok && (unsigned)ix < (unsigned)W) ? (float)((dw >> (8 * q)) & 0xffu) : 0.f;
      const float vn = v / 255.0f;
      const E hi = (E)vn;
      patch[r * ROWE + el] = hi;
      if (SPLIT) patch_lo[r * ROWE + el] = (E)(vn - (float)hi);
    }
  }
  for (int e = tid; e < PR * (ROWE - PC * 3); e += 256) {                               // row tails: read (times a zero filter tap) by the last pixels
    const int r = e / (ROWE - PC * 3), i = e - r * (ROWE - PC * 3);
    patch[r * ROWE + PC * 3 + i] = (E)0.f;
    if (SPLIT) patch_lo[r * ROWE + PC * 3 + i] = (E)0.f;
  }
  // K layout chosen for the GATHER, not for the filter: k = 10 kh + e, e = 3 kw + (BGR byte) for e < 9, e = 9 and k = 30, 31 carry
  // zero filter taps.  A filter row of the patch is then 10 consecutive elements starting at an even offset, so the 8 k values of
  // a lane (k = 16 s + 8 (lane >> 5) + j) are at most two runs of whole dwords: 4 ds_read_b32 per MFMA operand instead of 8
  // ds_read_u16 (measured: no change, 44.8 us either way - the launch is bound by the per-block patch staging, not the gather).
  // Pixel p starts at element 6 p: consecutive lanes are 3 banks apart.
  Ex8 wf[2], wfl[2];
  int doff[2][4];                     // dword j of the operand: element offset relative to the pixel's (2 r, 2 p) corner
#pragma unroll
  for (int s = 0; s < 2; ++s) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = 16 * s + 8 * h + j;
      const int kh = k / 10, e = k - kh * 10;
      const int kw = e / 3, cb = e - kw * 3;                                            // frame byte cb (BGR) = model channel 2 - cb (RGB)
      const int kr = (kh * 3 + kw) * 8 + (2 - cb);                                      // the filter's own K index (tap-major, 8 padded channels)
      const bool real = k < 30 && e < 9;
      if (SPLIT) {
        const size_t pos = (size_t)(lane & 31) * Kpad + ((kr >> 5) << 6) + (kr & 31);   // F16X2 row: group kr / 32, hi half
        wf[s][j] = real ? wq[pos] : (E)0.f;
        wfl[s][j] = real ? wq[pos + 32] : (E)0.f;
      } else {
        wf[s][j] = real ? wq[(size_t)(lane & 31) * Kpad + kr] : (E)0.f;              // zero taps: the pixel operand may be anything finite
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = 16 * s + 8 * h + 2 * j;
      const int kh = k / 10, e = k - kh * 10;
      doff[s][j] = k < 30 ? kh * ROWE + e : 0;                                          // e even: a whole dword of the row
    }
  }
  __syncthreads();
#pragma unroll 1
  for (int rr = 0; rr < 2; ++rr) {
    const int r = wv * 2 + rr;
    const int eoff = (2 * r) * ROWE + (2 * (lane & 31)) * 3;
    const E* base = patch + eoff;
    f32x16_s acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      typedef unsigned u4_s __attribute__((ext_vector_type(4)));
      unsigned xd[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) xd[j] = *(const unsigned*)(base + doff[s][j]);
      const u4_s xv = {xd[0], xd[1], xd[2], xd[3]};
      if constexpr (SPLIT) acc = mfma_pair32(wf[s], __builtin_bit_cast(Ex8, xv), acc);
      else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[s], __builtin_bit_cast(Ex8, xv), acc, 0, 0, 0);
      if constexpr (SPLIT) {
        unsigned xl[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) xl[j] = *(const unsigned*)(patch_lo + eoff + doff[s][j]);
        const u4_s xlv = {xl[0], xl[1], xl[2], xl[3]};
        acc = mfma_pair32(wf[s], __builtin_bit_cast(Ex8, xlv), acc);
        acc = mfma_pair32(wfl[s], __builtin_bit_cast(Ex8, xv), acc);
      }
    }
    const int oy = y0 + r;
    char* sw_ = stage[wv];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4_s bv = *(const f32x4_s*)(bias + 8 * q + 4 * h);
      Ex4 o, ol;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = acc[4 * q + e] + bv[e];
        if (act == ACT_RELU) v = fmaxf(v, 0.f);
        else if (act == ACT_SILU) v = v / (1.f + __expf(-v));
        if constexpr (SPLIT) { sp16 hh, ll; split2(v, hh, ll); o[e] = hh; ol[e] = ll; }
        else o[e] = (E)v;
      }
      *(Ex4*)(sw_ + (lane & 31) * ROWO + (8 * q + 4 * h) * 2) = o;
      if (SPLIT) *(Ex4*)(sw_ + (lane & 31) * ROWO + 64 + (8 * q + 4 * h) * 2) = ol;
    }
    __builtin_amdgcn_wave_barrier();
    if (oy < OH) {
      constexpr int CPP = SPLIT ? 8 : 4;                                               // 16-byte chunks per output pixel
      E* yrow = y + (SPLIT ? 2 : 1) * ((long long)b * y_bstride + ((long long)oy * OW + x0) * ldy);
#pragma unroll
      for (int i2 = 0; i2 < CPP / 2; ++i2) {
        const int idx = i2 * 64 + lane;
        const int p = idx / CPP, ch = idx % CPP;
        if (x0 + p < OW) *(Ex8*)(yrow + (SPLIT ? 2 : 1) * (long long)p * ldy + ch * 8) = *(const Ex8*)(sw_ + p * ROWO + ch * 16);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}
void launch_stem0_u8(const uint8_t* const* table_dev, int n, int H, int W, const void* w, int Kpad, const float* bias, const Tensor& y, int act,
                     hipStream_t s) {
  const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
  RTD_CHECK(y.dt == F16X2 && y.c == 32 && y.h == OH && y.w == OW && y.n >= n && y.ld % SPLIT_GROUP == 0 &&
                (act == ACT_RELU || act == ACT_NONE || act == ACT_SILU), 1, "stem0_u8: output must be F16X2 [n, H/2, W/2, 32]");
  const int tiles_x = (OW + 31) / 32, tiles_y = (OH + 7) / 8;
  rtd_launch(stem0_u8_kernel, dim3((unsigned)(n * tiles_x * tiles_y)), dim3(256), 0, s, table_dev, H, W, (const sp16*)w, Kpad, bias, (sp16*)y.p,
                     (long long)y.bstride, (long long)y.ld, OH, OW, tiles_x, tiles_y, act);
  HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------ Stage-2 crop batcher
// SpeciesClassifier.preprocess (/root/reference/src/species_classifier.py:298-352) for a whole batch of crops in one
// launch: slice frame[y1:y2, x1:x2] (src/two_stage_pipeline_yolox.py:289), BGR->RGB, bilinear resize to S x S with
// F.interpolate(align_corners=False) semantics (no antialias), /255, (x - mean) / std, NCHW fp32.
struct CropNorm { float mean[3], inv_std[3]; };
__global__ void k_crop_resize(const CropBatch cb, int n, int S, CropNorm nm, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t per = (int64_t)S * S;
  if (i >= per * n) return;
  const int c = (int)(i / per);
  const int p = (int)(i - (int64_t)c * per);
  const int oy = p / S, ox = p - oy * S;
  const int cw = cb.x2[c] - cb.x1[c], chh = cb.y2[c] - cb.y1[c];
  // area_pixel_compute_source_index(scale, dst, align_corners=False, cubic=False): max(scale * (dst + 0.5) - 0.5, 0)
  const float sx = (float)cw / (float)S, sy = (float)chh / (float)S;
  const float fx = fmaxf(sx * ((float)ox + 0.5f) - 0.5f, 0.f), fy = fmaxf(sy * ((float)oy + 0.5f) - 0.5f, 0.f);
  const int x0 = min((int)fx, cw - 1), y0 = min((int)fy, chh - 1);
  const int x1 = min(x0 + 1, cw - 1), y1 = min(y0 + 1, chh - 1);
  const float lx1 = fx - (float)x0, ly1 = fy - (float)y0, lx0 = 1.f - lx1, ly0 = 1.f - ly1;
  const uint8_t* f = cb.frame[c];
  const int64_t fw = cb.fw[c];
  const uint8_t* p00 = f + ((int64_t)(cb.y1[c] + y0) * fw + cb.x1[c] + x0) * 3;
  const uint8_t* p01 = f + ((int64_t)(cb.y1[c] + y0) * fw + cb.x1[c] + x1) * 3;
  const uint8_t* p10 = f + ((int64_t)(cb.y1[c] + y1) * fw + cb.x1[c] + x0) * 3;
  const uint8_t* p11 = f + ((int64_t)(cb.y1[c] + y1) * fw + cb.x1[c] + x1) * 3;
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    const int sc = 2 - ch;                                       // RGB channel ch = BGR channel 2 - ch
    const float v = ly0 * (lx0 * (float)p00[sc] + lx1 * (float)p01[sc]) + ly1 * (lx0 * (float)p10[sc] + lx1 * (float)p11[sc]);
    out[((int64_t)c * 3 + ch) * per + p] = (v / 255.0f - nm.mean[ch]) * nm.inv_std[ch];
  }
}
void launch_crop_resize(const CropBatch& cb, int n, int out_size, const float mean[3], const float stdv[3], float* out, hipStream_t s) {
  RTD_CHECK(n >= 1 && n <= 64 && out_size >= 1 && out_size <= 4096, 1, "crop batch: 1..64 crops per launch");
  for (int i = 0; i < n; ++i) {
    RTD_CHECK(cb.frame[i] && cb.x1[i] >= 0 && cb.y1[i] >= 0 && cb.x2[i] > cb.x1[i] && cb.y2[i] > cb.y1[i] && cb.x2[i] <= cb.fw[i] &&
                  cb.y2[i] <= cb.fh[i], 1, "crop batch: rectangle outside its frame");
  }
  CropNorm nm;
  for (int k = 0; k < 3; ++k) { nm.mean[k] = mean[k]; nm.inv_std[k] = 1.0f / stdv[k]; }
  const int64_t total = (int64_t)n * out_size * out_size;
  rtd_launch(k_crop_resize, dim3(blocks_for(total, 256)), dim3(256), 0, s, cb, n, out_size, nm, out);
  HIP_CHECK(hipGetLastError());
}

}  // namespace rtd
