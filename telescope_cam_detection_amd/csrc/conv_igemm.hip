// conv_igemm.hip - implicit-GEMM convolution / linear layer for gfx950 (CDNA4), NHWC.
//
// One kernel family covers every conv and every token GEMM of RT-DETRv2
// (HF:rt_detr/modeling_rt_detr_resnet.py:37-68 ConvLayer, HF:rt_detr_v2/modeling_rt_detr_v2.py:817-835
// ConvNormLayer, every nn.Linear): BN is already folded into (filter, bias) at load time.
//
//   D[n][m] = sum_k  Wt[n][k] * X[m][k]          m = output pixel (b, oy, ox), n = output channel,
//                                                  k = (kh, kw, ci)  tap-major / channel-minor
//
// The filter is the MFMA "A" operand and the pixels the "B" operand, so an accumulator register
// group holds 4 CONSECUTIVE channels of ONE pixel: the epilogue (bias, residual, activation, store)
// is vectorised along NHWC's contiguous dimension.
//
//   bf16 : v_mfma_f32_32x32x16_bf16   (fp32 accumulate)
//   fp32 : v_mfma_f32_32x32x2_f32     (exact fp32 fma chain - the parity mode)
//
// Block = 256 threads = 4 waves (2 along pixels x 2 along channels), tile BM x BN x 32, operands
// staged global -> registers -> LDS (rows padded by one 16-byte access so ds_read_b128 is
// conflict-free), next tile's global loads issued before the current tile's MFMAs.
#include "common.h"

namespace rtd {

struct ConvK {
  const void* x;
  const void* w;
  const float* bias;
  const void* res;
  void* y;
  int M, H, W, Cin;
  long long ldx, x_bstride;
  int OH, OW, OHW;
  int N, Kreal, Kpad;
  int KH, KW, stride, pad;
  long long ldy, y_bstride, ldr, r_bstride;
  int act, res_mode, y_f32, res_f32;
  int ntn;
};

__device__ __forceinline__ float act_fn(float v, int act) {
  if (act == ACT_RELU) return fmaxf(v, 0.f);
  if (act == ACT_SILU) return v / (1.f + __expf(-v));
  if (act == ACT_GELU) return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
  return v;
}

template <typename T> struct Mma;
template <> struct Mma<bf16> {
  static constexpr int KSUB = 16;  // k per fragment read (one MFMA)
  typedef bf16x8 Frag;
  static __device__ __forceinline__ void run(const Frag& a, const Frag& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  static constexpr int KSUB = 8;   // lanes 0-31 hold k 0..3, lanes 32-63 hold k 4..7 -> 4 MFMAs of K=2
  typedef f32x4 Frag;
  static __device__ __forceinline__ void run(const Frag& a, const Frag& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b[3], c, 0, 0, 0);
  }
};

template <typename T, int BM, int BN, bool SMALLC>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvK a) {
  constexpr int BK = 32;
  constexpr int EPC = 16 / (int)sizeof(T);   // elements per 16-byte chunk
  constexpr int CPR = BK / EPC;              // chunks per tile row
  constexpr int LDS_LD = BK + EPC;           // padded LDS row (elements)
  constexpr int A_CH = BM * CPR / 256;
  constexpr int B_CH = BN * CPR / 256;
  constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
  constexpr int KSUB = Mma<T>::KSUB;
  typedef typename Mma<T>::Frag Frag;
  static_assert(A_CH >= 1 && B_CH >= 1 && TM >= 1 && TN >= 1, "tile too small");

  __shared__ __attribute__((aligned(16))) T As[BM * LDS_LD];
  __shared__ __attribute__((aligned(16))) T Bs[BN * LDS_LD];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  const int wm = wv & 1, wn = wv >> 1;
  const int bid = blockIdx.x;
  const int nt = bid % a.ntn, mt = bid / a.ntn;
  const int m0 = mt * BM, n0 = nt * BN;

  // ---- per-thread staging assignments (fixed over the K loop) ---------------------------------
  long long a_base[A_CH];
  int a_iy0[A_CH], a_ix0[A_CH];
#pragma unroll
  for (int i = 0; i < A_CH; ++i) {
    const int id = tid + i * 256;
    const int row = id / CPR;
    const int m = m0 + row;
    if (m < a.M) {
      const int b = m / a.OHW;
      const int r = m - b * a.OHW;
      const int oy = r / a.OW;
      const int ox = r - oy * a.OW;
      a_base[i] = (long long)b * a.x_bstride;
      a_iy0[i] = oy * a.stride - a.pad;
      a_ix0[i] = ox * a.stride - a.pad;
    } else {
      a_base[i] = 0;
      a_iy0[i] = -(1 << 28);   // every tap out of bounds -> zeros
      a_ix0[i] = -(1 << 28);
    }
  }

  const T* __restrict__ xg = (const T*)a.x;
  const T* __restrict__ wg = (const T*)a.w;

  uint4 areg[A_CH], breg[B_CH];
  auto load_tiles = [&](int k0) {
    if constexpr (!SMALLC) {
      const int tap = k0 / a.Cin;
      const int c0 = k0 - tap * a.Cin;
      const int kh = tap / a.KW;
      const int kw = tap - kh * a.KW;
#pragma unroll
      for (int i = 0; i < A_CH; ++i) {
        const int ch = (tid + i * 256) % CPR;
        const int iy = a_iy0[i] + kh, ix = a_ix0[i] + kw;
        uint4 v = make_uint4(0, 0, 0, 0);
        if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) {
          const long long off = a_base[i] + ((long long)iy * a.W + ix) * a.ldx + c0 + ch * EPC;
          v = *(const uint4*)(xg + off);
        }
        areg[i] = v;
      }
    } else {
#pragma unroll
      for (int i = 0; i < A_CH; ++i) {
        const int ch = (tid + i * 256) % CPR;
        const int k = k0 + ch * EPC;
        const int tap = k / a.Cin;
        const int c = k - tap * a.Cin;
        const int kh = tap / a.KW;
        const int kw = tap - kh * a.KW;
        const int iy = a_iy0[i] + kh, ix = a_ix0[i] + kw;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (k < a.Kreal && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) {
          const long long off = a_base[i] + ((long long)iy * a.W + ix) * a.ldx + c;
          v = *(const uint4*)(xg + off);
        }
        areg[i] = v;
      }
    }
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {
      const int id = tid + i * 256;
      const int row = id / CPR, ch = id % CPR;
      breg[i] = *(const uint4*)(wg + (long long)(n0 + row) * a.Kpad + k0 + ch * EPC);
    }
  };
  auto store_tiles = [&]() {
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
      const int id = tid + i * 256;
      *(uint4*)(&As[(id / CPR) * LDS_LD + (id % CPR) * EPC]) = areg[i];
    }
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {
      const int id = tid + i * 256;
      *(uint4*)(&Bs[(id / CPR) * LDS_LD + (id % CPR) * EPC]) = breg[i];
    }
  };

  f32x16 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = a.Kpad / BK;
  const int frow = lane & 31;
  const int fk = (lane >> 5) * (KSUB / 2);
  load_tiles(0);
  for (int ks = 0; ks < nk; ++ks) {
    __syncthreads();               // previous tile fully consumed
    store_tiles();
    __syncthreads();
    if (ks + 1 < nk) load_tiles((ks + 1) * BK);   // in flight during the MFMAs below
#pragma unroll
    for (int kk = 0; kk < BK / KSUB; ++kk) {
      Frag xf[TM], wf[TN];
#pragma unroll
      for (int j = 0; j < TM; ++j)
        xf[j] = *(const Frag*)(&As[(wm * WM + j * 32 + frow) * LDS_LD + kk * KSUB + fk]);
#pragma unroll
      for (int i = 0; i < TN; ++i)
        wf[i] = *(const Frag*)(&Bs[(wn * WN + i * 32 + frow) * LDS_LD + kk * KSUB + fk]);
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) Mma<T>::run(wf[i], xf[j], acc[i][j]);
    }
  }

  // ---- epilogue: bias (+ residual) + activation, 4 consecutive channels per store --------------
  const int h = lane >> 5;
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const int m = m0 + wm * WM + j * 32 + (lane & 31);
    if (m >= a.M) continue;
    const int b = m / a.OHW;
    const int p = m - b * a.OHW;
    const long long yoff = (long long)b * a.y_bstride + (long long)p * a.ldy;
    const long long roff = (long long)b * a.r_bstride + (long long)p * a.ldr;
#pragma unroll
    for (int i = 0; i < TN; ++i) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c = n0 + wn * WN + i * 32 + 8 * g + 4 * h;
        if (c >= a.N) continue;
        const f32x4 bv = *(const f32x4*)(a.bias + c);
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = acc[i][j][4 * g + q] + bv[q];
        float rv[4] = {0.f, 0.f, 0.f, 0.f};
        if (a.res_mode != RES_NONE) {
          if (a.res_f32) {
            const f32x4 t = *(const f32x4*)((const float*)a.res + roff + c);
#pragma unroll
            for (int q = 0; q < 4; ++q) rv[q] = t[q];
          } else {
            const bf16x4 t = *(const bf16x4*)((const bf16*)a.res + roff + c);
#pragma unroll
            for (int q = 0; q < 4; ++q) rv[q] = (float)t[q];
          }
        }
        if (a.res_mode == RES_PRE) {
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] += rv[q];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = act_fn(v[q], a.act);
        if (a.res_mode == RES_POST) {
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] += rv[q];
        }
        if (a.y_f32) {
          f32x4 o = {v[0], v[1], v[2], v[3]};
          *(f32x4*)((float*)a.y + yoff + c) = o;
        } else {
          bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
          *(bf16x4*)((bf16*)a.y + yoff + c) = o;
        }
      }
    }
  }
}

int conv_kpad(int K) { return (K + 31) / 32 * 32; }
int conv_npad(int N) { return (N + 127) / 128 * 128; }

template <typename T>
static void dispatch(const ConvK& k, bool smallc, hipStream_t s) {
  const long long b128 = (long long)((k.M + 127) / 128);
  const int n128 = (k.N + 127) / 128, n64 = (k.N + 63) / 64;
  int cfg;  // 0: 128x128, 1: 128x64, 2: 64x64
  if (b128 * n128 >= 512 && k.N > 64) cfg = 0;
  else if (b128 * n64 >= 384) cfg = 1;
  else cfg = 2;
  ConvK kk = k;
#define RTD_LAUNCH(BM, BN)                                                                        \
  do {                                                                                            \
    kk.ntn = (k.N + BN - 1) / BN;                                                                 \
    const long long blocks = (long long)((k.M + BM - 1) / BM) * kk.ntn;                           \
    if (smallc) hipLaunchKernelGGL((conv_igemm_kernel<T, BM, BN, true>), dim3((unsigned)blocks), dim3(256), 0, s, kk); \
    else hipLaunchKernelGGL((conv_igemm_kernel<T, BM, BN, false>), dim3((unsigned)blocks), dim3(256), 0, s, kk);       \
  } while (0)
  if (cfg == 0) RTD_LAUNCH(128, 128);
  else if (cfg == 1) RTD_LAUNCH(128, 64);
  else RTD_LAUNCH(64, 64);
#undef RTD_LAUNCH
}

void launch_conv(const ConvArgs& a, hipStream_t s) {
  const Tensor& x = a.x;
  const Tensor& y = a.y;
  RTD_CHECK(x.dt == BF16 || x.dt == F32, 1, "conv: input dtype");
  RTD_CHECK(y.dt == BF16 || y.dt == F32, 1, "conv: output dtype");
  const int epc = x.dt == BF16 ? 8 : 4;
  const int OH = (x.h + 2 * a.pad - a.KH) / a.stride + 1;
  const int OW = (x.w + 2 * a.pad - a.KW) / a.stride + 1;
  RTD_CHECK(OH == y.h && OW == y.w && x.n == y.n, 1, "conv: output shape mismatch");
  RTD_CHECK(x.c % epc == 0 && x.ld % epc == 0, 1, "conv: Cin / pixel stride must be a multiple of one 16-byte chunk");
  RTD_CHECK(((uintptr_t)x.p & 15) == 0 && ((uintptr_t)a.w & 15) == 0, 1, "conv: 16-byte alignment");
  RTD_CHECK(y.c % 4 == 0 && y.ld % 4 == 0, 1, "conv: Cout / output stride must be multiples of 4");
  RTD_CHECK(((uintptr_t)y.p & (y.dt == BF16 ? 7 : 15)) == 0, 1, "conv: output alignment");
  const int K = a.KH * a.KW * x.c;
  RTD_CHECK(a.Kpad == conv_kpad(K) && a.Npad >= y.c && a.Npad % 128 == 0, 1, "conv: filter padding");
  RTD_CHECK((long long)x.n * OH * OW < (1ll << 31), 1, "conv: M overflow");
  ConvK k;
  k.x = x.p; k.w = a.w; k.bias = a.bias; k.y = y.p;
  k.res = a.res_mode != RES_NONE ? a.res.p : nullptr;
  k.M = x.n * OH * OW; k.H = x.h; k.W = x.w; k.Cin = x.c;
  k.ldx = x.ld; k.x_bstride = x.bstride;
  k.OH = OH; k.OW = OW; k.OHW = OH * OW;
  k.N = y.c; k.Kreal = K; k.Kpad = a.Kpad;
  k.KH = a.KH; k.KW = a.KW; k.stride = a.stride; k.pad = a.pad;
  k.ldy = y.ld; k.y_bstride = y.bstride;
  k.ldr = 0; k.r_bstride = 0; k.res_f32 = 0;
  if (a.res_mode != RES_NONE) {
    RTD_CHECK(a.res.p && a.res.n == y.n && a.res.h == y.h && a.res.w == y.w && a.res.c == y.c, 1, "conv: residual shape");
    RTD_CHECK(a.res.ld % 4 == 0 && ((uintptr_t)a.res.p & (a.res.dt == BF16 ? 7 : 15)) == 0, 1, "conv: residual alignment");
    k.ldr = a.res.ld; k.r_bstride = a.res.bstride; k.res_f32 = a.res.dt == F32;
  }
  k.act = a.act; k.res_mode = a.res_mode; k.y_f32 = y.dt == F32;
  k.ntn = 1;
  const bool smallc = (x.c % 32) != 0;
  if (x.dt == BF16) dispatch<bf16>(k, smallc, s);
  else dispatch<float>(k, smallc, s);
  HIP_CHECK(hipGetLastError());
}

}  // namespace rtd
