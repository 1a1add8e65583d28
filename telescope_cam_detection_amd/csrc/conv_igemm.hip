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
#include <stdlib.h>
#include <string.h>

#include <algorithm>

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
  const void* pf;        // prefetch target (next layer's filter) or nullptr
  unsigned pf_bytes;
  int reg_epi;           // 1: the ws kernels may finish residual-free bf16 tiles in registers (A/B: rtd_debug_option "reg_epilogue")
  const void* x2;        // second input (ConvArgs::x2) or nullptr; K elements k2_start.. come from it
  long long ldx2, x2_bstride;
  int k2_start;
  int prefer256;         // ConvArgs::prefer256
  int x_up2;             // ConvArgs::x_up2 (ws kernel loader only)
  const void* next_w;    // ConvArgs::next_* (streaming kernel only)
  const float* next_bias;
  void* next_y;
  long long next_ldy, next_y_bstride;
  int next_kpad, next_act;
  // F16X2 operands (common.h): `split` = x / x2 / w are [32 hi | 32 lo] grouped bf16 and Cin, ldx, x_bstride, Kreal, Kpad, k2_start,
  // ldx2, x2_bstride count bf16 ELEMENTS (twice the channels); y_split / res_split = the output / residual is a F16X2 tensor, its
  // ldy / y_bstride (ldr / r_bstride) count channels as for fp32 and addresses go through split_off()
  int split = 0, y_split = 0, res_split = 0;
  int raw = 0;           // two-pass split-K: the tile kernels write bare fp32 partial sums (no bias) to slice `s` of ConvG::slab
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
typedef unsigned int u32x4_ __attribute__((ext_vector_type(4)));
// Touch `pf_bytes` at `pf` (one dword per 128-byte line, this block's share) with LDS-DMA into a 256-byte dummy: no VGPR is
// written, nothing waits for the data; the lines land in this XCD's L2 and in the Infinity Cache.
__device__ __forceinline__ void prefetch_share(const ConvK& a, unsigned block, unsigned nblocks, unsigned t, unsigned nthreads, char* dummy) {
  if (a.pf_bytes == 0) return;
  const unsigned lines = a.pf_bytes >> 7;
  const unsigned per = (lines + nblocks - 1) / nblocks;
  const unsigned l0 = block * per;
  const unsigned l1 = min(l0 + per, lines);
  if (l0 >= l1) return;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.pf, 0, a.pf_bytes, 0x00020000);
  for (unsigned l = l0 + t; l < l1; l += nthreads) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)dummy, 4, l << 7, 0, 0, 0);
}

// the activation is a kernel argument: the epilogue loops are instantiated once per activation (dispatch_act) so that the
// choice costs one scalar branch per block - as a per-element switch (8 scalar branches around each of a thread's 32-64 outputs)
// it was 4.5 us of a 20 us launch (s_memtime stamps, tools/conv_stamps.py)
template <int A> struct ActC { static constexpr int value = A; };
template <int ACT> __device__ __forceinline__ float act_c(float v) {
  if (ACT == ACT_RELU) return fmaxf(v, 0.f);
  if (ACT == ACT_SILU) return v / (1.f + __expf(-v));
  if (ACT == ACT_GELU) return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
  return v;
}
template <typename F> __device__ __forceinline__ void dispatch_act(int act, F&& f) {
  switch (act) {
    case ACT_RELU: f(ActC<ACT_RELU>{}); break;
    case ACT_SILU: f(ActC<ACT_SILU>{}); break;
    case ACT_GELU: f(ActC<ACT_GELU>{}); break;
    default: f(ActC<ACT_NONE>{}); break;
  }
}
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
  __shared__ __attribute__((aligned(16))) char pf_dummy[256];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  const int wm = wv & 1, wn = wv >> 1;
  const int bid = blockIdx.x;
  prefetch_share(a, bid, gridDim.x, tid, 256, pf_dummy);
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
        if (k0 < a.Kreal && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) {   // k0 >= Kreal: zero-padded filter tail
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
  dispatch_act(a.act, [&](auto actc) {
  constexpr int ACT = decltype(actc)::value;
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
        for (int q = 0; q < 4; ++q) v[q] = act_c<ACT>(v[q]);
        if (a.res_mode == RES_POST) {
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] += rv[q];
        }
        if (a.y_split) {
          sp16x4 oh, ol;
#pragma unroll
          for (int q = 0; q < 4; ++q) { sp16 hi, lo; split2(v[q], hi, lo); oh[q] = hi; ol[q] = lo; }
          sp16* yb = (sp16*)a.y + split_off(yoff, c);
          *(sp16x4*)yb = oh;
          *(sp16x4*)(yb + SPLIT_GROUP) = ol;
        } else if (a.y_f32) {
          f32x4 o = {v[0], v[1], v[2], v[3]};
          *(f32x4*)((float*)a.y + yoff + c) = o;
        } else {
          bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
          *(bf16x4*)((bf16*)a.y + yoff + c) = o;
        }
      }
    }
  }
  });
}


// ------------------------------------------------------------------------------------------------
// The LDS-DMA tile kernels (everything below the register-staged fallback above).  128-byte tile rows, both operands go
// global -> LDS with `buffer_load_dwordx4 ... lds`: no VGPR staging and no ds_write (the ds_write_b128 traffic of register-staged
// tiles costs more LDS-pipe cycles than the MFMAs).
//  * one wave-instruction writes 1 KiB = 8 tile rows x 128 B linearly, so rows are unpadded and the
//    bank-conflict swizzle lives on the SOURCE side: LDS slot s of row r holds the row's 16-byte
//    chunk s ^ ((r >> 1) & 7); readers apply the same involution (conflict-free for ds_read_b128's
//    16-lane groups, checked against the bank rule of MI355X_MICROARCH.md §LDS).
//  * the im2col gather is the per-lane source offset; zero padding = an out-of-range buffer offset
//    (the range check makes the DMA write zeros - probed with tools/glds_probe.hip).
//  * the epilogue is staged through LDS so that every global store / residual load is a coalesced 16-byte access along NHWC's
//    channel dimension; XCD-aware bijective block order (neighbouring pixel tiles share halo rows in one XCD's L2).
// ------------------------------------------------------------------------------------------------

// (image, row, column) of an output pixel, walked 32 pixels at a time: the loaders of the tile kernels need it for every 32nd row of their
// tile, and a division by a run-time extent is ~35 instructions with quarter-rate multiplies - seven rows' worth was 5000 cycles before a
// block's first DMA (stamps of conv_igemm_wsq_kernel: first barrier passed at 4944 cycles with the DMA switched off).
struct PixWalk { int b, oy, ox; };
__device__ __forceinline__ void pix_init(const ConvK& a, int m, PixWalk& p) {
  p.b = m / a.OHW;
  const int r = m - p.b * a.OHW;
  p.oy = r / a.OW;
  p.ox = r - p.oy * a.OW;
}
__device__ __forceinline__ void pix_step32(const ConvK& a, int dq, int dr, PixWalk& p) {   // dq = 32 / OW, dr = 32 % OW
  p.ox += dr; p.oy += dq;
  if (p.ox >= a.OW) { p.ox -= a.OW; ++p.oy; }
  while (p.oy >= a.OH) { p.oy -= a.OH; ++p.b; }
}

struct ConvG {
  ConvK k;
  unsigned x_bytes, w_bytes;   // extents of the two buffers from their base pointers (buffer descriptors)
  unsigned x2_bytes;           // extent of the second input
  unsigned y_bytes;            // extent of the output (A-stationary kernel: buffer stores), 0 = not provided
  int probe;                   // timing-only probes (results wrong): bit 2 = issue no DMA at all; bit 5 = block stamps into `slab`
  int splitk;                  // pair kernels, two-pass split-K: the grid is splitk x tiles (1 = off), see launch_conv_split
  float* slab;                 // diagnostics (probe bit 5): [blocks][8] stamps
};

__device__ __forceinline__ bool g_reg_epilogue_ok(const ConvK& a) { return a.res_mode == RES_NONE && !a.y_f32 && a.reg_epi; }

// Copy-out of a tile that is already final (bias + activation applied in the MFMA waves' registers, bf16 rows in LDS): pure
// 16-byte LDS -> global moves, 32 rows apart per iteration.
template <int ITERS, int BN = 128>   // 512 threads: BN / 8 chunks per row, 4096 / BN rows per pass
__device__ __forceinline__ void ws_copy_out_bf16(const ConvK& a, const bf16* sb, int SLB, int tid, int m0, int n0) {
  constexpr int CH8 = BN / 8, RSTEP = 512 / CH8;
  const int c8 = tid % CH8;
  const int c = n0 + c8 * 8;
  if (c >= a.N) return;
  int m = m0 + tid / CH8;
  const int b = m / a.OHW;
  int p = m - b * a.OHW;
  long long yoff = (long long)b * a.y_bstride + (long long)p * a.ldy + c;
  const bf16* srow = sb + (tid / CH8) * SLB + c8 * 8;
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    if (m < a.M) *(bf16x8*)((bf16*)a.y + yoff) = *(const bf16x8*)srow;
    m += RSTEP; p += RSTEP; yoff += RSTEP * a.ldy; srow += RSTEP * SLB;
    while (p >= a.OHW) { p -= a.OHW; yoff += a.y_bstride - (long long)a.OHW * a.ldy; }
  }
}

// Copy-out of the wave-specialised kernels: staging tile (fp32, [rows][SLD]) -> bias (+ residual) -> activation -> y, 8 channels
// (16 bytes of bf16) per thread and iteration, 32 rows apart.  The loop is instruction-bound (2 waves per SIMD walk it), so the
// pixel -> (image, offset) division is done once and carried, and the activation is a template parameter.
template <int ITERS, int ACT, int BN = 128>
__device__ __forceinline__ void ws_copy_out(const ConvK& a, const float* st, int SLD, int tid, int m0, int n0, const bf16x8* rpre) {
  constexpr int CH8 = BN / 8, RSTEP = 512 / CH8;
  const int c8 = tid % CH8;
  const int c = n0 + c8 * 8;
  if (c >= a.N) return;
  const f32x4 b0 = *(const f32x4*)(a.bias + c), b1 = *(const f32x4*)(a.bias + c + 4);
  int m = m0 + tid / CH8;
  int b = m / a.OHW;
  int p = m - b * a.OHW;
  long long yoff = (long long)b * a.y_bstride + (long long)p * a.ldy + c;
  long long roff = (long long)b * a.r_bstride + (long long)p * a.ldr + c;
  const float* srow = st + (tid / CH8) * SLD + c8 * 8;
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    if (m < a.M) {
      const f32x4 s0 = *(const f32x4*)(srow), s1 = *(const f32x4*)(srow + 4);
      float v[8] = {s0[0] + b0[0], s0[1] + b0[1], s0[2] + b0[2], s0[3] + b0[3],
                    s1[0] + b1[0], s1[1] + b1[1], s1[2] + b1[2], s1[3] + b1[3]};
      float rv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (a.res_mode != RES_NONE) {
        if (a.res_f32) {
          const f32x4 t0 = *(const f32x4*)((const float*)a.res + roff), t1 = *(const f32x4*)((const float*)a.res + roff + 4);
#pragma unroll
          for (int q = 0; q < 4; ++q) { rv[q] = t0[q]; rv[4 + q] = t1[q]; }
        } else {
#pragma unroll
          for (int q = 0; q < 8; ++q) rv[q] = (float)rpre[it][q];
        }
      }
      if (a.res_mode == RES_PRE) {
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] += rv[q];
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = act_c<ACT>(v[q]);
      if (a.res_mode == RES_POST) {
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] += rv[q];
      }
      if (a.y_f32) {
        f32x4 o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
        *(f32x4*)((float*)a.y + yoff) = o0;
        *(f32x4*)((float*)a.y + yoff + 4) = o1;
      } else {
        bf16x8 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3], (bf16)v[4], (bf16)v[5], (bf16)v[6], (bf16)v[7]};
        *(bf16x8*)((bf16*)a.y + yoff) = o;
      }
    }
    m += RSTEP;
    p += RSTEP;
    yoff += RSTEP * a.ldy;
    roff += RSTEP * a.ldr;
    srow += RSTEP * SLD;
    while (p >= a.OHW) {                                        // next image (maps smaller than 32 pixels wrap more than once)
      p -= a.OHW;
      yoff += a.y_bstride - (long long)a.OHW * a.ldy;
      roff += a.r_bstride - (long long)a.OHW * a.ldr;
    }
  }
}


// F16X2 output, tile already final in LDS as split rows ([pixel][2 BN sp16]: the tile's BN / 32 channel groups, each [32 hi | 32 lo],
// exactly the bytes of the pixel's slice in global memory): pure 16-byte moves, 4 BN bytes per pixel row.
template <int BN, int ROWS = 128>
__device__ __forceinline__ void ws_copy_out_split_rows(const ConvK& a, const sp16* sb, int SLB, int tid, int m0, int n0) {
  constexpr int CH = 2 * BN / 8, RSTEP = 512 / CH, ITERS = (ROWS + RSTEP - 1) / RSTEP;
  const int ch = tid % CH;
  if (n0 + (ch >> 3) * SPLIT_GROUP >= a.N) return;               // whole 32-channel groups (N % 32 == 0)
  int m = m0 + tid / CH;
  const int b = m / a.OHW;
  int p = m - b * a.OHW;
  long long yoff = 2 * ((long long)b * a.y_bstride + (long long)p * a.ldy + n0) + ch * 8;   // sp16 elements
  const sp16* srow = sb + (tid / CH) * SLB + ch * 8;
  int row = tid / CH;
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    if (m < a.M && (ROWS % RSTEP == 0 || row < ROWS)) *(sp16x8*)((sp16*)a.y + yoff) = *(const sp16x8*)srow;
    m += RSTEP; p += RSTEP; row += RSTEP; yoff += 2 * RSTEP * a.ldy; srow += RSTEP * SLB;
    while (p >= a.OHW) { p -= a.OHW; yoff += 2 * (a.y_bstride - (long long)a.OHW * a.ldy); }
  }
}

// Copy-out of the split kernels' fp32 staging tile: bias (+ residual: F16X2 or fp32) -> activation -> F16X2 or fp32 output, 8 channels
// per thread and iteration.  ldy / ldr count channels for either type (ConvK::y_split).
template <int ITERS, int ACT, int BN, int ROWS = 128>
__device__ __forceinline__ void ws_copy_out_sp(const ConvK& a, const float* st, int SLD, int tid, int m0, int n0, long long yoff = 0) {
  constexpr int CH8 = BN / 8, RSTEP = 512 / CH8;
  const int c8 = tid % CH8;
  const int c = n0 + c8 * 8;
  if (c >= a.N) return;
  f32x4 b0 = *(const f32x4*)(a.bias + c), b1 = *(const f32x4*)(a.bias + c + 4);
  if (a.raw) { b0 = f32x4{0.f, 0.f, 0.f, 0.f}; b1 = b0; }     // split-K partial sums: the reduce pass adds the bias
  int m = m0 + tid / CH8;
  const int b = m / a.OHW;
  int p = m - b * a.OHW;
  long long ypix = (long long)b * a.y_bstride + (long long)p * a.ldy;
  long long rpix = (long long)b * a.r_bstride + (long long)p * a.ldr;
  const float* srow = st + (tid / CH8) * SLD + c8 * 8;
  int row = tid / CH8;
#pragma unroll
  for (int it = 0; it < ITERS; ++it, row += RSTEP) {
    if (m < a.M && (ROWS % RSTEP == 0 || row < ROWS)) {
      const f32x4 s0 = *(const f32x4*)(srow), s1 = *(const f32x4*)(srow + 4);
      float v[8] = {s0[0] + b0[0], s0[1] + b0[1], s0[2] + b0[2], s0[3] + b0[3],
                    s1[0] + b1[0], s1[1] + b1[1], s1[2] + b1[2], s1[3] + b1[3]};
      float rv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (a.res_mode != RES_NONE) {
        if (a.res_split) {
          split_load8((const sp16*)a.res, rpix, c, rv);
        } else {
          const f32x4 t0 = *(const f32x4*)((const float*)a.res + rpix + c), t1 = *(const f32x4*)((const float*)a.res + rpix + c + 4);
#pragma unroll
          for (int q = 0; q < 4; ++q) { rv[q] = t0[q]; rv[4 + q] = t1[q]; }
        }
      }
      if (a.res_mode == RES_PRE) {
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] += rv[q];
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = act_c<ACT>(v[q]);
      if (a.res_mode == RES_POST) {
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] += rv[q];
      }
      if (a.y_split) {
        split_store8((sp16*)a.y, ypix, c, v);
      } else {
        f32x4 o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
        *(f32x4*)((float*)a.y + yoff + ypix + c) = o0;
        *(f32x4*)((float*)a.y + yoff + ypix + c + 4) = o1;
      }
    }
    m += RSTEP;
    p += RSTEP;
    ypix += RSTEP * a.ldy;
    rpix += RSTEP * a.ldr;
    srow += RSTEP * SLD;
    while (p >= a.OHW) {
      p -= a.OHW;
      ypix += a.y_bstride - (long long)a.OHW * a.ldy;
      rpix += a.r_bstride - (long long)a.OHW * a.ldr;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Wave-specialised LDS-DMA kernel (bf16 / fp32 operands).  8 waves per block with fixed roles: waves 0-3 only read fragments and
// issue MFMAs (one per SIMD, 64x64 outputs each), waves 4-7 only compute im2col offsets and issue the `buffer_load ... lds` DMA,
// STAGES-1 tiles ahead: the DMA issue cost (60-185 cycles per instruction next to MFMAs, MI355X_MICROARCH.md cycle constants) runs
// on the loader waves beside the MFMA waves.  One s_barrier per K-step orders both roles:
//   loader : wait tile ks (counted vmcnt) | barrier | issue tile ks+STAGES-1
//   compute:                                barrier | MFMA tile ks
// BN = 64: a 128 pixel x 64 channel tile for grids that leave most CUs idle with 128 x 128 tiles (stage 3 / PAN at batch 8, nearly
// everything at batch 1): twice the blocks, each MFMA wave owns 32 pixels x 64 channels (one pixel tile, two channel tiles), the
// loaders stage 24 KiB per K-step.  Same K order per output, so a layer's results do not depend on which tile width ran it.
// ------------------------------------------------------------------------------------------------
template <typename T, int STAGES, int BN = 128>
__global__ __launch_bounds__(512, (STAGES == 2 ? 4 : 2)) void conv_igemm_ws_kernel(const ConvG g) {
  const ConvK& a = g.k;
  constexpr int BM = 128;
  static_assert(BN == 128 || BN == 64, "tile widths");
  constexpr int TJ = BN == 128 ? 2 : 1;          // pixel tiles (32 rows) per MFMA wave; channel tiles per wave: always 2
  constexpr int NBI = BN / 32;                    // 32-row filter pieces per loader wave and K-step
  constexpr int PPT = 4 + NBI;                    // LDS-DMA pieces per loader wave and K-step (counted waits)
  constexpr int CH8 = BN / 8, RSTEP = 512 / CH8, CITERS = BM / RSTEP;   // copy-out: 16-byte chunks per row, rows per pass, passes
  constexpr int ES = (int)sizeof(T);
  constexpr int BK = 128 / ES;
  constexpr int STAGE = (BM + BN) * 128;
  constexpr int SLD = BN + 4;
  constexpr int SMEM = (STAGES * STAGE > BM * SLD * 4) ? STAGES * STAGE : BM * SLD * 4;
  constexpr int AHEAD = STAGES - 1;
  static_assert(STAGES >= 2 && STAGES <= 4, "counted waits below assume 1..3 tiles ahead");
  typedef typename Mma<T>::Frag Frag;
  __shared__ __attribute__((aligned(16))) char smem[SMEM + 256];   // + the prefetch dummy

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);      // provably wave-uniform -> scalar role branches
  const bool loader = wv >= 4;
  const int w4 = wv & 3;
  const int wm = BN == 128 ? (w4 & 1) : w4, wn = BN == 128 ? (w4 >> 1) : 0;
  int wg;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int nt = wg % a.ntn, mt = wg / a.ntn;
  const int m0 = mt * BM, n0 = nt * BN;
  const int nk = a.Kpad / BK;

  // diagnostic (glds_drop bit 5): block-level stamps [block][0..7] (shader clocks from kernel entry; [6],[7] = 100 MHz wall clock
  // at entry / exit): first tile landed, K loop done, staged, stores issued, stores complete
  long long* stamps = ((g.probe & 32) && g.slab && blockIdx.x < 4096 && lane == 0) ? (long long*)g.slab + (size_t)blockIdx.x * 8 : nullptr;
  const long long t_base = stamps ? (long long)__builtin_amdgcn_s_memtime() : 0;
  if (stamps && wv == 0) stamps[6] = (long long)__builtin_amdgcn_s_memrealtime();

  f32x16 acc[2][TJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // bf16 residual rows of this thread's 4 epilogue iterations: every load is issued at kernel start, beside the first tile's DMA (and
  // before any store to y, which the compiler must assume aliases res): the residual latency overlaps the K loop instead of
  // being paid four times in series in the epilogue
  const int c8 = tid % CH8;
  const int c = n0 + c8 * 8;
  bf16x8 rpre[4];
  if (a.res_mode != RES_NONE && !a.res_f32 && c < a.N) {
    int m = m0 + tid / CH8;
    const int b = m / a.OHW;
    int p = m - b * a.OHW;
    long long roff = (long long)b * a.r_bstride + (long long)p * a.ldr + c;
#pragma unroll
    for (int it = 0; it < CITERS; ++it) {                           // rows RSTEP apart: carry (image, pixel) instead of dividing again
      if (m < a.M) rpre[it] = *(const bf16x8*)((const bf16*)a.res + roff);
      m += RSTEP; p += RSTEP; roff += RSTEP * a.ldr;
      while (p >= a.OHW) { p -= a.OHW; roff += a.r_bstride - (long long)a.OHW * a.ldr; }
    }
  }
  if (!loader) prefetch_share(a, blockIdx.x, gridDim.x, tid, 256, smem + SMEM);   // next layer's filter; the MFMA waves idle until tile 0 lands

  if (loader) {
    // ---- loader role -------------------------------------------------------------------------------
    const int lrow = w4 * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((w4 * 4 + (lane >> 4)) & 7);
    int a_off[4], a_iy0[4], a_ix0[4], b_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + i * 32 + lrow;
      if (m < a.M) {
        const int b = m / a.OHW;
        const int r = m - b * a.OHW;
        const int oy = r / a.OW;
        const int ox = r - oy * a.OW;
        a_iy0[i] = oy * a.stride - a.pad;
        a_ix0[i] = ox * a.stride - a.pad;
        a_off[i] = (int)(((long long)b * a.x_bstride + ((long long)a_iy0[i] * a.W + a_ix0[i]) * a.ldx) * ES) + chunk * 16;
        if (a.x_up2)   // 1x1 over a nearest-upsampled x: the source pixel of (oy, ox) is (oy / 2, ox / 2) of the half-size tensor
          a_off[i] = (int)(((long long)b * a.x_bstride + ((long long)(oy >> 1) * (a.W >> 1) + (ox >> 1)) * a.ldx) * ES) + chunk * 16;
      } else {
        a_iy0[i] = -(1 << 28);
        a_ix0[i] = -(1 << 28);
        a_off[i] = 0;
      }
      b_off[i] = (n0 + i * 32 + lrow) * a.Kpad * ES + chunk * 16;
    }
    // dual input: K elements from k2_start on are channels of x2 at the OUTPUT pixel (a 1x1 tap of another tensor)
    int a2_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + i * 32 + lrow;
      a2_off[i] = (int)0x80000000;
      if (a.x2 && m < a.M) {
        const int b = m / a.OHW;
        const int r = m - b * a.OHW;
        a2_off[i] = (int)(((long long)b * a.x2_bstride + (long long)r * a.ldx2) * ES) + chunk * 16;
      }
    }
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, g.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx2 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x2 ? a.x2 : a.x), 0, a.x2 ? g.x2_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, g.w_bytes, 0x00020000);
    int k0 = 0, kh = 0, kw = 0, c0 = 0;
    auto issue = [&](int buf) {
      if (g.probe & 4) return;
      char* sa = smem + buf * STAGE + w4 * 1024;
      if (a.x2 && k0 >= a.k2_start) {
        const int d2 = (k0 - a.k2_start) * ES;
#pragma unroll
        for (int i = 0; i < 4; ++i)                                 // rows past M keep the out-of-range bit: zeros
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rx2, (lds_ptr_t)(sa + i * 4096), 16, (unsigned)a2_off[i] + (a2_off[i] < 0 ? 0u : (unsigned)d2), 0, 0, 0);
      } else {
        const int delta = ((kh * a.W + kw) * (int)a.ldx + c0) * ES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int iy = a_iy0[i] + kh, ix = a_ix0[i] + kw;
          const bool ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
          const unsigned vo = ok ? (unsigned)(a_off[i] + delta) : 0x80000000u;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(sa + i * 4096), 16, vo, 0, 0, 0);
        }
      }
#pragma unroll
      for (int i = 0; i < NBI; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(sa + BM * 128 + i * 4096), 16, (unsigned)(b_off[i] + k0 * ES), 0, 0, 0);
      k0 += BK;
      c0 += BK;
      if (c0 >= a.Cin) {
        c0 = 0;
        if (++kw == a.KW) { kw = 0; ++kh; }
      }
    };
    for (int t = 0; t < AHEAD && t < nk; ++t) issue(t);
    for (int ks = 0; ks < nk; ++ks) {
      const int younger = nk - 1 - ks;
      if (STAGES == 4 && younger >= 2) wait_vmcnt<2 * PPT>();
      else if (STAGES >= 3 && younger >= 1) wait_vmcnt<PPT>();
      else wait_vmcnt<0>();
      if (stamps && wv == 4 && ks == 0) stamps[0] = (long long)__builtin_amdgcn_s_memtime() - t_base;
      __builtin_amdgcn_s_barrier();
      if (ks + AHEAD < nk) issue((ks + AHEAD) % STAGES);
    }
  } else {
    // ---- MFMA role ---------------------------------------------------------------------------------
    int foff[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) foff[kk] = (lane & 31) * 128 + (((2 * kk + (lane >> 5)) ^ ((lane >> 1) & 7)) << 4);
    for (int ks = 0; ks < nk; ++ks) {
      __builtin_amdgcn_s_barrier();
      const char* sa = smem + (ks % STAGES) * STAGE + wm * (32 * TJ) * 128;
      const char* sb = smem + (ks % STAGES) * STAGE + (BM + wn * 64) * 128;
      Frag xf[2][TJ], wf[2][2];                    // [register buffer][tile]: fragments of step kk+1 load under the MFMAs of kk
#pragma unroll
      for (int j = 0; j < TJ; ++j) xf[0][j] = *(const Frag*)(sa + j * 4096 + foff[0]);
#pragma unroll
      for (int i = 0; i < 2; ++i) wf[0][i] = *(const Frag*)(sb + i * 4096 + foff[0]);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        if (kk < 3) {
#pragma unroll
          for (int j = 0; j < TJ; ++j) xf[(kk + 1) & 1][j] = *(const Frag*)(sa + j * 4096 + foff[kk + 1]);
#pragma unroll
          for (int i = 0; i < 2; ++i) wf[(kk + 1) & 1][i] = *(const Frag*)(sb + i * 4096 + foff[kk + 1]);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < TJ; ++j) Mma<T>::run(wf[kk & 1][i], xf[kk & 1][j], acc[i][j]);
      }
    }
  }
  if (stamps && wv == 0) stamps[1] = (long long)__builtin_amdgcn_s_memtime() - t_base;
  __syncthreads();                                 // every MFMA operand read is done: smem becomes the fp32 staging tile

  if (g_reg_epilogue_ok(a)) {
    // no residual, bf16 output: bias + activation on the accumulators, bf16 rows through LDS (half the staging bytes), then a
    // plain copy-out - the fp32 staging + per-element epilogue was ~2 us of every launch
    constexpr int SLB = BN + 8;
    bf16* sb = (bf16*)smem;
    if (!loader) {
      const int h = lane >> 5;
      dispatch_act(a.act, [&](auto actc) {
        constexpr int ACT = decltype(actc)::value;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 bv = *(const f32x4*)(a.bias + n0 + wn * 64 + i * 32 + 8 * q + 4 * h);
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
              const int pl = wm * (32 * TJ) + j * 32 + (lane & 31);
              bf16x4 o;
#pragma unroll
              for (int e = 0; e < 4; ++e) o[e] = (bf16)act_c<ACT>(acc[i][j][4 * q + e] + bv[e]);
              *(bf16x4*)(sb + pl * SLB + wn * 64 + i * 32 + 8 * q + 4 * h) = o;
            }
          }
      });
    }
    __syncthreads();
    ws_copy_out_bf16<CITERS, BN>(a, sb, SLB, tid, m0, n0);
    return;
  }
  float* st = (float*)smem;
  if (!loader) {
    const int h = lane >> 5;
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
      const int pl = wm * (32 * TJ) + j * 32 + (lane & 31);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
          *(f32x4*)(&st[pl * SLD + wn * 64 + i * 32 + 8 * q + 4 * h]) = v;
        }
    }
  }
  __syncthreads();
  if (stamps && wv == 0) stamps[2] = (long long)__builtin_amdgcn_s_memtime() - t_base;
  dispatch_act(a.act, [&](auto actc) { ws_copy_out<CITERS, decltype(actc)::value, BN>(a, st, SLD, tid, m0, n0, rpre); });
  if (stamps && wv == 0) {
    stamps[3] = (long long)__builtin_amdgcn_s_memtime() - t_base;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamps[4] = (long long)__builtin_amdgcn_s_memtime() - t_base;
    stamps[7] = (long long)__builtin_amdgcn_s_memrealtime();
  }
}


// ------------------------------------------------------------------------------------------------
// The wave-specialised kernel for PAIR operands (F16X2 tensors: the f16x3 engine's convolution).  The loader role, the LDS image
// (128-byte rows = one [32 hi | 32 lo] channel group, source-side XOR swizzle) and the barrier protocol are conv_igemm_ws_kernel's: to
// the loader a pair tensor is a 16-bit tensor of twice the channels.  The MFMA waves run hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_f16
// (one 32-deep step per K-step: lane l reads row l & 15, 16-byte chunk l >> 4 of the hi half, chunk 4 + (l >> 4) of the lo half -
// conflict-free under the same swizzle); the dropped lo*lo term is 2^-22 relative.  Microbenchmarks with the DMA switched off
// (tools/conv_bench.py --opt glds_drop --vals 0,4) show the MFMA-wave side, not the data movement, bounds this kernel (3x3 256 -> 256 at
// 80^2: 133 us with, 113 us without any DMA), and on random data the chip's clock under MFMA load; the 16x16x32 shape holds a higher
// clock than 32x32x16 for the same flops (measured +3.6 % end to end in round 2; MI355X_MICROARCH.md, DVFS give-back item 7).
template <int STAGES, int BN>
__global__ __launch_bounds__(512, (STAGES == 2 ? 4 : 2)) void conv_igemm_wsx_kernel(const ConvG g) {
  const ConvK& a = g.k;
  constexpr int BM = 128;
  static_assert(BN == 128 || BN == 64, "tile widths");
  constexpr int TJ = BN == 128 ? 2 : 1;          // 32-pixel tiles per MFMA wave; 64 channels per wave
  constexpr int NBI = BN / 32;
  constexpr int PPT = 4 + NBI;
  constexpr int CH8 = BN / 8, RSTEP = 512 / CH8, CITERS = BM / RSTEP;
  constexpr int BK = 64;                          // sp16 elements per K-step = 32 channels x (hi, lo)
  constexpr int STAGE = (BM + BN) * 128;
  constexpr int SLD = BN + 4;
  constexpr int SMEM = (STAGES * STAGE > BM * SLD * 4) ? STAGES * STAGE : BM * SLD * 4;
  constexpr int AHEAD = STAGES - 1;
  static_assert(STAGES >= 2 && STAGES <= 4, "counted waits below assume 1..3 tiles ahead");
  __shared__ __attribute__((aligned(16))) char smem[SMEM + 256];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wv >= 4;
  const int w4 = wv & 3;
  const int wm = BN == 128 ? (w4 & 1) : w4, wn = BN == 128 ? (w4 >> 1) : 0;
  int wg;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  // two-pass split-K (g.splitk = S > 1): the grid is S x tiles, slice-major; slice sl multiplies the channel groups [sl, sl + 1) Cin / S of every
  // tap and writes bare partial sums to its slab slice (ConvK::raw, set up by the host); k_splitk_reduce finishes the layer
  int sl = 0;
  if (g.splitk > 1) { const int tiles = (int)gridDim.x / g.splitk; sl = wg / tiles; wg -= sl * tiles; }
  const long long yoff = (long long)sl * a.M * a.N;
  const int nt = wg % a.ntn, mt = wg / a.ntn;
  const int m0 = mt * BM, n0 = nt * BN;
  const int nk = a.Kpad / BK / g.splitk;

  // accumulators: 16x16 tiles [4 channel tiles][2 TJ pixel tiles] x 4 floats
  f32x4 acc16[4][2 * TJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2 * TJ; ++j) acc16[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (!loader) prefetch_share(a, blockIdx.x, gridDim.x, tid, 256, smem + SMEM);

  if (loader) {
    // ---- loader role (conv_igemm_ws_kernel's) ---------------------------------------------------------
    const int lrow = w4 * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((w4 * 4 + (lane >> 4)) & 7);
    int a_off[4], a_iy0[4], a_ix0[4], b_off[4], a2_off[4];
    PixWalk pw;
    pix_init(a, m0 + lrow, pw);
    const int dq32 = 32 / a.OW, dr32 = 32 - dq32 * a.OW;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + i * 32 + lrow;
      a2_off[i] = (int)0x80000000;
      const int b = pw.b, oy = pw.oy, ox = pw.ox, r = oy * a.OW + ox;
      pix_step32(a, dq32, dr32, pw);
      if (m < a.M) {
        a_iy0[i] = oy * a.stride - a.pad;
        a_ix0[i] = ox * a.stride - a.pad;
        a_off[i] = (int)(((long long)b * a.x_bstride + ((long long)a_iy0[i] * a.W + a_ix0[i]) * a.ldx) * 2) + chunk * 16;
        if (a.x_up2) a_off[i] = (int)(((long long)b * a.x_bstride + ((long long)(oy >> 1) * (a.W >> 1) + (ox >> 1)) * a.ldx) * 2) + chunk * 16;
        if (a.x2) a2_off[i] = (int)(((long long)b * a.x2_bstride + (long long)r * a.ldx2) * 2) + chunk * 16;
      } else {
        a_iy0[i] = -(1 << 28);
        a_ix0[i] = -(1 << 28);
        a_off[i] = 0;
      }
      b_off[i] = (n0 + i * 32 + lrow) * a.Kpad * 2 + chunk * 16;
    }
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, g.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx2 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x2 ? a.x2 : a.x), 0, a.x2 ? g.x2_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, g.w_bytes, 0x00020000);
    // K order: channel group outer, taps inner (the filter keeps its tap-major layout, only the walk changes).  With taps outer a pixel's 128-byte
    // group slice is re-read by the nine taps eight K-steps apart - 8 MB of other slices per XCD in between, more than its L2 holds: the 3x3 layers
    // at 80^2 fetched their input 4-6 times from beyond L2.  Every F16X2 tile kernel walks K the same way (one summation order per output).
    // (one walk only: offered both at run time, hipcc merges the two mirror-image counters by selecting a POINTER to kh / kw / c0 and
    // keeps them in scratch memory - the loaders then run a scratch round trip per K-step and every layer is 40-60 % slower)
    const int nk_main = a.x2 ? a.k2_start / BK : nk;
    int ksi = 0, kh = 0, kw = 0, c0 = sl * (a.Cin / g.splitk);
    auto issue = [&](int buf) __attribute__((always_inline)) {      // (called twice: left to the inliner's budget, its captures live in scratch)
      if (g.probe & 4) return;
      char* sa = smem + buf * STAGE + w4 * 1024;
      int kf;                                                   // filter column of this K-step (sp16 elements)
      if (ksi >= nk_main) {
        const int d2 = (ksi - nk_main) * BK * 2;
        kf = a.k2_start + (ksi - nk_main) * BK;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rx2, (lds_ptr_t)(sa + i * 4096), 16, (unsigned)a2_off[i] + (a2_off[i] < 0 ? 0u : (unsigned)d2), 0, 0, 0);
      } else {
        const int delta = ((kh * a.W + kw) * (int)a.ldx + c0) * 2;
        kf = (kh * a.KW + kw) * a.Cin + c0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int iy = a_iy0[i] + kh, ix = a_ix0[i] + kw;
          const bool ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
          const unsigned vo = ok ? (unsigned)(a_off[i] + delta) : 0x80000000u;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(sa + i * 4096), 16, vo, 0, 0, 0);
        }
      }
#pragma unroll
      for (int i = 0; i < NBI; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(sa + BM * 128 + i * 4096), 16, (unsigned)(b_off[i] + kf * 2), 0, 0, 0);
      ++ksi;
      if (++kw == a.KW) {
        kw = 0;
        if (++kh == a.KH) { kh = 0; c0 += BK; }
      }
    };
    for (int t = 0; t < AHEAD && t < nk; ++t) issue(t);
    for (int ks = 0; ks < nk; ++ks) {
      const int younger = nk - 1 - ks;
      if (STAGES == 4 && younger >= 2) wait_vmcnt<2 * PPT>();
      else if (STAGES >= 3 && younger >= 1) wait_vmcnt<PPT>();
      else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      if (ks + AHEAD < nk) issue((ks + AHEAD) % STAGES);
    }
  } else {
    // ---- MFMA role, 16x16x32: one 32-deep step per K-step.  The wave's 8 filter fragments stay live; pixel fragments come tile by tile ----
    const int r16 = lane & 15, c4 = lane >> 4;
    const int sw = (r16 >> 1) & 7;
    const int foh = r16 * 128 + ((c4 ^ sw) << 4), fol = r16 * 128 + (((c4 + 4) ^ sw) << 4);
    for (int ks = 0; ks < nk; ++ks) {
      __builtin_amdgcn_s_barrier();
      const char* sa = smem + (ks % STAGES) * STAGE + wm * (32 * TJ) * 128;
      const char* sb = smem + (ks % STAGES) * STAGE + (BM + wn * 64) * 128;
      sp16x8 wh[4], wl[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { wh[i] = *(const sp16x8*)(sb + i * 2048 + foh); wl[i] = *(const sp16x8*)(sb + i * 2048 + fol); }
#pragma unroll
      for (int j = 0; j < 2 * TJ; ++j) {
        const sp16x8 xh = *(const sp16x8*)(sa + j * 2048 + foh), xl = *(const sp16x8*)(sa + j * 2048 + fol);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc16[i][j] = mfma_pair16(wh[i], xh, acc16[i][j]);
          acc16[i][j] = mfma_pair16(wh[i], xl, acc16[i][j]);
          acc16[i][j] = mfma_pair16(wl[i], xh, acc16[i][j]);
        }
      }
    }
  }
  __syncthreads();                                 // every MFMA operand read is done: smem becomes the staging tile

  // every accumulator group = 4 consecutive channels (cl..cl+3 inside the tile) of one pixel (row pl of the tile)
  auto for_each_group = [&](auto&& f) {
    const int r16 = lane & 15, c4 = lane >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2 * TJ; ++j) f(wn * 64 + 16 * i + 4 * c4, wm * (32 * TJ) + 16 * j + r16, acc16[i][j]);
  };
  if (a.y_split && a.res_mode == RES_NONE) {
    // bias + activation + hi/lo split on the accumulators, the tile's F16X2 rows through LDS, plain copy-out
    constexpr int SLB = 2 * BN + 8;
    sp16* sb = (sp16*)smem;
    if (!loader) {
      dispatch_act(a.act, [&](auto actc) {
        constexpr int ACT = decltype(actc)::value;
        for_each_group([&](int cl, int pl, const f32x4& v) {
          const f32x4 bv = *(const f32x4*)(a.bias + n0 + cl);
          sp16x4 oh, ol;
#pragma unroll
          for (int e = 0; e < 4; ++e) { sp16 hi, lo; split2(act_c<ACT>(v[e] + bv[e]), hi, lo); oh[e] = hi; ol[e] = lo; }
          sp16* d = sb + pl * SLB + ((cl >> 5) << 6) + (cl & 31);
          *(sp16x4*)d = oh;
          *(sp16x4*)(d + SPLIT_GROUP) = ol;
        });
      });
    }
    __syncthreads();
    ws_copy_out_split_rows<BN>(a, sb, SLB, tid, m0, n0);
    return;
  }
  float* st = (float*)smem;
  if (!loader) for_each_group([&](int cl, int pl, const f32x4& v) { *(f32x4*)(&st[pl * SLD + cl]) = v; });
  __syncthreads();
  dispatch_act(a.act, [&](auto actc) { ws_copy_out_sp<CITERS, decltype(actc)::value, BN>(a, st, SLD, tid, m0, n0, yoff); });
}


// ------------------------------------------------------------------------------------------------
// v4f: the F16X2 kernel on a tile of FLEXIBLE height: MT pixel tiles of 16 rows (MT = 4 .. 13: 64 .. 208 pixels) x BN channels.
// The MFMA-bound layers lose ~20 % to tile-count quantization with fixed 128 x 128 tiles (3x3 256 -> 256 at 80^2 = 800 tiles takes as long
// as 1014 tiles would: tools/conv_bench.py --only quant; 40^2 maps give 200 tiles for 256 CUs, 20^2 maps 100).  Here the host picks the
// tile height so that the grid is close to a whole number of rounds of the chip (launch_conv_split), e.g. 208 x 128 for 80^2 x 256
// channels: 494 blocks = 2 rounds of one block per CU, 96 % full.
// Wave layout 1 x 4: every MFMA wave owns ALL MT pixel tiles and BN / 4 channels (any MT balances), v_mfma_f32_16x16x32_f16, its
// 2 (BN = 128) or 1 (BN = 64) filter fragment pairs live for the K-step, pixel fragments tile by tile.  Loader role, LDS image, swizzle and
// barrier protocol as conv_igemm_wsx_kernel; the A part of a stage holds AROWS = MT x 16 rounded up to 32 rows.
template <int STAGES, int BN, int MT>
__global__ __launch_bounds__(512, (STAGES == 2 ? 4 : 2)) void conv_igemm_wsf_kernel(const ConvG g) {
  const ConvK& a = g.k;
  constexpr int BM = MT * 16;
  constexpr int AROWS = (BM + 31) / 32 * 32;
  constexpr int NAI = AROWS / 32, NBI = BN / 32;
  constexpr int PPT = NAI + NBI;
  constexpr int CT = BN / 64;                     // 16-channel tiles per MFMA wave
  constexpr int CH8 = BN / 8, RSTEP = 512 / CH8, CITERS = (BM + RSTEP - 1) / RSTEP;
  constexpr int BK = 64;
  constexpr int STAGE = (AROWS + BN) * 128;
  constexpr int SLD = BN + 4, SLB = 2 * BN + 8;
  constexpr int EPI = (BM * SLD * 4 > BM * SLB * 2) ? BM * SLD * 4 : BM * SLB * 2;
  constexpr int SMEM = (STAGES * STAGE > EPI) ? STAGES * STAGE : EPI;
  constexpr int AHEAD = STAGES - 1;
  static_assert(STAGES >= 2 && STAGES <= 4 && MT >= 1 && MT <= 14 && (BN == 64 || BN == 128), "tile shape");
  static_assert(SMEM + 256 <= 160 * 1024, "LDS");
  __shared__ __attribute__((aligned(16))) char smem[SMEM + 256];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wv >= 4;
  const int w4 = wv & 3;
  int wg;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  // two-pass split-K (g.splitk = S > 1): the grid is S x tiles, slice-major; slice sl multiplies the channel groups [sl, sl + 1) Cin / S of every
  // tap and writes bare partial sums to its slab slice (ConvK::raw, set up by the host); k_splitk_reduce finishes the layer
  int sl = 0;
  if (g.splitk > 1) { const int tiles = (int)gridDim.x / g.splitk; sl = wg / tiles; wg -= sl * tiles; }
  const long long yoff = (long long)sl * a.M * a.N;
  const int nt = wg % a.ntn, mt = wg / a.ntn;
  const int m0 = mt * BM, n0 = nt * BN;
  const int nk = a.Kpad / BK / g.splitk;

  f32x4 acc[CT][MT];
#pragma unroll
  for (int i = 0; i < CT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (!loader) prefetch_share(a, blockIdx.x, gridDim.x, tid, 256, smem + SMEM);

  if (loader) {
    const int lrow = w4 * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((w4 * 4 + (lane >> 4)) & 7);
    int a_off[NAI], a_iy0[NAI], a_ix0[NAI], b_off[NBI];
    unsigned a2_off[8];   // fixed extent (NAI <= 7): with a dependent extent hipcc (ROCm 7.2) silently drops the HOST stub of every instantiation
    static_assert(NAI <= 8, "a2_off");
    PixWalk pw;
    pix_init(a, m0 + lrow, pw);
    const int dq32 = 32 / a.OW, dr32 = 32 - dq32 * a.OW;
#pragma unroll
    for (int i = 0; i < NAI; ++i) {
      const int row = i * 32 + lrow;
      const int m = m0 + row;
      a2_off[i] = 0x80000000u;
      const int b = pw.b, oy = pw.oy, ox = pw.ox, r = oy * a.OW + ox;
      pix_step32(a, dq32, dr32, pw);
      if (m < a.M && row < BM) {
        a_iy0[i] = oy * a.stride - a.pad;
        a_ix0[i] = ox * a.stride - a.pad;
        a_off[i] = (int)(((long long)b * a.x_bstride + ((long long)a_iy0[i] * a.W + a_ix0[i]) * a.ldx) * 2) + chunk * 16;
        if (a.x_up2) a_off[i] = (int)(((long long)b * a.x_bstride + ((long long)(oy >> 1) * (a.W >> 1) + (ox >> 1)) * a.ldx) * 2) + chunk * 16;
        if (a.x2) a2_off[i] = (unsigned)(((long long)b * a.x2_bstride + (long long)r * a.ldx2) * 2) + chunk * 16;
      } else {
        a_iy0[i] = -(1 << 28);
        a_ix0[i] = -(1 << 28);
        a_off[i] = 0;
      }
    }
#pragma unroll
    for (int i = 0; i < NBI; ++i) b_off[i] = (n0 + i * 32 + lrow) * a.Kpad * 2 + chunk * 16;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, g.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx2 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x2 ? a.x2 : a.x), 0, a.x2 ? g.x2_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, g.w_bytes, 0x00020000);
    // K order: channel group outer, taps inner (the filter keeps its tap-major layout, only the walk changes).  With taps outer a pixel's 128-byte
    // group slice is re-read by the nine taps eight K-steps apart - 8 MB of other slices per XCD in between, more than its L2 holds: the 3x3 layers
    // at 80^2 fetched their input 4-6 times from beyond L2.  Every F16X2 tile kernel walks K the same way (one summation order per output).
    // (one walk only: offered both at run time, hipcc merges the two mirror-image counters by selecting a POINTER to kh / kw / c0 and
    // keeps them in scratch memory - the loaders then run a scratch round trip per K-step and every layer is 40-60 % slower)
    const int nk_main = a.x2 ? a.k2_start / BK : nk;
    int ksi = 0, kh = 0, kw = 0, c0 = sl * (a.Cin / g.splitk);
    auto issue = [&](int buf) __attribute__((always_inline)) {      // (called twice: left to the inliner's budget, its captures live in scratch)
      if (g.probe & 4) return;
      char* sa = smem + buf * STAGE + w4 * 1024;
      int kf;                                                   // filter column of this K-step (sp16 elements)
      if (ksi >= nk_main) {
        const int d2 = (ksi - nk_main) * BK * 2;
        kf = a.k2_start + (ksi - nk_main) * BK;
#pragma unroll
        for (int i = 0; i < NAI; ++i)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rx2, (lds_ptr_t)(sa + i * 4096), 16, a2_off[i] + ((a2_off[i] >> 31) ? 0u : (unsigned)d2), 0, 0, 0);
      } else {
        const int delta = ((kh * a.W + kw) * (int)a.ldx + c0) * 2;
        kf = (kh * a.KW + kw) * a.Cin + c0;
#pragma unroll
        for (int i = 0; i < NAI; ++i) {
          const int iy = a_iy0[i] + kh, ix = a_ix0[i] + kw;
          const bool ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
          const unsigned vo = ok ? (unsigned)(a_off[i] + delta) : 0x80000000u;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(sa + i * 4096), 16, vo, 0, 0, 0);
        }
      }
#pragma unroll
      for (int i = 0; i < NBI; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(sa + AROWS * 128 + i * 4096), 16, (unsigned)(b_off[i] + kf * 2), 0, 0, 0);
      ++ksi;
      if (++kw == a.KW) {
        kw = 0;
        if (++kh == a.KH) { kh = 0; c0 += BK; }
      }
    };
    for (int t = 0; t < AHEAD && t < nk; ++t) issue(t);
    for (int ks = 0; ks < nk; ++ks) {
      const int younger = nk - 1 - ks;
      if (STAGES == 4 && younger >= 2) wait_vmcnt<2 * PPT>();
      else if (STAGES >= 3 && younger >= 1) wait_vmcnt<PPT>();
      else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      if (ks + AHEAD < nk) issue((ks + AHEAD) % STAGES);
    }
  } else {
    const int r16 = lane & 15, c4 = lane >> 4;
    const int sw = (r16 >> 1) & 7;
    const int foh = r16 * 128 + ((c4 ^ sw) << 4), fol = r16 * 128 + (((c4 + 4) ^ sw) << 4);
    if constexpr (STAGES >= 3 && MT >= 2) {
      // One block per CU (one MFMA wave per SIMD): conv_igemm_wsq_kernel's K-step schedule - the last tile's MFMAs are deferred past the next
      // barrier and run from registers beside the new step's first fragment reads, never more than one ds_read_b128 per MFMA gap, filter
      // fragments double-buffered by step parity.  (The 2-stage form runs two blocks per CU inside 128 registers: its waves cover each other.)
      sp16x8 wh[2][CT], wl[2][CT], xh[2], xl[2], ph, pl;
#pragma unroll
      for (int i = 0; i < CT; ++i) { wh[1][i] = sp16x8{0, 0, 0, 0, 0, 0, 0, 0}; wl[1][i] = wh[1][i]; }
      ph = sp16x8{0, 0, 0, 0, 0, 0, 0, 0}; pl = ph;
#define RTD_SB() __builtin_amdgcn_sched_barrier(0)
#define RTD_MF(A, B, C) C = mfma_pair16(A, B, C)
      auto step = [&](auto parc, const int ks) __attribute__((always_inline)) {
        constexpr int P = decltype(parc)::value, Q = P ^ 1;
        __builtin_amdgcn_s_barrier();
        const char* sa = smem + (ks % STAGES) * STAGE;
        const char* sb = sa + (AROWS + w4 * (BN / 4)) * 128;
        wh[P][0] = *(const sp16x8*)(sb + foh); xh[0] = *(const sp16x8*)(sa + foh); RTD_SB();
        RTD_MF(wh[Q][0], ph, acc[0][MT - 1]);
        if (CT == 2) { wh[P][1] = *(const sp16x8*)(sb + 2048 + foh); RTD_SB(); RTD_MF(wh[Q][1], ph, acc[1][MT - 1]); }
        xl[0] = *(const sp16x8*)(sa + fol); RTD_SB();
        RTD_MF(wh[Q][0], pl, acc[0][MT - 1]); wl[P][0] = *(const sp16x8*)(sb + fol); RTD_SB();
        if (CT == 2) { RTD_MF(wh[Q][1], pl, acc[1][MT - 1]); wl[P][1] = *(const sp16x8*)(sb + 2048 + fol); RTD_SB(); }
        RTD_MF(wl[Q][0], ph, acc[0][MT - 1]);
        if (CT == 2) RTD_MF(wl[Q][1], ph, acc[1][MT - 1]);
        RTD_SB();
#pragma unroll
        for (int j = 0; j < MT - 1; ++j) {
          RTD_MF(wh[P][0], xh[j & 1], acc[0][j]); xh[(j + 1) & 1] = *(const sp16x8*)(sa + (j + 1) * 2048 + foh); RTD_SB();
          if (CT == 2) RTD_MF(wh[P][1], xh[j & 1], acc[1][j]);
          RTD_MF(wh[P][0], xl[j & 1], acc[0][j]); xl[(j + 1) & 1] = *(const sp16x8*)(sa + (j + 1) * 2048 + fol); RTD_SB();
          if (CT == 2) RTD_MF(wh[P][1], xl[j & 1], acc[1][j]);
          RTD_MF(wl[P][0], xh[j & 1], acc[0][j]);
          if (CT == 2) RTD_MF(wl[P][1], xh[j & 1], acc[1][j]);
          RTD_SB();
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // every read of this stage has returned before the next barrier
        ph = xh[(MT - 1) & 1]; pl = xl[(MT - 1) & 1];
      };
      int ks = 0;
      for (; ks + 1 < nk; ks += 2) { step(ActC<0>{}, ks); step(ActC<1>{}, ks + 1); }
      if (ks < nk) {
        step(ActC<0>{}, ks);
#pragma unroll
        for (int i = 0; i < CT; ++i) RTD_MF(wh[0][i], ph, acc[i][MT - 1]);
#pragma unroll
        for (int i = 0; i < CT; ++i) RTD_MF(wh[0][i], pl, acc[i][MT - 1]);
#pragma unroll
        for (int i = 0; i < CT; ++i) RTD_MF(wl[0][i], ph, acc[i][MT - 1]);
      } else {
#pragma unroll
        for (int i = 0; i < CT; ++i) RTD_MF(wh[1][i], ph, acc[i][MT - 1]);
#pragma unroll
        for (int i = 0; i < CT; ++i) RTD_MF(wh[1][i], pl, acc[i][MT - 1]);
#pragma unroll
        for (int i = 0; i < CT; ++i) RTD_MF(wl[1][i], ph, acc[i][MT - 1]);
      }
#undef RTD_SB
#undef RTD_MF
    } else {
    for (int ks = 0; ks < nk; ++ks) {
      __builtin_amdgcn_s_barrier();
      const char* sa = smem + (ks % STAGES) * STAGE;
      const char* sb = sa + (AROWS + w4 * (BN / 4)) * 128;
      sp16x8 wh[CT], wl[CT];
#pragma unroll
      for (int i = 0; i < CT; ++i) { wh[i] = *(const sp16x8*)(sb + i * 2048 + foh); wl[i] = *(const sp16x8*)(sb + i * 2048 + fol); }
      // pixel fragments of tile j + 1 are read while the MFMAs of tile j issue (pinned: with one MFMA wave per SIMD an LDS round trip in
      // front of every tile's MFMAs idles the matrix pipe)
      sp16x8 xh[2], xl[2];
      xh[0] = *(const sp16x8*)(sa + foh); xl[0] = *(const sp16x8*)(sa + fol);
#pragma unroll
      for (int j = 0; j < MT; ++j) {
        if (j + 1 < MT) { xh[(j + 1) & 1] = *(const sp16x8*)(sa + (j + 1) * 2048 + foh); xl[(j + 1) & 1] = *(const sp16x8*)(sa + (j + 1) * 2048 + fol); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < CT; ++i) {
          acc[i][j] = mfma_pair16(wh[i], xh[j & 1], acc[i][j]);
          acc[i][j] = mfma_pair16(wh[i], xl[j & 1], acc[i][j]);
          acc[i][j] = mfma_pair16(wl[i], xh[j & 1], acc[i][j]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    }
  }
  __syncthreads();

  auto for_each_group = [&](auto&& f) {
    const int r16 = lane & 15, c4 = lane >> 4;
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
      for (int j = 0; j < MT; ++j) f(w4 * (BN / 4) + 16 * i + 4 * c4, 16 * j + r16, acc[i][j]);
  };
  if (a.y_split && a.res_mode == RES_NONE) {
    sp16* sb = (sp16*)smem;
    if (!loader) {
      dispatch_act(a.act, [&](auto actc) {
        constexpr int ACT = decltype(actc)::value;
        for_each_group([&](int cl, int pl, const f32x4& v) {
          const f32x4 bv = *(const f32x4*)(a.bias + n0 + cl);
          sp16x4 oh, ol;
#pragma unroll
          for (int e = 0; e < 4; ++e) { sp16 hi, lo; split2(act_c<ACT>(v[e] + bv[e]), hi, lo); oh[e] = hi; ol[e] = lo; }
          sp16* d = sb + pl * SLB + ((cl >> 5) << 6) + (cl & 31);
          *(sp16x4*)d = oh;
          *(sp16x4*)(d + SPLIT_GROUP) = ol;
        });
      });
    }
    __syncthreads();
    ws_copy_out_split_rows<BN, BM>(a, sb, SLB, tid, m0, n0);
    return;
  }
  float* st = (float*)smem;
  if (!loader) for_each_group([&](int cl, int pl, const f32x4& v) { *(f32x4*)(&st[pl * SLD + cl]) = v; });
  __syncthreads();
  dispatch_act(a.act, [&](auto actc) { ws_copy_out_sp<CITERS, decltype(actc)::value, BN, BM>(a, st, SLD, tid, m0, n0, yoff); });
}


// ------------------------------------------------------------------------------------------------
// v4q: the F16X2 kernel for grids of MORE than one round of the chip: (MTA + MTB) x 16 pixels x 128 channels per block (160 .. 256 pixels),
// 2 x 2 wave layout, 3 stages, one block per CU.  The 128 x 128 kernel's K-step keeps LDS exactly as busy as the matrix pipes (16 fragment
// reads per 48 MFMAs and wave + 32 KiB of DMA writes: 768 cycles each) and its grids quantize badly (800 tiles on 512
// slots).  Here an MFMA wave owns MT x 4 accumulator tiles (MT = MTA for the upper row half, MTB for the lower; 64 channels): 2 MT + 8
// fragment reads per 12 MT MFMAs (7 x 4: 22 reads per 84 MFMAs, 3.8 MFMAs per read instead of 3), one filter tile staged per 256 instead
// of 128 pixels, 12 instead of 16 DMA instructions per loader wave and 1536 MFMA cycles, and the host picks MTA + MTB so that the grid is
// close to whole rounds of 256 blocks (launch_conv_split: 224 x 128 for 80^2 x 8 x 256 channels = 458 blocks, 160 x 128 for 512 channels
// = 1280 blocks = 5 rounds).  The pixel fragments of tile j + 1 are read under the MFMAs of tile j.  Loader role, LDS image, swizzle,
// barrier protocol, K walk and the order of the three products are conv_igemm_wsx_kernel's: an output's arithmetic does not depend on which
// of the tile kernels ran it (bit-identical results, tested).
template <int MTA, int MTB, bool STAMP = false>   // STAMP: the diagnostic build (tools/conv_bench.py with glds_drop = 32 and RTD_CONV_STAMPS=2), never dispatched by a plan
__global__ __launch_bounds__(512, 2) void conv_igemm_wsq_kernel(const ConvG g) {
  const ConvK& a = g.k;
  constexpr int BN = 128, STAGES = 3;
  constexpr int BM = (MTA + MTB) * 16;
  constexpr int AROWS = (BM + 31) / 32 * 32;
  constexpr int NAI = AROWS / 32, NBI = BN / 32;
  constexpr int PPT = NAI + NBI;
  constexpr int MTX = MTA > MTB ? MTA : MTB;
  constexpr int CH8 = BN / 8, RSTEP = 512 / CH8, CITERS = (BM + RSTEP - 1) / RSTEP;
  constexpr int BK = 64;
  constexpr int STAGE = (AROWS + BN) * 128;
  constexpr int SLD = BN + 4, SLB = 2 * BN + 8;
  constexpr int EPI = (BM * SLD * 4 > BM * SLB * 2) ? BM * SLD * 4 : BM * SLB * 2;
  constexpr int SMEM = (STAGES * STAGE > EPI) ? STAGES * STAGE : EPI;
  constexpr int AHEAD = STAGES - 1;
  static_assert(MTA >= MTB && MTB >= 1 && MTA <= 8, "row split");
  static_assert(SMEM + 256 <= 160 * 1024, "LDS");
  __shared__ __attribute__((aligned(16))) char smem[SMEM + 256];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wv >= 4;
  const int w4 = wv & 3;
  const int wm = w4 & 1, wn = w4 >> 1;
  int wg;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int nt = wg % a.ntn, mt = wg / a.ntn;
  const int m0 = mt * BM, n0 = nt * BN;
  const int nk = a.Kpad / BK;
  long long* stamps = (STAMP && g.slab && blockIdx.x < 4096 && lane == 0) ? (long long*)g.slab + (size_t)blockIdx.x * 8 : nullptr;
  const long long t_base = STAMP ? (long long)__builtin_amdgcn_s_memtime() : 0;
  if (STAMP && stamps && wv == 0) stamps[6] = (long long)__builtin_amdgcn_s_memrealtime();

  f32x4 acc[4][MTX];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < MTX; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (!loader) prefetch_share(a, blockIdx.x, gridDim.x, tid, 256, smem + SMEM);

  if (loader) {
    const int lrow = w4 * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((w4 * 4 + (lane >> 4)) & 7);
    int a_off[NAI], a_iy0[NAI], a_ix0[NAI], b_off[NBI];
    unsigned a2_off[8];   // fixed extent (see conv_igemm_wsf_kernel)
    static_assert(NAI <= 8, "a2_off");
    PixWalk pw;
    pix_init(a, m0 + lrow, pw);
    const int dq32 = 32 / a.OW, dr32 = 32 - dq32 * a.OW;
#pragma unroll
    for (int i = 0; i < NAI; ++i) {
      const int row = i * 32 + lrow;
      const int m = m0 + row;
      a2_off[i] = 0x80000000u;
      const int b = pw.b, oy = pw.oy, ox = pw.ox, r = oy * a.OW + ox;
      pix_step32(a, dq32, dr32, pw);
      if (m < a.M && row < BM) {
        a_iy0[i] = oy * a.stride - a.pad;
        a_ix0[i] = ox * a.stride - a.pad;
        a_off[i] = (int)(((long long)b * a.x_bstride + ((long long)a_iy0[i] * a.W + a_ix0[i]) * a.ldx) * 2) + chunk * 16;
        if (a.x_up2) a_off[i] = (int)(((long long)b * a.x_bstride + ((long long)(oy >> 1) * (a.W >> 1) + (ox >> 1)) * a.ldx) * 2) + chunk * 16;
        if (a.x2) a2_off[i] = (unsigned)(((long long)b * a.x2_bstride + (long long)r * a.ldx2) * 2) + chunk * 16;
      } else {
        a_iy0[i] = -(1 << 28);
        a_ix0[i] = -(1 << 28);
        a_off[i] = 0;
      }
    }
#pragma unroll
    for (int i = 0; i < NBI; ++i) b_off[i] = (n0 + i * 32 + lrow) * a.Kpad * 2 + chunk * 16;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, g.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx2 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x2 ? a.x2 : a.x), 0, a.x2 ? g.x2_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, g.w_bytes, 0x00020000);
    // K order: channel group outer, taps inner (conv_igemm_wsx_kernel)
    const int nk_main = a.x2 ? a.k2_start / BK : nk;
    int ksi = 0, kh = 0, kw = 0, c0 = 0;
    auto issue = [&](int buf) __attribute__((always_inline)) {
      if (g.probe & 4) return;
      char* sa = smem + buf * STAGE + w4 * 1024;
      int kf;
      if (ksi >= nk_main) {
        const int d2 = (ksi - nk_main) * BK * 2;
        kf = a.k2_start + (ksi - nk_main) * BK;
#pragma unroll
        for (int i = 0; i < NAI; ++i)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rx2, (lds_ptr_t)(sa + i * 4096), 16, a2_off[i] + ((a2_off[i] >> 31) ? 0u : (unsigned)d2), 0, 0, 0);
      } else {
        const int delta = ((kh * a.W + kw) * (int)a.ldx + c0) * 2;
        kf = (kh * a.KW + kw) * a.Cin + c0;
#pragma unroll
        for (int i = 0; i < NAI; ++i) {
          const int iy = a_iy0[i] + kh, ix = a_ix0[i] + kw;
          const bool ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
          const unsigned vo = ok ? (unsigned)(a_off[i] + delta) : 0x80000000u;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(sa + i * 4096), 16, vo, 0, 0, 0);
        }
      }
#pragma unroll
      for (int i = 0; i < NBI; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(sa + AROWS * 128 + i * 4096), 16, (unsigned)(b_off[i] + kf * 2), 0, 0, 0);
      ++ksi;
      if (++kw == a.KW) {
        kw = 0;
        if (++kh == a.KH) { kh = 0; c0 += BK; }
      }
    };
    for (int t = 0; t < AHEAD && t < nk; ++t) issue(t);
    for (int ks = 0; ks < nk; ++ks) {
      if (nk - 1 - ks >= 1) wait_vmcnt<PPT>();
      else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      if (ks + AHEAD < nk) issue((ks + AHEAD) % STAGES);
    }
  } else {
    const int r16 = lane & 15, c4 = lane >> 4;
    const int sw = (r16 >> 1) & 7;
    const int foh = r16 * 128 + ((c4 ^ sw) << 4), fol = r16 * 128 + (((c4 + 4) ^ sw) << 4);
    // the wave's row half: MT tiles from row `rb` (wave-uniform branch; each body has a compile-time tile count).
    // Schedule of a K-step (tools/kstep_probe.hip: every ds_read_b128 costs the matrix pipe ~10 cycles wherever it sits, a barrier ~90, and
    // the round trip of a step's first reads another ~100 when nothing covers it - 1675 cycles for 1344 of MFMA with all ten first reads
    // up front, 1588 like this): the MFMAs of a step's LAST tile are deferred past the next barrier, where they run from registers one to
    // one with the new step's first ten fragment reads; the two reads of pixel tile j + 1 follow the first and the fifth MFMA of tile j.
    // Filter fragments are double-buffered by step parity (two steps per loop trip).  Per accumulator the order of the products is
    // conv_igemm_wsx_kernel's (hh, hl, lh per K-step); the first step's deferred slot multiplies zeros (+0 onto +0).
    auto body = [&](auto mtc, const int rb) __attribute__((always_inline)) {
      constexpr int MT = decltype(mtc)::value;
      sp16x8 wh[2][4], wl[2][4];
      sp16x8 xh[2], xl[2], ph, pl;                               // ph / pl: pixel fragments of the deferred tile
#pragma unroll
      for (int i = 0; i < 4; ++i) { wh[1][i] = sp16x8{0, 0, 0, 0, 0, 0, 0, 0}; wl[1][i] = wh[1][i]; }
      ph = sp16x8{0, 0, 0, 0, 0, 0, 0, 0}; pl = ph;
#define RTD_SB() __builtin_amdgcn_sched_barrier(0)
#define RTD_MF(A, B, C) C = mfma_pair16(A, B, C)
      auto step = [&](auto parc, const int ks) __attribute__((always_inline)) {
        constexpr int P = decltype(parc)::value, Q = P ^ 1;
        __builtin_amdgcn_s_barrier();
        if (STAMP && stamps && wv == 0 && ks == 0) stamps[0] = (long long)__builtin_amdgcn_s_memtime() - t_base;
        const char* sa = smem + (ks % STAGES) * STAGE + rb * 128;
        const char* sb = smem + (ks % STAGES) * STAGE + (AROWS + wn * 64) * 128;
        wh[P][0] = *(const sp16x8*)(sb + foh); xh[0] = *(const sp16x8*)(sa + foh); RTD_SB();
        RTD_MF(wh[Q][0], ph, acc[0][MT - 1]); wh[P][1] = *(const sp16x8*)(sb + 2048 + foh); RTD_SB();
        RTD_MF(wh[Q][1], ph, acc[1][MT - 1]); wh[P][2] = *(const sp16x8*)(sb + 4096 + foh); RTD_SB();
        RTD_MF(wh[Q][2], ph, acc[2][MT - 1]); wh[P][3] = *(const sp16x8*)(sb + 6144 + foh); RTD_SB();
        RTD_MF(wh[Q][3], ph, acc[3][MT - 1]); xl[0] = *(const sp16x8*)(sa + fol); RTD_SB();
        RTD_MF(wh[Q][0], pl, acc[0][MT - 1]); wl[P][0] = *(const sp16x8*)(sb + fol); RTD_SB();
        RTD_MF(wh[Q][1], pl, acc[1][MT - 1]); wl[P][1] = *(const sp16x8*)(sb + 2048 + fol); RTD_SB();
        RTD_MF(wh[Q][2], pl, acc[2][MT - 1]); wl[P][2] = *(const sp16x8*)(sb + 4096 + fol); RTD_SB();
        RTD_MF(wh[Q][3], pl, acc[3][MT - 1]); wl[P][3] = *(const sp16x8*)(sb + 6144 + fol); RTD_SB();
        RTD_MF(wl[Q][0], ph, acc[0][MT - 1]); RTD_MF(wl[Q][1], ph, acc[1][MT - 1]); RTD_MF(wl[Q][2], ph, acc[2][MT - 1]); RTD_MF(wl[Q][3], ph, acc[3][MT - 1]); RTD_SB();
#pragma unroll
        for (int j = 0; j < MT - 1; ++j) {
          RTD_MF(wh[P][0], xh[j & 1], acc[0][j]); xh[(j + 1) & 1] = *(const sp16x8*)(sa + (j + 1) * 2048 + foh); RTD_SB();
          RTD_MF(wh[P][1], xh[j & 1], acc[1][j]); RTD_MF(wh[P][2], xh[j & 1], acc[2][j]); RTD_MF(wh[P][3], xh[j & 1], acc[3][j]); RTD_SB();
          RTD_MF(wh[P][0], xl[j & 1], acc[0][j]); xl[(j + 1) & 1] = *(const sp16x8*)(sa + (j + 1) * 2048 + fol); RTD_SB();
          RTD_MF(wh[P][1], xl[j & 1], acc[1][j]); RTD_MF(wh[P][2], xl[j & 1], acc[2][j]); RTD_MF(wh[P][3], xl[j & 1], acc[3][j]); RTD_SB();
          RTD_MF(wl[P][0], xh[j & 1], acc[0][j]); RTD_MF(wl[P][1], xh[j & 1], acc[1][j]); RTD_MF(wl[P][2], xh[j & 1], acc[2][j]); RTD_MF(wl[P][3], xh[j & 1], acc[3][j]); RTD_SB();
        }
        // every LDS read of this stage has returned before the wave reaches the next barrier (after it the loaders refill the stage)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        ph = xh[(MT - 1) & 1]; pl = xl[(MT - 1) & 1];
      };
      int ks = 0;
      for (; ks + 1 < nk; ks += 2) { step(ActC<0>{}, ks); step(ActC<1>{}, ks + 1); }
      if (ks < nk) {
        step(ActC<0>{}, ks);
#pragma unroll
        for (int i = 0; i < 4; ++i) RTD_MF(wh[0][i], ph, acc[i][MT - 1]);
#pragma unroll
        for (int i = 0; i < 4; ++i) RTD_MF(wh[0][i], pl, acc[i][MT - 1]);
#pragma unroll
        for (int i = 0; i < 4; ++i) RTD_MF(wl[0][i], ph, acc[i][MT - 1]);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) RTD_MF(wh[1][i], ph, acc[i][MT - 1]);
#pragma unroll
        for (int i = 0; i < 4; ++i) RTD_MF(wh[1][i], pl, acc[i][MT - 1]);
#pragma unroll
        for (int i = 0; i < 4; ++i) RTD_MF(wl[1][i], ph, acc[i][MT - 1]);
      }
#undef RTD_SB
#undef RTD_MF
    };
    if (wm == 0) body(ActC<MTA>{}, 0);              // (ActC: a compile-time int)
    else body(ActC<MTB>{}, MTA * 16);
  }
  if (STAMP && stamps && wv == 0) stamps[1] = (long long)__builtin_amdgcn_s_memtime() - t_base;
  __syncthreads();

  auto for_each_group = [&](auto&& f) {
    const int r16 = lane & 15, c4 = lane >> 4;
    const int rb = wm ? MTA * 16 : 0, mtw = wm ? MTB : MTA;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < MTX; ++j)
        if (j < mtw) f(wn * 64 + 16 * i + 4 * c4, rb + 16 * j + r16, acc[i][j]);
  };
  if (a.y_split && a.res_mode == RES_NONE) {
    sp16* sb = (sp16*)smem;
    if (!loader) {
      dispatch_act(a.act, [&](auto actc) {
        constexpr int ACT = decltype(actc)::value;
        for_each_group([&](int cl, int pl, const f32x4& v) {
          const f32x4 bv = *(const f32x4*)(a.bias + n0 + cl);
          sp16x4 oh, ol;
#pragma unroll
          for (int e = 0; e < 4; ++e) { sp16 hi, lo; split2(act_c<ACT>(v[e] + bv[e]), hi, lo); oh[e] = hi; ol[e] = lo; }
          sp16* d = sb + pl * SLB + ((cl >> 5) << 6) + (cl & 31);
          *(sp16x4*)d = oh;
          *(sp16x4*)(d + SPLIT_GROUP) = ol;
        });
      });
    }
    __syncthreads();
    if (STAMP && stamps && wv == 0) stamps[2] = (long long)__builtin_amdgcn_s_memtime() - t_base;
    ws_copy_out_split_rows<BN, BM>(a, sb, SLB, tid, m0, n0);
    if (STAMP && stamps && wv == 0) {
      stamps[3] = (long long)__builtin_amdgcn_s_memtime() - t_base;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      stamps[4] = (long long)__builtin_amdgcn_s_memtime() - t_base;
      stamps[7] = (long long)__builtin_amdgcn_s_memrealtime();
    }
    return;
  }
  float* st = (float*)smem;
  if (!loader) for_each_group([&](int cl, int pl, const f32x4& v) { *(f32x4*)(&st[pl * SLD + cl]) = v; });
  __syncthreads();
  dispatch_act(a.act, [&](auto actc) { ws_copy_out_sp<CITERS, decltype(actc)::value, BN, BM>(a, st, SLD, tid, m0, n0, 0); });
}



int conv_kpad(int K) { return (K + 63) / 64 * 64; }
int conv_npad(int N) { return (N + 127) / 128 * 128; }

static ConvOpts g_conv_opts;
ConvOpts& conv_opts_template() { return g_conv_opts; }
bool conv_set_option(const char* name, int value) {
  ConvOpts& o = g_conv_opts;
  const struct { const char* n; int* p; } table[] = {
      {"conv_mode", &o.conv_mode}, {"glds_min_blocks", &o.glds_min_blocks}, {"glds_min_n", &o.glds_min_n}, 
      {"ws2_min_blocks", &o.ws2_min_blocks}, {"ws64_max_blocks", &o.ws64_max_blocks}, 
      {"reg_epilogue", &o.reg_epilogue}, {"conv_reg", &o.conv_reg}, 
      {"prefetch", &o.prefetch},
      {"glds_drop", &o.glds_drop}, {"split_ws2_min_blocks", &o.split_ws2_min_blocks}, {"split_ws64_max_blocks", &o.split_ws64_max_blocks},
      {"split_flex", &o.split_flex}, {"split_flex_min_nk", &o.split_flex_min_nk}, {"split_flex_small_max", &o.split_flex_small_max},
      {"split_sx", &o.split_sx}, {"split_k2", &o.split_k2}, {"split_wsq", &o.split_wsq}, {"split_wsq_min_blocks", &o.split_wsq_min_blocks},
      {"split_wsq_min_nk", &o.split_wsq_min_nk},
  };
  for (const auto& t : table)
    if (strcmp(name, t.n) == 0) { *t.p = value; return true; }
  return false;
}
static inline const ConvOpts& opts_of(const ConvArgs& a) { return a.opts ? *a.opts : g_conv_opts; }

// LDS-DMA tile kernels on bf16 / fp32 operands; returns true when the launch was taken
template <typename T>
static bool dispatch_glds(const ConvOpts& o, const ConvK& k, bool ok, bool prefer256, long long x_bytes, long long w_bytes, unsigned y_bytes, unsigned x2_bytes,
                          const ConvWorkspace& ws, hipStream_t s) {
  if (!ok || (o.conv_mode == 1 && !k.x2)) return false;
  static_assert(sizeof(T) == 2 || sizeof(T) == 4, "bf16 / fp32");
  const long long mt = (k.M + 127) / 128, ntn = (k.N + 127) / 128;
  // N >= 64 may use a partly empty N tile (the filter is padded to 128 rows): the N = 64 reduce convs of stage 0 are HBM-bound,
  // and the LDS-DMA pipeline (2 blocks per CU) streams their input faster than the register-staged kernel
  if (k.N < (sizeof(T) == 2 ? o.glds_min_n : 64) || mt * ntn < (sizeof(T) == 2 ? o.glds_min_blocks : 512) || x_bytes >= (1ll << 31) || w_bytes >= (1ll << 31)) return false;
  ConvG g;
  g.k = k;
  g.k.ntn = (int)ntn;
  g.probe = o.glds_drop;
  g.splitk = 1; g.slab = (o.glds_drop & 32) ? ws.slab : nullptr; g.y_bytes = 0;
  g.x_bytes = (o.glds_drop & 1) ? 0u : (unsigned)x_bytes;
  g.w_bytes = (o.glds_drop & 2) ? 0u : (unsigned)w_bytes;
  g.x2_bytes = x2_bytes;
  // small grids: 128 x 64 tiles double the blocks (latency profile only: with other batches in flight the idle CUs are taken anyway and
  // the narrower tile stages 1.5x the bytes per MFMA - measured +1.4 % for one handle, -1 % for three); conv_mode 10 forces it for the tests
  if ((o.conv_mode == 0 && !prefer256 && mt * ntn < o.ws64_max_blocks && k.N > 64) || o.conv_mode == 10) {
    const long long ntn64 = (k.N + 63) / 64;
    g.k.ntn = (int)ntn64;
    rtd_launch((conv_igemm_ws_kernel<T, 4, 64>), dim3((unsigned)(mt * ntn64)), dim3(512), 0, s, g);
    return true;
  }
  if (k.x2) {
    // dual-input launches exist in the wave-specialised kernel only (every conv_mode): 4 stages on small grids, 2 above
    if (mt * ntn < o.ws2_min_blocks) rtd_launch((conv_igemm_ws_kernel<T, 4>), dim3((unsigned)(mt * ntn)), dim3(512), 0, s, g);
    else rtd_launch((conv_igemm_ws_kernel<T, 2>), dim3((unsigned)(mt * ntn)), dim3(512), 0, s, g);
    return true;
  }
  // loader / MFMA wave roles win at every grid size (tools/profile_layers.py): grids beyond one block per CU run 2 blocks per CU with
  // 2 stages, smaller grids 1 block per CU with 4 stages (3 tiles of DMA in flight); conv_mode 3 / 4 force either
  const bool four = o.conv_mode == 3 || (o.conv_mode != 4 && mt * ntn < o.ws2_min_blocks);
  if (four) rtd_launch((conv_igemm_ws_kernel<T, 4>), dim3((unsigned)(mt * ntn)), dim3(512), 0, s, g);
  else rtd_launch((conv_igemm_ws_kernel<T, 2>), dim3((unsigned)(mt * ntn)), dim3(512), 0, s, g);
  return true;
}


// F16X2 form of the direct 3x3 kernel for 32 input channels (stem.1 32 -> 32, stem.2 32 -> 64 at 320^2): a pixel's 128 bytes are
// [32 hi | 32 lo], so the patch, its LDS-DMA and its swizzle are the sp16 kernel's CIN = 64 case; each wave keeps the hi AND the lo filter
// of its 32 output channels in registers (36 fragments = 144 VGPRs) and runs hi*hi + hi*lo + lo*hi per tap (6 MFMAs per tap and row).
// 8 waves per block, one block per CU (two 43 KiB patch buffers): COG channel groups x (8 / COG) waves, each wave 8 / (8 / COG) rows.
// The epilogue writes the row's F16X2 pixels ([32 hi | 32 lo] per 32-channel group) through the wave's slab, 16-byte stores.
// POOL (stem.2 -> 3x3 / stride-2 / pad-1 max-pool, HF:rt_detr_resnet.py:100-113): the conv rows are never written.  The waves of a
// channel group own the tile's rows in pairs (rows 2 w, 2 w + 1 = the pooled row w's lower two); a wave takes the row above from its
// neighbour through LDS, pools vertically in registers and horizontally through its slab, and writes the 4 x 16 pooled pixels.  A window
// that crosses the tile's top row or left column lacks the conv pixels of the tile above / to the left: every tile also leaves its last row
// (`side_row`) and its vertically pooled last column (`side_col`) in a side buffer and k_pool_fixup completes the first pooled row and
// column of every tile afterwards.  The conv outputs are >= 0 (ReLU) and max is exact: the result equals conv -> pool bit for bit.
struct PoolOut {
  sp16* y;                 // pooled output [B][OH][OW][C] as F16X2
  long long ldy, bstride;  // channels
  int OH, OW;
  unsigned y_bytes;
  sp16* side_row;          // [tile][32 px][2 groups][32 hi | 32 lo]
  sp16* side_col;          // [tile][4 rows][2 groups][32 hi | 32 lo]
  unsigned row_bytes, col_bytes;
};
template <int COG, bool POOL = false>   // output channel groups of 32 (1 or 2)
__global__ __launch_bounds__(512, 2) void conv3x3_reg_split_kernel(const ConvK a, unsigned x_bytes, unsigned y_bytes, int tiles_x, int tiles_y, int ntiles,
                                                                   const PoolOut po) {
  static_assert(!POOL || COG == 2, "the pooled form is stem.2's (32 -> 64 channels)");
  constexpr int NW = 8, NT_ = 512, WPG = NW / COG, RPW = 8 / WPG;    // waves per channel group, rows per wave
  constexpr int TH = 8, TW = 32, PW = TW + 2, PH = TH + 2, NPIX = PW * PH;
  constexpr int ROWB = 128, CPP = 8, PPI = 8;
  constexpr int NINSTR = (NPIX + PPI - 1) / PPI;
  constexpr int PBUF = NINSTR * 1024;
  constexpr int ROWO = 128 + 16;                  // slab row: one pixel's [32 hi | 32 lo] + 16 bytes (bank skew)
  constexpr int WROW = 9 * 64 * 2 + 16;           // filter row in LDS: 9 taps x [32 hi | 32 lo] + 16 bytes
  static_assert(32 * COG * WROW <= 2 * PBUF, "the filter is staged through the two patch buffers");
  __shared__ __attribute__((aligned(16))) char patch[2 * PBUF];
  __shared__ __attribute__((aligned(16))) char stage[NW][32 * ROWO];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wv / WPG, wq = wv % WPG;
  const int nbase = grp * 32;
  const int h = lane >> 5;

  sp16x8 wf[9][4];                                  // [tap][kc]: kc 0, 1 = hi channels 0-15 / 16-31, kc 2, 3 = lo
  {
    const sp16* wg = (const sp16*)a.w;
    constexpr int CPR = 9 * 64 / 8;
    for (int e = tid; e < 32 * COG * CPR; e += NT_) {
      const int row = e / CPR, ch = e - row * CPR;
      *(sp16x8*)(patch + row * WROW + ch * 16) = *(const sp16x8*)(wg + (size_t)row * a.Kpad + ch * 8);
    }
    __syncthreads();
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int kc = 0; kc < 4; ++kc)
        wf[tap][kc] = *(const sp16x8*)(patch + (nbase + (lane & 31)) * WROW + (tap * 64 + kc * 16 + 8 * h) * 2);
    __syncthreads();
  }

  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(POOL ? (void*)po.y : (void*)a.y, 0, POOL ? po.y_bytes : y_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsr = __builtin_amdgcn_make_buffer_rsrc((void*)(POOL ? po.side_row : (sp16*)a.y), 0, POOL ? po.row_bytes : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsc = __builtin_amdgcn_make_buffer_rsrc((void*)(POOL ? po.side_col : (sp16*)a.y), 0, POOL ? po.col_bytes : 0u, 0x00020000);
  auto swz = [](int pi) { return (pi >> 1) & 7; };
  auto issue_patch = [&](int tile, int buf) {
    const int tx = tile % tiles_x;
    const int t2 = tile / tiles_x;
    const int ty = t2 % tiles_y;
    const int b = t2 / tiles_y;
    const int x0 = tx * TW, y0 = ty * TH;
    for (int j = wv; j < NINSTR; j += NW) {
      const int pi = j * PPI + lane / CPP;
      const int py = pi / PW, px = pi - py * PW;
      const int iy = y0 - 1 + py, ix = x0 - 1 + px;
      const bool ok = pi < NPIX && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      const int src_chunk = (lane % CPP) ^ swz(pi);
      const unsigned vo = ok ? (unsigned)(((long long)b * a.x_bstride + ((long long)iy * a.W + ix) * a.ldx) * 2 + src_chunk * 16) : 0x80000000u;   // ldx: sp16 elements
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(patch + buf * PBUF + j * 1024), 16, vo, 0, 0, 0);
    }
  };

  int tile = blockIdx.x;
  if (tile < ntiles) issue_patch(tile, 0);
  for (int it = 0; tile < ntiles; tile += gridDim.x, ++it) {
    const int buf = it & 1;
    // the 4 RPW stores of the previous tile were issued after this tile's DMA and may stay in flight (every row issues exactly 4)
    if (it == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (RPW == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) issue_patch(tile + gridDim.x, buf ^ 1);
    const int tx = tile % tiles_x;
    const int t2 = tile / tiles_x;
    const int ty = t2 % tiles_y;
    const int b = t2 / tiles_y;
    const int x0 = tx * TW, y0 = ty * TH;
    const char* pbuf = patch + buf * PBUF;

    float vm[16];                                   // POOL: max over the wave's two rows, per (pixel = lane & 31, channel 8 q + 4 h + e)
#pragma unroll
    for (int e = 0; e < 16; ++e) vm[e] = 0.f;
#pragma unroll 1
    for (int rr_ = 0; rr_ < RPW; ++rr_) {
      const int r = wq * RPW + rr_;
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
      sp16x8 xf[2][4];
      auto read_tap = [&](sp16x8 (&dst)[4], int tap) {
        const int kh = tap / 3, kw = tap - kh * 3;
        const int pi = (r + kh) * PW + kw + (lane & 31);
        const int sw = swz(pi);
        const char* prow = pbuf + pi * ROWB;
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) dst[kc] = *(const sp16x8*)(prow + (((2 * kc + h) ^ sw) << 4));
      };
      read_tap(xf[0], 0);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        if (tap + 1 < 9) read_tap(xf[(tap + 1) & 1], tap + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          acc = mfma_pair32(wf[tap][c], xf[tap & 1][c], acc);
          acc = mfma_pair32(wf[tap][c], xf[tap & 1][c + 2], acc);
          acc = mfma_pair32(wf[tap][c + 2], xf[tap & 1][c], acc);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // ---- epilogue of this 32-pixel row: lane = pixel (lane & 31), channels 8 q + 4 h + (0..3) of the wave's group ----
      const int oy = y0 + r;
      char* sw_ = stage[wv];
      const bool pix_ok = !POOL || (oy < a.H && x0 + (lane & 31) < a.W);       // POOL: pixels outside the map count as 0 (every window holds a real one)
      dispatch_act(a.act, [&](auto actc) {
        constexpr int ACT = decltype(actc)::value;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 bv = *(const f32x4*)(a.bias + nbase + 8 * q + 4 * h);
          sp16x4 oh, ol;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            sp16 hi, lo;
            split2(pix_ok ? act_c<ACT>(acc[4 * q + e] + bv[e]) : 0.f, hi, lo);
            oh[e] = hi; ol[e] = lo;
            if (POOL) vm[4 * q + e] = fmaxf(vm[4 * q + e], (float)hi + (float)lo);      // the represented value
          }
          *(sp16x4*)(sw_ + (lane & 31) * ROWO + (8 * q + 4 * h) * 2) = oh;
          *(sp16x4*)(sw_ + (lane & 31) * ROWO + 64 + (8 * q + 4 * h) * 2) = ol;
        }
      });
      __builtin_amdgcn_wave_barrier();
      if constexpr (POOL) {
        // only the tile's LAST row leaves the block as it is: the side buffer of the tile below (4 stores per wave, out of range for the others)
        if (rr_ == RPW - 1) {
          const long long rrow = (long long)tile * 8192 + grp * 128;
#pragma unroll
          for (int i2 = 0; i2 < 4; ++i2) {
            const int idx = i2 * 64 + lane;
            const int p = idx >> 3, ch = idx & 7;
            const unsigned vo = (wq == WPG - 1) ? (unsigned)(rrow + p * 256 + ch * 16) : 0x80000000u;
            __builtin_amdgcn_raw_buffer_store_b128(*(const u32x4_*)(sw_ + p * ROWO + ch * 16), rsr, vo, 0, 0);
          }
        }
      } else {
        const long long yrow = 4 * ((long long)b * a.y_bstride + ((long long)oy * a.W + x0) * a.ldy + nbase);   // bytes; ldy: channels
#pragma unroll
        for (int i2 = 0; i2 < 4; ++i2) {
          const int idx = i2 * 64 + lane;
          const int p = idx >> 3, ch = idx & 7;
          const bool ok = oy < a.H && x0 + p < a.W;
          const unsigned vo = ok ? (unsigned)(yrow + 4 * (long long)p * a.ldy + ch * 16) : 0x80000000u;
          __builtin_amdgcn_raw_buffer_store_b128(*(const u32x4_*)(sw_ + p * ROWO + ch * 16), ry, vo, 0, 0);
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    if constexpr (POOL) {
      char* sw_ = stage[wv];
      __syncthreads();                               // every wave's second row sits in its slab as hi / lo
      if (wq > 0) {                                  // the row above this wave's pair: the neighbour's second row (wave 0: the tile above, k_pool_fixup)
        const char* nb = stage[wv - 1];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const sp16x4 nh = *(const sp16x4*)(nb + (lane & 31) * ROWO + (8 * q + 4 * h) * 2);
          const sp16x4 nl = *(const sp16x4*)(nb + (lane & 31) * ROWO + 64 + (8 * q + 4 * h) * 2);
#pragma unroll
          for (int e = 0; e < 4; ++e) vm[4 * q + e] = fmaxf(vm[4 * q + e], (float)nh[e] + (float)nl[e]);
        }
      }
      __syncthreads();                               // the slabs may be overwritten
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *(f32x4*)(sw_ + (lane & 31) * ROWO + (8 * q + 4 * h) * 4) = f32x4{vm[4 * q], vm[4 * q + 1], vm[4 * q + 2], vm[4 * q + 3]};
      __builtin_amdgcn_wave_barrier();
      // horizontal: pooled pixel k = lane >> 2 of this wave's pooled row, channels 8 (lane & 3) .. + 7 of the group, from pixels 2 k - 1 .. 2 k + 1
      const int k = lane >> 2, cq = lane & 3;
      float pm[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) pm[e] = 0.f;
#pragma unroll
      for (int d = -1; d <= 1; ++d) {
        const int px = 2 * k + d;
        if (px >= 0) {                               // (px <= 31 always; pixel -1 belongs to the tile on the left: k_pool_fixup)
          const f32x4 t0 = *(const f32x4*)(sw_ + px * ROWO + cq * 32), t1 = *(const f32x4*)(sw_ + px * ROWO + cq * 32 + 16);
#pragma unroll
          for (int e = 0; e < 4; ++e) { pm[e] = fmaxf(pm[e], t0[e]); pm[4 + e] = fmaxf(pm[4 + e], t1[e]); }
        }
      }
      sp16x8 ph, pl;
#pragma unroll
      for (int e = 0; e < 8; ++e) { sp16 hi, lo; split2(pm[e], hi, lo); ph[e] = hi; pl[e] = lo; }
      const int py = (y0 >> 1) + wq, pxo = (x0 >> 1) + k;
      {
        const bool ok = py < po.OH && pxo < po.OW;
        const long long pb = 4 * ((long long)b * po.bstride + ((long long)py * po.OW + pxo) * po.ldy + nbase) + cq * 16;   // bytes
        const unsigned vo = ok ? (unsigned)pb : 0x80000000u;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, ph), ry, vo, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, pl), ry, vo == 0x80000000u ? vo : vo + 64u, 0, 0);
      }
      {
        // the tile's last column, vertically pooled over this wave's rows: what the tile on the right lacks for its first pooled column
        float cm[8];
        const f32x4 t0 = *(const f32x4*)(sw_ + 31 * ROWO + cq * 32), t1 = *(const f32x4*)(sw_ + 31 * ROWO + cq * 32 + 16);
#pragma unroll
        for (int e = 0; e < 4; ++e) { cm[e] = t0[e]; cm[4 + e] = t1[e]; }
        sp16x8 chh, cll;
#pragma unroll
        for (int e = 0; e < 8; ++e) { sp16 hi, lo; split2(cm[e], hi, lo); chh[e] = hi; cll[e] = lo; }
        const unsigned vo = (k == 15) ? (unsigned)((long long)tile * 1024 + wq * 256 + grp * 128 + cq * 16) : 0x80000000u;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, chh), rsc, vo, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, cll), rsc, vo == 0x80000000u ? vo : vo + 64u, 0, 0);
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// completes the first pooled row and the first pooled column of every tile of the fused stem.2 -> max-pool launch (see PoolOut): one
// thread per (tile, 16 + 3 boundary pixels, 8 channels)
__global__ __launch_bounds__(256) void k_pool_fixup(const PoolOut po, int tiles_x, int tiles_y, int ntiles, int C) {
  const int c8n = C / 8;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)ntiles * 19 * c8n) return;
  const int cc = (int)(i % c8n) * 8;
  const long long t = i / c8n;
  const int item = (int)(t % 19), tile = (int)(t / 19);
  const int tx = tile % tiles_x, t2 = tile / tiles_x, ty = t2 % tiles_y, b = t2 / tiles_y;
  const int j = item < 16 ? 0 : item - 15, k = item < 16 ? item : 0;
  const int py = 4 * ty + j, px = 16 * tx + k;
  if (py >= po.OH || px >= po.OW) return;
  const bool up = j == 0 && ty > 0, left = k == 0 && tx > 0;
  if (!up && !left) return;
  const long long pix = (long long)b * po.bstride + ((long long)py * po.OW + px) * po.ldy;
  float v[8];
  split_load8(po.y, pix, cc, v);
  const int grp = cc >> 5, cg = cc & 31;
  auto row_at = [&](int tl, int p) {      // side_row[tl][p][grp][hi | lo]
    const sp16* q = po.side_row + (size_t)tl * 4096 + p * 128 + grp * 64 + cg;
    const sp16x8 hh = *(const sp16x8*)q, ll = *(const sp16x8*)(q + 32);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], (float)hh[e] + (float)ll[e]);
  };
  if (up) {
    const int above = tile - tiles_x;
#pragma unroll
    for (int d = -1; d <= 1; ++d) {
      const int p = 2 * k + d;
      if (p >= 0) row_at(above, p);
    }
    if (left) row_at(above - 1, 31);
  }
  if (left) {
    const sp16* q = po.side_col + (size_t)(tile - 1) * 512 + j * 128 + grp * 64 + cg;
    const sp16x8 hh = *(const sp16x8*)q, ll = *(const sp16x8*)(q + 32);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], (float)hh[e] + (float)ll[e]);
  }
  split_store8(po.y, pix, cc, v);
}


// The same direct scheme for the 64 -> 64 channel 3x3 convs of stage 0 (bottleneck c2 at 160^2: 15 GFLOP on 105 MB, 3.5x off both rooflines on the
// 128 x 64 implicit-GEMM tile, whose nine taps re-stage every input pixel and whose LDS traffic per MFMA is the highest of all tiles).
//   * tile 8 x 16 output pixels; the (8+2) x (16+2) x 256-byte input patch is staged once by LDS-DMA (double buffered, source-side XOR swizzle:
//     16-byte slot c of pixel pi holds chunk c ^ (pi & 15), so the 16 consecutive pixels of a B-fragment read hit 16 different bank groups);
//   * wave = (16-channel group, row half): its 16 x 576 filter slice, hi and lo, lives in 144 VGPRs as v_mfma_f32_16x16x32_f16 A operands;
//   * a patch-row fragment (one ds_read_b128 pair: hi, lo) serves up to three output rows (kh = patch row - output row): 72 fragment reads for
//     216 MFMAs per wave and tile - the loop is MFMA-bound, not LDS-bound;
//   * the tile's 128 x 64 outputs are collected in LDS as F16X2 rows and stored row-shaped (16 lanes = one pixel's 256-byte run).
// Per output: taps in (kh, kw) order, channel groups inside a tap, lo-terms first - fixed per pixel, so batch size does not change a frame's bits.
__global__ __launch_bounds__(512, 1) void conv3x3_reg_split64_kernel(const ConvK a, unsigned x_bytes, unsigned y_bytes, int tiles_x, int tiles_y, int ntiles) {
  constexpr int NW = 8;
  constexpr int TH = 8, TW = 16, PW = TW + 2, PH = TH + 2, NPIX = PW * PH;
  constexpr int PPI = 4;                          // pixels per DMA instruction (64 lanes x 16 bytes, 256 bytes per pixel)
  constexpr int NINSTR = (NPIX + PPI - 1) / PPI;
  constexpr int PBUF = NINSTR * 1024;
  constexpr int SROWB = 256 + 16;
  __shared__ __attribute__((aligned(16))) char patch[2 * PBUF];
  __shared__ __attribute__((aligned(16))) char stage[TH * TW * SROWB];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cgp = wv & 3, rh = wv >> 2;           // 16-channel group, row half (output rows 4 rh .. 4 rh + 3)
  const int r16 = lane & 15, kq = lane >> 4;

  sp16x8 wf[9][2][2];                             // [tap][32-channel group][hi / lo]
  {
    const sp16* wr = (const sp16*)a.w + (size_t)(16 * cgp + r16) * a.Kpad + 8 * kq;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        wf[tap][c][0] = *(const sp16x8*)(wr + (tap * 2 + c) * 64);
        wf[tap][c][1] = *(const sp16x8*)(wr + (tap * 2 + c) * 64 + 32);
      }
  }
  const f32x4 bv = *(const f32x4*)(a.bias + 16 * cgp + 4 * kq);

  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, y_bytes, 0x00020000);
  auto issue_patch = [&](int tile, int buf) {
    const int tx = tile % tiles_x;
    const int t2 = tile / tiles_x;
    const int ty = t2 % tiles_y;
    const int b = t2 / tiles_y;
    const int x0 = tx * TW, y0 = ty * TH;
    for (int j = wv; j < NINSTR; j += NW) {
      const int pi = j * PPI + (lane >> 4);
      const int py = pi / PW, px = pi - py * PW;
      const int iy = y0 - 1 + py, ix = x0 - 1 + px;
      const bool ok = pi < NPIX && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      const int src_chunk = (lane & 15) ^ (pi & 15);
      const unsigned vo = ok ? (unsigned)(((long long)b * a.x_bstride + ((long long)iy * a.W + ix) * a.ldx) * 2 + src_chunk * 16) : 0x80000000u;   // ldx: sp16 elements
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(patch + buf * PBUF + j * 1024), 16, vo, 0, 0, 0);
    }
  };

  int tile = blockIdx.x;
  if (tile < ntiles) issue_patch(tile, 0);
  for (int it = 0; tile < ntiles; tile += gridDim.x, ++it) {
    const int buf = it & 1;
    // the 4 stores of the previous tile were issued after this tile's DMA and may stay in flight
    if (it == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __syncthreads();                                // patch[buf] landed for every wave; everyone has left the previous tile (patch[buf ^ 1], stage)
    if (tile + (int)gridDim.x < ntiles) issue_patch(tile + gridDim.x, buf ^ 1);
    const int tx = tile % tiles_x;
    const int t2 = tile / tiles_x;
    const int ty = t2 % tiles_y;
    const int b = t2 / tiles_y;
    const int x0 = tx * TW, y0 = ty * TH;
    const char* pbuf = patch + buf * PBUF;

    f32x4 acc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
    sp16x8 xf[2][2][2];                             // [parity][group][hi / lo]
    // fragment addresses are tile-invariant; recomputed per tile behind an opaque copy (hoisted out of the tile loop they cost 70 VGPRs and spill)
    int q0 = 4 * rh * PW + r16;
    asm volatile("" : "+v"(q0));
    auto read_px = [&](sp16x8 (&dst)[2][2], int pr, int kw) {
      const int pi = q0 + pr * PW + kw;
      const unsigned a0 = (unsigned)(pi << 8) | (unsigned)(((pi & 15) ^ kq) << 4);     // chunk kq of group 0, hi; the others differ in slot bits 2, 3
      dst[0][0] = *(const sp16x8*)(pbuf + a0);
      dst[0][1] = *(const sp16x8*)(pbuf + (a0 ^ 0x40));
      dst[1][0] = *(const sp16x8*)(pbuf + (a0 ^ 0x80));
      dst[1][1] = *(const sp16x8*)(pbuf + (a0 ^ 0xC0));
    };
    read_px(xf[0], 0, 0);
#pragma unroll
    for (int s = 0; s < 18; ++s) {                  // s = patch row * 3 + kw
      const int pr = s / 3, kw = s - pr * 3;
      if (s + 1 < 18) read_px(xf[(s + 1) & 1], (s + 1) / 3, (s + 1) % 3);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kh = pr - r;
        if (kh >= 0 && kh <= 2) {
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            acc[r] = mfma_pair16(wf[kh * 3 + kw][c][0], xf[s & 1][c][1], acc[r]);
            acc[r] = mfma_pair16(wf[kh * 3 + kw][c][1], xf[s & 1][c][0], acc[r]);
            acc[r] = mfma_pair16(wf[kh * 3 + kw][c][0], xf[s & 1][c][0], acc[r]);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- epilogue: lane = pixel r16 of each of the wave's 4 rows, channels 16 cgp + 4 kq + (0..3) -> F16X2 rows in LDS ----
    dispatch_act(a.act, [&](auto actc) {
      constexpr int ACT = decltype(actc)::value;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        sp16x4 oh, ol;
#pragma unroll
        for (int e = 0; e < 4; ++e) { sp16 hi, lo; split2(act_c<ACT>(acc[r][e] + bv[e]), hi, lo); oh[e] = hi; ol[e] = lo; }
        char* q = stage + ((4 * rh + r) * TW + r16) * SROWB + (cgp >> 1) * 128 + (16 * (cgp & 1) + 4 * kq) * 2;
        *(sp16x4*)q = oh;
        *(sp16x4*)(q + 64) = ol;
      }
    });
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + 512 * i;
      const int p = idx >> 4, ck = idx & 15;
      const int oy = y0 + (p >> 4), ox = x0 + (p & 15);
      const bool ok = oy < a.H && ox < a.W;
      const unsigned vo = ok ? (unsigned)(4 * ((long long)b * a.y_bstride + ((long long)oy * a.W + ox) * a.ldy) + ck * 16) : 0x80000000u;   // ldy: channels
      __builtin_amdgcn_raw_buffer_store_b128(*(const u32x4_*)(stage + p * SROWB + ck * 16), ry, vo, 0, 0);
    }
  }
}



// register-staged fallback: every shape (small channel counts, grids below the LDS-DMA kernels' thresholds)
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
    if (smallc) rtd_launch((conv_igemm_kernel<T, BM, BN, true>), dim3((unsigned)blocks), dim3(256), 0, s, kk); \
    else rtd_launch((conv_igemm_kernel<T, BM, BN, false>), dim3((unsigned)blocks), dim3(256), 0, s, kk);       \
  } while (0)
  if (cfg == 0) RTD_LAUNCH(128, 128);
  else if (cfg == 1) RTD_LAUNCH(128, 64);
  else RTD_LAUNCH(64, 64);
#undef RTD_LAUNCH
}

// The second input exists in the wave-specialised LDS-DMA kernel (and, for 64 + 64 channels, the streaming kernel): the shapes
// dispatch_glds accepts, a K-step-aligned split point and whole K-steps of x2.
bool conv_dual_supported(const ConvArgs& a) {
  const ConvOpts& o = opts_of(a);
  const Tensor& x = a.x;
  const Tensor& y = a.y;
  const Tensor& x2 = a.x2;
  if (!x2.p || x2.dt != x.dt) return false;
  const int up = a.x_up2 ? 2 : 1;
  if (a.x_up2 && !(a.KH == 1 && a.KW == 1 && a.stride == 1 && a.pad == 0)) return false;
  const int OH = (x.h * up + 2 * a.pad - a.KH) / a.stride + 1, OW = (x.w * up + 2 * a.pad - a.KW) / a.stride + 1;
  if (x2.n != x.n || x2.h != OH || x2.w != OW) return false;
  if (x.dt == F16X2) {                                           // the split kernel takes every grid size
    const long long xb = ((long long)(x.n - 1) * x.bstride + ((long long)x.h * x.w - 1) * x.ld + x.c) * 4;
    const long long x2b = ((long long)(x2.n - 1) * x2.bstride + ((long long)x2.h * x2.w - 1) * x2.ld + x2.c) * 4;
    return conv_split_supported(a) && xb < (1ll << 31) && x2b < (1ll << 31);
  }
  const int es = (int)dtype_size(x.dt), bk = 128 / es, epc = 16 / es;
  if ((a.KH * a.KW * x.c) % bk || x2.c % bk || x.c % bk || x2.ld % epc || ((uintptr_t)x2.p & 15)) return false;
  if (!(y.c % 8 == 0 && y.ld % 8 == 0 && ((uintptr_t)y.p & 15) == 0)) return false;                       // tile_ok
  if (a.res_mode != RES_NONE && !(a.res.ld % 8 == 0 && ((uintptr_t)a.res.p & 15) == 0)) return false;
  const long long M = (long long)x.n * OH * OW;
  const long long mt = (M + 127) / 128, ntn = (y.c + 127) / 128;
  if (y.c < (es == 2 ? o.glds_min_n : 64) || mt * ntn < (es == 2 ? o.glds_min_blocks : 512)) return false;
  const long long x_bytes = ((long long)(x.n - 1) * x.bstride + ((long long)x.h * x.w - 1) * x.ld + x.c) * es;
  const long long x2_bytes = ((long long)(x2.n - 1) * x2.bstride + ((long long)x2.h * x2.w - 1) * x2.ld + x2.c) * es;
  const long long w_bytes = (long long)conv_npad(y.c) * conv_kpad(a.KH * a.KW * x.c + x2.c) * es;
  return x_bytes < (1ll << 31) && x2_bytes < (1ll << 31) && w_bytes < (1ll << 31);
}

// The fused following conv exists in the streaming kernel for N = 256 (K = 64, or 64 + 64 with a second input) -> 64 channels and
// N = 512 (K = 128) -> 128 channels.  Asked per plan (batch size): the fused and the separate form use the same filter tensors and
// give bit-identical outputs, so plans of different batch sizes may differ.
static bool sx_shape_ok(const ConvArgs& a);
// The streaming kernel addresses every tensor through a 2 GiB buffer descriptor and counts tiles in 30 bits: limits that depend on the
// BATCH, unlike every shape test (which looks at one image).  The plan builder asks this with the plan's real batch before it relies on a
// fusion that only the streaming kernel implements (ConvArgs::avg_y); dispatch_sx applies the same limits.
bool conv_sx_batch_fits(const ConvArgs& a) {
  auto span = [](const Tensor& t) { return ((long long)(t.n - 1) * t.bstride + ((long long)t.h * t.w - 1) * t.ld + t.c) * 4; };
  const Tensor& y = a.y;
  if (span(y) >= (1ll << 31)) return false;
  if (a.res_mode != RES_NONE && span(a.res) >= (1ll << 31)) return false;
  if (a.next_y.p && span(a.next_y) >= (1ll << 31)) return false;
  if (a.avg_y.p && span(a.avg_y) >= (1ll << 31)) return false;
  const long long ntiles = a.avg_y.p ? (long long)y.n * (y.h / 2) * (y.w / 16) : ((long long)y.n * y.h * y.w + 31) / 32;
  return ntiles < (1ll << 30);
}
bool conv_avg_supported(const ConvArgs& a) {
  const Tensor& x = a.x;
  const Tensor& y = a.y;
  if (x.dt != F16X2 || y.dt != F16X2 || !sx_shape_ok(a)) return false;
  if (a.x2.p || a.res_mode == RES_NONE || (y.h & 1) || (y.w & 15) || y.ld % SPLIT_GROUP) return false;
  if (x.c == 64) return y.c == 256 && a.next_y.p && a.next_y.c == 128 && conv_next_supported(a);
  return x.c == 128 && !a.next_y.p;
}
bool conv_next_supported(const ConvArgs& a) {
  const Tensor& x = a.x;
  const Tensor& y = a.y;
  if (x.dt == F16X2)
    return sx_shape_ok(a) && y.c == 256 && (a.next_y.c == 64 || a.next_y.c == 128) && a.next_y.dt == F16X2 && a.next_y.ld % SPLIT_GROUP == 0 && y.ld % SPLIT_GROUP == 0;
  return false;                                                 // bf16 / fp32 operands: no kernel with a fused following conv (round 5: the bf16 streaming kernels were removed)
}

// ------------------------------------------------------------------------------------------------
// Pair operands (F16X2 tensors, common.h).  Shapes: Cin (and C2) multiples of 32 - one channel group per K-step; output F16X2 (N % 32 == 0)
// or fp32; residual F16X2 or fp32.  Measured inside the network (R50 bs 8, random frames, same box): flexible tile heights on EVERY grid lose to
// fixed 128 x 128 / 128 x 64 tiles (5578 vs 5529 us of kernels per step; on zero-filled microbenchmarks they win 10-15 % on the 80^2 maps -
// at the clocks random data allows and with cold operands they do not), a persistent three-role kernel lost more (5740 us: its store waves
// compete with the MFMA waves for the SIMDs' issue slots) - both were removed in round 3; flexible heights stay on the small grids where
// they fill idle CUs (ConvOpts::split_flex).
int conv_kpad_split(int K) { return 2 * ((K + SPLIT_GROUP - 1) / SPLIT_GROUP * SPLIT_GROUP); }   // 16-bit elements of a filter row

// ------------------------------------------------------------------------------------------------
// Streaming 1x1 convolution on F16X2 operands for the thin, very wide-grid expand convs of the first backbone stages
// (stage-0 c3: 64 [+ 64 shortcut] -> 256 channels at 160^2, stage-1 c3: 128 -> 512 at 80^2): 0.3-0.5 GB per launch and 7-13 GFLOP, i.e.
// HBM-bound, and the tiled kernels pay a prologue, an LDS staging round trip and a barrier-separated epilogue per 128-pixel
// block for ONE or two K-steps.  Same scheme as conv1x1_stream_kernel (no LDS tile, no role split), on hi/lo pairs:
//   * a WAVE owns one 32-channel group of the output ( = one [32 hi | 32 lo] 128-byte run per pixel) and keeps that group's
//     filter, hi and lo, in registers (4 x K/16 fragments); the block's 8 waves cover 256 channels of a 32-pixel tile, wider
//     layers run several 256-channel blocks per tile on one XCD; persistent grid, tiles cyclic per tile stream;
//   * pixel fragments come straight from global memory in MFMA B-operand shape (lane = pixel, 16 bytes of K; the 8 waves' copies
//     hit in L1); three v_mfma_f32_32x32x16_f16 per 16-deep chunk (w_hi x_lo, w_lo x_hi, w_hi x_hi);
//   * filter rows are permuted at load time (MFMA row 8b+4h+r <- channel 16h+4b+r): a lane's 16 accumulators are 16 CONSECUTIVE
//     channels of its pixel; residual and output cross a wave-private slab (32 rows x 128 bytes + skew) so that their global
//     accesses are row-shaped (8 lanes x 16 bytes = one pixel's 128-byte run);
//   * NEXTN (N == 256): the following 256 -> NEXTN reduce conv (the next block's c1) runs on the tile while the eight slabs
//     hold y as hi/lo sp16 = exactly what that conv would read back from HBM: wave w takes 16 output channels (NEXTN = 64: of one
//     16-pixel half) with 8 x 3 v_mfma_f32_16x16x32_f16, its 16 x 256 filter slice (hi and lo) in registers, between two block
//     barriers.  The c1 launch and its read of the 4-bytes-per-channel y tensor (R50 bs 8 stage 0: 210 MB) disappear.
// Arithmetic per output: K chunks in order into a zero accumulator (lo-terms first inside a chunk), + bias, + residual (hi + lo),
// activation, one hi/lo rounding.  Kernel choice depends on the per-IMAGE extents only, so every batch size runs the same arithmetic.
// ------------------------------------------------------------------------------------------------
// AVG (a stage's last expand conv): the pixel tile is a 2-row x 16-column PATCH instead of 32 consecutive pixels, and while a wave's slab
// holds its 32 channels of the tile it also writes the 2 x 2 / stride-2 average of the patch (8 pixels) to AvgOut: the next stage's
// vd-shortcut input (HF:rt_detr_resnet.py:199-205, AvgPool2d(2, 2)) - that launch and its read of the whole stage output disappear.  The
// average is ((a00 + a01) + (a10 + a11)) * 0.25 on the represented values hi + lo, re-split: k_avgpool2_split's arithmetic bit for bit.
struct AvgOut {
  sp16* y;                 // [B][OH / 2][OW / 2][C] as F16X2
  long long ldy, bstride;  // channels
  unsigned y_bytes;
};
template <int NGX, int NG2, bool RES, int NEXTN, bool F32OUT = false, bool AVG = false>   // NGX / NG2: 32-channel K groups read from x / from ConvK::x2; F32OUT: fp32 rows out
__global__ __launch_bounds__(512, 2) void conv1x1_sx_kernel(const ConvK a, unsigned x_bytes, unsigned r_bytes, unsigned y_bytes, unsigned x2_bytes, int ntiles,
                                                            unsigned yn_bytes, int ny, const AvgOut ao) {
  static_assert(!AVG || !F32OUT, "the averaged copy is a F16X2 tensor");
  constexpr int NKX = 2 * NGX, NK2 = 2 * NG2, NKK = NKX + NK2;   // 16-deep MFMA chunks
  constexpr int NG_ = NGX + NG2;
  constexpr int SROW = 144;
  __shared__ __attribute__((aligned(16))) char slabs[8][32 * SROW];
  constexpr int XBUFS = NG_ <= 4 ? 2 : 1;                        // K = 256: one buffer (two barriers per tile) keeps two blocks per CU
  static_assert(!F32OUT || (!RES && !NEXTN), "fp32 output: plain conv only");
  __shared__ __attribute__((aligned(16))) char xs[XBUFS][32 * (NG_ * 128 + 16)];
  constexpr int YROW = NEXTN * 4 + 16;
  __shared__ __attribute__((aligned(16))) char y1s[NEXTN ? 32 * YROW : 16];   // the follower's output tile: accumulator-shaped writes, row-shaped stores
  __shared__ __attribute__((aligned(16))) float sbias[256];
  __shared__ __attribute__((aligned(16))) char pf_dummy[256];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, pl = lane & 31;
  // 1-D grid: block -> (XCD = id & 7, slot = id >> 3) -> (tile stream = slot / ny, 256-channel block cy = slot % ny).  The ny channel blocks of a
  // tile stream sit on ONE XCD and walk the same tiles at the same time, so a pixel tile comes from HBM once and from that XCD's L2 ny - 1
  // times (with the channel blocks in the grid's y dimension the value projection fetched every tile six times: 426 MB instead of 69 - HBM-bound)
  const int bid = blockIdx.x, slot = bid >> 3;
  const int cy = slot % ny, t0 = (slot / ny) * 8 + (bid & 7), tstride = ((int)gridDim.x >> 3) / ny * 8;
  const int cb = (cy * 8 + wv) * 32;                             // this wave's output channel group
  if (tid < 256) sbias[tid] = a.bias[cy * 256 + tid];

  sp16x8 wfh[NKK], wfl[NKK];
  {
    const int ch = cb + 16 * ((pl >> 2) & 1) + 4 * (pl >> 3) + (pl & 3);
    const sp16* wr = (const sp16*)a.w + (size_t)ch * a.Kpad + 8 * h;
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) {
      wfh[kk] = *(const sp16x8*)(wr + (kk >> 1) * 64 + 16 * (kk & 1));
      wfl[kk] = *(const sp16x8*)(wr + (kk >> 1) * 64 + 16 * (kk & 1) + 32);
    }
  }
  // NEXTN: this wave's 16 x 256 slice of the following filter as 16x16x32 A operands (row lane & 15, k 32 s + 8 (lane >> 4) ..)
  constexpr int NS = NEXTN ? 8 : 1;
  sp16x8 w1h[NS], w1l[NS];
  f32x4 b1v = {0.f, 0.f, 0.f, 0.f};
  const int slice = NEXTN == 64 ? (wv & 3) : wv;
  if (NEXTN) {
    const sp16* w1 = (const sp16*)a.next_w + (size_t)(16 * slice + (lane & 15)) * a.next_kpad + 8 * (lane >> 4);
#pragma unroll
    for (int s = 0; s < NS; ++s) { w1h[s] = *(const sp16x8*)(w1 + 64 * s); w1l[s] = *(const sp16x8*)(w1 + 64 * s + 32); }
    b1v = *(const f32x4*)(a.next_bias + 16 * slice + 4 * (lane >> 4));
  }
  __syncthreads();
  prefetch_share(a, bid, gridDim.x, tid, 512, pf_dummy);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rx2 = __builtin_amdgcn_make_buffer_rsrc((void*)(NG2 ? a.x2 : a.x), 0, NG2 ? x2_bytes : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)(RES ? a.res : a.x), 0, RES ? r_bytes : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, y_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ryn = __builtin_amdgcn_make_buffer_rsrc((void*)(NEXTN ? a.next_y : a.y), 0, NEXTN ? yn_bytes : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rya = __builtin_amdgcn_make_buffer_rsrc((void*)(AVG ? (void*)ao.y : a.y), 0, AVG ? ao.y_bytes : 0u, 0x00020000);
  // tile t -> (image, first pixel); pixel px of the tile -> pixel index inside the image.  AVG: 2 x 16 patches, (OH / 2) x (OW / 16) per image
  const int tpr = AVG ? a.OW >> 4 : 1, tpi = AVG ? (a.OH >> 1) * tpr : 1;
  auto tile_origin = [&](int t, int& b0, int& p0) {
    if (AVG) { b0 = t / tpi; const int r = t - b0 * tpi, ty = r / tpr; p0 = 2 * ty * a.OW + 16 * (r - ty * tpr); }
    else { const int m0 = t * 32; b0 = m0 / a.OHW; p0 = m0 - b0 * a.OHW; }
  };
  auto tile_pixel = [&](int t, int b0, int p0, int px, int& b, int& p) -> bool {      // false: the pixel does not exist
    b = b0;
    if (AVG) { p = p0 + (px >> 4) * a.OW + (px & 15); return true; }
    p = p0 + px;
    if (p >= a.OHW) { p -= a.OHW; ++b; }
    return t * 32 + px < a.M;
  };
  const float* bl = sbias + wv * 32 + 16 * h;
  char* sl = slabs[wv];
  char* sl_acc = sl + pl * SROW + 32 * h;                         // this lane's 16 channels in accumulator shape: hi here, lo at + 64
  char* sl_row = sl + (lane >> 3) * SROW + (lane & 7) * 16;       // ... in row shape (pixel lane / 8 + 8 j, 16-byte chunk lane % 8)

  // The pixel tile (32 rows x NG x 128 bytes) is staged ONCE per block through LDS, double buffered: thread -> (pixel tid / CPR, 16-byte chunk
  // tid % CPR) so that a pixel's run is read by consecutive lanes (fragment-shaped global reads - 64 lanes on 64 different lines, issued by
  // all 8 waves - cost 4x the address cycles: 76 us instead of 51 for the stage-1 layers).  The loads of tile i + 1 (and its residual rows)
  // are in flight while tile i is computed; one block barrier per tile.
  constexpr int XROW = NG_ * 128 + 16;                            // + 16: consecutive rows shift by 4 banks (conflict-free ds_read_b128)
  constexpr int CPRX = NGX * 8, CPR2 = NG2 ? NG2 * 8 : 1;         // 16-byte chunks per pixel and source
  constexpr int LX = (32 * CPRX + 511) / 512, L2 = NG2 ? (32 * CPR2 + 511) / 512 : 0;
  static_assert(LX <= 4 && L2 <= 1, "staging registers");       // (fixed extents below: hipcc drops the host stub of a kernel template whose lambdas see dependent-extent arrays)
  auto issue = [&](int t, u32x4_ (&gx)[4], u32x4_ (&g2)[1], u32x4_ (&rv)[4], unsigned (&yrow)[4]) {
    // one division per tile: the tile's first pixel (uniform); a tile crosses at most one image boundary (OHW >= 32)
    int b0, p0;
    tile_origin(t, b0, p0);
#pragma unroll
    for (int i = 0; i < LX; ++i) {
      const int e = tid + 512 * i, px = e / CPRX, ck = e - px * CPRX;
      int b, p;
      const bool ok = tile_pixel(t, b0, p0, px & 31, b, p) && px < 32;
      gx[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? (unsigned)(((long long)b * a.x_bstride + (long long)p * a.ldx) * 2 + ck * 16) : 0x80000000u, 0, 0);
    }
    if (NG2) {
      const int px = tid / CPR2, ck = tid - px * CPR2;
      int b, p;
      const bool ok = tile_pixel(t, b0, p0, px & 31, b, p) && px < 32;
      g2[0] = __builtin_amdgcn_raw_buffer_load_b128(rx2, ok ? (unsigned)(((long long)b * a.x2_bstride + (long long)p * a.ldx2) * 2 + ck * 16) : 0x80000000u, 0, 0);
    }
    // row-shaped offsets of this lane's 4 output chunks (rows 8 apart); residual chunks requested now
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int px = (lane >> 3) + 8 * j;
      int b, p;
      const bool ok2 = tile_pixel(t, b0, p0, px, b, p);
      yrow[j] = ok2 ? (unsigned)(((long long)b * a.y_bstride + (long long)p * a.ldy + cb) * 4 + (lane & 7) * 16) : 0x80000000u;   // channels: 4 bytes each in a F16X2 row
      if (RES) rv[j] = __builtin_amdgcn_raw_buffer_load_b128(rr, ok2 ? (unsigned)(((long long)b * a.r_bstride + (long long)p * a.ldr + cb) * 4 + (lane & 7) * 16) : 0x80000000u, 0, 0);
    }
  };

  dispatch_act(a.act, [&](auto actc) {
    constexpr int ACT = decltype(actc)::value;
    u32x4_ gx[4], g2[1], rvn[4];
    unsigned yrown[4];
    int t = t0;
    if (t < ntiles) issue(t, gx, g2, rvn, yrown);
    for (int it = 0; t < ntiles; t += tstride, ++it) {
      char* xb = xs[XBUFS == 2 ? (it & 1) : 0];
      if (XBUFS == 1) __syncthreads();                           // every wave has read the previous tile
#pragma unroll
      for (int i = 0; i < LX; ++i) {
        const int e = tid + 512 * i, px = e / CPRX, ck = e - px * CPRX;
        if (px < 32) *(u32x4_*)(xb + px * XROW + ck * 16) = gx[i];
      }
      if (NG2) {
        const int px = tid / CPR2, ck = tid - px * CPR2;
        if (px < 32) *(u32x4_*)(xb + px * XROW + NGX * 128 + ck * 16) = g2[0];
      }
      u32x4_ rv[4];
      unsigned yrow[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { rv[j] = rvn[j]; yrow[j] = yrown[j]; }
      __syncthreads();                                           // the tile is in xs[it & 1]; every wave has left tile it - 1 (whose buffer the NEXT write takes)
      if (t + tstride < ntiles) issue(t + tstride, gx, g2, rvn, yrown);
      const char* xf = xb + pl * XROW + 16 * h;
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
      for (int kk = 0; kk < NKK; ++kk) {
        const sp16x8 fh = *(const sp16x8*)(xf + (kk >> 1) * 128 + 32 * (kk & 1)), fl = *(const sp16x8*)(xf + (kk >> 1) * 128 + 32 * (kk & 1) + 64);
        acc = mfma_pair32(wfh[kk], fl, acc);
        acc = mfma_pair32(wfl[kk], fh, acc);
        acc = mfma_pair32(wfh[kk], fh, acc);
      }
      float r[16];
      if (RES) {                                                 // residual: row shape -> slab -> accumulator shape
#pragma unroll
        for (int j = 0; j < 4; ++j) *(u32x4_*)(sl_row + j * 8 * SROW) = rv[j];
        __builtin_amdgcn_wave_barrier();
        const sp16x8 h0 = *(const sp16x8*)sl_acc, h1 = *(const sp16x8*)(sl_acc + 16), l0 = *(const sp16x8*)(sl_acc + 64), l1 = *(const sp16x8*)(sl_acc + 80);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int e = 0; e < 8; ++e) { r[e] = (float)h0[e] + (float)l0[e]; r[8 + e] = (float)h1[e] + (float)l1[e]; }
      }
      if (F32OUT) {                                              // fp32 rows: the wave's 32 channels are one 128-byte run per pixel as well
        char* sf = sl + pl * SROW + 64 * h;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = act_c<ACT>(acc[4 * q + e] + bl[4 * q + e]);
          *(f32x4*)(sf + 16 * q) = o;
        }
      } else {
      sp16x8 oh[2], ol[2];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float v = acc[e] + bl[e];
        if (RES && a.res_mode == RES_PRE) v += r[e];
        v = act_c<ACT>(v);
        if (RES && a.res_mode == RES_POST) v += r[e];
        sp16 hi, lo;
        split2(v, hi, lo);
        oh[e >> 3][e & 7] = hi; ol[e >> 3][e & 7] = lo;
      }
      *(sp16x8*)sl_acc = oh[0]; *(sp16x8*)(sl_acc + 16) = oh[1]; *(sp16x8*)(sl_acc + 64) = ol[0]; *(sp16x8*)(sl_acc + 80) = ol[1];
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int j = 0; j < 4; ++j) __builtin_amdgcn_raw_buffer_store_b128(*(const u32x4_*)(sl_row + j * 8 * SROW), ry, yrow[j], 0, 0);
      if constexpr (AVG) {
        // lane -> averaged pixel lane / 8 of the patch's 8, channels 4 (lane % 8) .. + 3 of the wave's group; its four sources are the patch's
        // pixels (row 0 | 1, column 2 pp | 2 pp + 1) = tile pixels 2 pp, 2 pp + 1, 16 + 2 pp, 17 + 2 pp
        const int pp = lane >> 3, cq = lane & 7;
        const char* s0 = sl + (2 * pp) * SROW + cq * 8;
        float o[4];
        {
          const sp16x4 h00 = *(const sp16x4*)s0, l00 = *(const sp16x4*)(s0 + 64), h01 = *(const sp16x4*)(s0 + SROW), l01 = *(const sp16x4*)(s0 + SROW + 64);
          const sp16x4 h10 = *(const sp16x4*)(s0 + 16 * SROW), l10 = *(const sp16x4*)(s0 + 16 * SROW + 64);
          const sp16x4 h11 = *(const sp16x4*)(s0 + 17 * SROW), l11 = *(const sp16x4*)(s0 + 17 * SROW + 64);
#pragma unroll
          for (int e = 0; e < 4; ++e)
            o[e] = ((((float)h00[e] + (float)l00[e]) + ((float)h01[e] + (float)l01[e])) + (((float)h10[e] + (float)l10[e]) + ((float)h11[e] + (float)l11[e]))) * 0.25f;
        }
        sp16x4 ah, al;
#pragma unroll
        for (int e = 0; e < 4; ++e) { sp16 hi, lo; split2(o[e], hi, lo); ah[e] = hi; al[e] = lo; }
        int b0, p0;
        tile_origin(t, b0, p0);
        const int oy2 = (p0 / a.OW) >> 1, ox2 = ((p0 % a.OW) >> 1) + pp;
        const long long ab = 4 * ((long long)b0 * ao.bstride + ((long long)oy2 * (a.OW >> 1) + ox2) * ao.ldy + cb) + cq * 8;      // bytes
        typedef unsigned int u32x2_ __attribute__((ext_vector_type(2)));
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_, ah), rya, (unsigned)ab, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_, al), rya, (unsigned)ab + 64u, 0, 0);
      }
      if (NEXTN) {
        __syncthreads();                                         // the eight slabs hold the tile's 256 output channels (hi | lo, activated)
        const int r16 = lane & 15, kq = lane >> 4;
#pragma unroll
        for (int hf = 0; hf < (NEXTN == 64 ? 1 : 2); ++hf) {
          const int half = NEXTN == 64 ? (wv >> 2) : hf;
          f32x4 c = {0.f, 0.f, 0.f, 0.f};                        // pixel 16 half + (lane & 15), channels 16 slice + 4 (lane >> 4) + e
#pragma unroll
          for (int s = 0; s < NS; ++s) {
            const char* sb = slabs[s] + (16 * half + r16) * SROW + kq * 16;
            const sp16x8 ph = *(const sp16x8*)sb, pq = *(const sp16x8*)(sb + 64);
            c = mfma_pair16(w1h[s], pq, c);
            c = mfma_pair16(w1l[s], ph, c);
            c = mfma_pair16(w1h[s], ph, c);
          }
          sp16x4 vh, vl;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v = c[e] + b1v[e];
            if (a.next_act == ACT_RELU) v = fmaxf(v, 0.f);
            sp16 hi, lo;
            split2(v, hi, lo);
            vh[e] = hi; vl[e] = lo;
          }
          char* yq = y1s + (16 * half + r16) * YROW + (slice >> 1) * 128 + (16 * (slice & 1) + 4 * kq) * 2;
          *(sp16x4*)yq = vh;
          *(sp16x4*)(yq + 64) = vl;
        }
        __syncthreads();                                         // every wave has read the slabs: the next tile may overwrite them; y1s is complete
        {
          constexpr int CPN = NEXTN / 4;                         // 16-byte chunks per pixel of the follower's output
          int b0, p0;
          tile_origin(t, b0, p0);
#pragma unroll
          for (int i = 0; i < (32 * CPN) / 512; ++i) {
            const int e = tid + 512 * i, px = e / CPN, ck = e - px * CPN;
            int b, p;
            const bool okn = tile_pixel(t, b0, p0, px, b, p);
            const unsigned off = okn ? (unsigned)(((long long)b * a.next_y_bstride + (long long)p * a.next_ldy) * 4 + ck * 16) : 0x80000000u;
            __builtin_amdgcn_raw_buffer_store_b128(*(const u32x4_*)(y1s + px * YROW + ck * 16), ryn, off, 0, 0);
          }
        }
      } else {
        __builtin_amdgcn_wave_barrier();
      }
    }
  });
}

// shapes the streaming pair kernel takes (ConvOpts::split_sx); per-IMAGE extents only (see the kernel comment)
static bool sx_shape_ok(const ConvArgs& a) {
  const int g_split_sx = opts_of(a).split_sx;
  const Tensor& x = a.x;
  const Tensor& y = a.y;
  const bool dual = a.x2.p != nullptr;
  if (!g_split_sx || x.dt != F16X2 || !(y.dt == F16X2 || y.dt == F32) || a.KH != 1 || a.KW != 1 || a.stride != 1 || a.pad != 0 || a.x_up2) return false;
  if (a.res_mode != RES_NONE && (a.res.dt != F16X2 || dual)) return false;
  if (y.c % 256) return false;
  // K = 256 (value projection 256 -> 1536 fp32; split_sx 4: stage-2 expand convs 256 -> 1024 + residual at 40^2): either output type
  // (N >= 1024 only: with one or two channel blocks per pixel tile the tiled kernel is faster - decoder input projection 36 vs 46 us)
  if (x.c == 256) return g_split_sx >= 3 && y.c >= 1024 && !dual && a.next_y.p == nullptr && (a.res_mode == RES_NONE || (g_split_sx >= 4 && y.dt == F16X2)) &&
                         (long long)y.h * y.w >= (g_split_sx >= 4 ? 1600 : 6400);
  if ((long long)y.h * y.w < 6400) return false;
  if (y.dt != F16X2) return false;
  if (dual) return x.c == 64 && a.x2.c == 64 && y.c == 256;
  if (x.c == 64) return true;
  return x.c == 128 && g_split_sx >= 2 && a.next_y.p == nullptr;
}
static bool dispatch_sx(const ConvK& k, const ConvArgs& a, long long x_bytes, long long x2_bytes, hipStream_t s) {
  if (!sx_shape_ok(a)) return false;
  const Tensor& y = a.y;
  const bool dual = a.x2.p != nullptr, res = a.res_mode != RES_NONE, next = a.next_y.p != nullptr;
  const long long y_bytes = ((long long)(y.n - 1) * y.bstride + ((long long)y.h * y.w - 1) * y.ld + y.c) * 4;
  long long r_bytes = 0, yn_bytes = 0;
  if (res) r_bytes = ((long long)(a.res.n - 1) * a.res.bstride + ((long long)a.res.h * a.res.w - 1) * a.res.ld + a.res.c) * 4;
  if (next) {
    if (!(y.c == 256 && (a.next_y.c == 64 || a.next_y.c == 128) && a.next_y.dt == F16X2 && a.next_y.ld % SPLIT_GROUP == 0 && ((uintptr_t)a.next_y.p & 15) == 0 &&
          a.next_kpad == conv_kpad_split(256) && a.next_w && a.next_bias && (a.next_act == ACT_RELU || a.next_act == ACT_NONE))) return false;
    yn_bytes = ((long long)(a.next_y.n - 1) * a.next_y.bstride + ((long long)a.next_y.h * a.next_y.w - 1) * a.next_y.ld + a.next_y.c) * 4;
  }
  if (y_bytes >= (1ll << 31) || r_bytes >= (1ll << 31) || yn_bytes >= (1ll << 31)) return false;
  // the averaged copy for the next stage's vd shortcut (ConvArgs::avg_y): 2 x 16 patch tiles, the two variants a stage ends with
  const bool avg = a.avg_y.p != nullptr;
  AvgOut ao{};
  if (avg) {
    const Tensor& v = a.avg_y;
    if (!(v.dt == F16X2 && y.dt == F16X2 && res && !dual && (y.h & 1) == 0 && (y.w & 15) == 0 && v.h == y.h / 2 && v.w == y.w / 2 && v.c == y.c && v.n == y.n &&
          v.ld % SPLIT_GROUP == 0 && ((uintptr_t)v.p & 15) == 0 && ((a.x.c == 64 && next && a.next_y.c == 128) || (a.x.c == 128 && !next)))) return false;
    const long long vb = ((long long)(v.n - 1) * v.bstride + ((long long)v.h * v.w - 1) * v.ld + v.c) * 4;
    if (vb >= (1ll << 31)) return false;
    ao.y = (sp16*)v.p; ao.ldy = v.ld; ao.bstride = v.bstride; ao.y_bytes = (unsigned)vb;
  }
  const long long ntiles = avg ? (long long)y.n * (y.h / 2) * (y.w / 16) : ((long long)k.M + 31) / 32;
  if (ntiles >= (1ll << 30)) return false;
  const int ny = y.c / 256;
  // persistent, two 8-wave blocks per CU: 8 XCDs x nts tile streams x ny channel blocks
  const int nts = (int)std::max<long long>(1, std::min<long long>(64 / ny, (ntiles + 7) / 8));
  const dim3 grid((unsigned)(8 * nts * ny)), blk(512);
#define RTD_SX(NGX, NG2, RES_, NX) rtd_launch((conv1x1_sx_kernel<NGX, NG2, RES_, NX>), grid, blk, 0, s, k, (unsigned)x_bytes, (unsigned)r_bytes, \
                                                     (unsigned)y_bytes, (unsigned)x2_bytes, (int)ntiles, (unsigned)yn_bytes, ny, ao)
  const int nx = next ? a.next_y.c : 0;
  if (avg) {
    // ConvArgs::y_dead: y's only readers are the two fused consumers of this very launch - a zero-byte descriptor drops its stores
    const unsigned yb = (a.y_dead && next) ? 0u : (unsigned)y_bytes;
    if (a.x.c == 64) rtd_launch((conv1x1_sx_kernel<2, 0, true, 128, false, true>), grid, blk, 0, s, k, (unsigned)x_bytes, (unsigned)r_bytes, yb,
                                        (unsigned)x2_bytes, (int)ntiles, (unsigned)yn_bytes, ny, ao);
    else rtd_launch((conv1x1_sx_kernel<4, 0, true, 0, false, true>), grid, blk, 0, s, k, (unsigned)x_bytes, (unsigned)r_bytes, (unsigned)y_bytes,
                            (unsigned)x2_bytes, (int)ntiles, (unsigned)yn_bytes, ny, ao);
  } else if (dual) { if (nx == 64) RTD_SX(2, 2, false, 64); else if (nx == 128) RTD_SX(2, 2, false, 128); else RTD_SX(2, 2, false, 0); }
  else if (a.x.c == 64 && res) { if (nx == 64) RTD_SX(2, 0, true, 64); else if (nx == 128) RTD_SX(2, 0, true, 128); else RTD_SX(2, 0, true, 0); }
  else if (a.x.c == 64) { if (nx == 64) RTD_SX(2, 0, false, 64); else if (nx == 128) RTD_SX(2, 0, false, 128); else RTD_SX(2, 0, false, 0); }
  else if (a.x.c == 256 && y.dt == F32) rtd_launch((conv1x1_sx_kernel<8, 0, false, 0, true>), grid, blk, 0, s, k, (unsigned)x_bytes, 0u, (unsigned)y_bytes, 0u, (int)ntiles, 0u, ny, ao);
  else if (a.x.c == 256 && res) RTD_SX(8, 0, true, 0);
  else if (a.x.c == 256) RTD_SX(8, 0, false, 0);
  else if (res) RTD_SX(4, 0, true, 0);
  else RTD_SX(4, 0, false, 0);
#undef RTD_SX
  return true;
}


// Second pass of the two-pass split-K: y = act(sum_s slab[s] + bias (+ res)), slices in fixed order (one summation order per output whatever the
// batch size), written in the layer's own output format.  One thread = 8 channels of a pixel.
__global__ __launch_bounds__(256) void k_splitk_reduce(const ConvK a, const float* __restrict__ slab, int S) {
  const int n8 = a.N >> 3;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)a.M * n8) return;
  const int m = (int)(idx / n8), c = (int)(idx - (long long)m * n8) << 3;
  const long long MN = (long long)a.M * a.N;
  const float* q = slab + (long long)m * a.N + c;
  f32x4 s0 = *(const f32x4*)q, s1 = *(const f32x4*)(q + 4);
  for (int s = 1; s < S; ++s) {
    const f32x4 t0 = *(const f32x4*)(q + s * MN), t1 = *(const f32x4*)(q + s * MN + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { s0[e] += t0[e]; s1[e] += t1[e]; }
  }
  const f32x4 b0 = *(const f32x4*)(a.bias + c), b1 = *(const f32x4*)(a.bias + c + 4);
  float v[8] = {s0[0] + b0[0], s0[1] + b0[1], s0[2] + b0[2], s0[3] + b0[3], s1[0] + b1[0], s1[1] + b1[1], s1[2] + b1[2], s1[3] + b1[3]};
  const int b = m / a.OHW, p = m - b * a.OHW;
  const long long ypix = (long long)b * a.y_bstride + (long long)p * a.ldy;
  float rv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (a.res_mode != RES_NONE) {
    const long long rpix = (long long)b * a.r_bstride + (long long)p * a.ldr;
    if (a.res_split) split_load8((const sp16*)a.res, rpix, c, rv);
    else {
      const f32x4 t0 = *(const f32x4*)((const float*)a.res + rpix + c), t1 = *(const f32x4*)((const float*)a.res + rpix + c + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { rv[e] = t0[e]; rv[4 + e] = t1[e]; }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    if (a.res_mode == RES_PRE) v[e] += rv[e];
    v[e] = act_fn(v[e], a.act);
    if (a.res_mode == RES_POST) v[e] += rv[e];
  }
  if (a.y_split) split_store8((sp16*)a.y, ypix, c, v);
  else {
    *(f32x4*)((float*)a.y + ypix + c) = f32x4{v[0], v[1], v[2], v[3]};
    *(f32x4*)((float*)a.y + ypix + c + 4) = f32x4{v[4], v[5], v[6], v[7]};
  }
}
bool conv_split_supported(const ConvArgs& a) {
  const Tensor& x = a.x;
  const Tensor& y = a.y;
  if (x.dt != F16X2 || !(y.dt == F16X2 || y.dt == F32)) return false;
  if (x.c % SPLIT_GROUP || x.ld % SPLIT_GROUP || ((uintptr_t)x.p & 15) || ((uintptr_t)a.w & 15)) return false;
  if (a.x2.p && (a.x2.dt != F16X2 || a.x2.c % SPLIT_GROUP || a.x2.ld % SPLIT_GROUP || ((uintptr_t)a.x2.p & 15))) return false;
  if (a.x_up2 && !(a.KH == 1 && a.KW == 1 && a.stride == 1 && a.pad == 0 && a.x2.p)) return false;
  if (y.dt == F16X2 && (y.c % SPLIT_GROUP || y.ld % SPLIT_GROUP)) return false;
  if (y.c % 8 || y.ld % 4 || ((uintptr_t)y.p & 15)) return false;
  if (a.res_mode != RES_NONE) {
    const Tensor& r = a.res;
    if (!(r.dt == F16X2 || r.dt == F32) || !r.p || ((uintptr_t)r.p & 15) || r.ld % 4) return false;
    if (r.dt == F16X2 && r.ld % SPLIT_GROUP) return false;
  }
  return a.next_y.p == nullptr || sx_shape_ok(a);
}
// Two-pass split-K (ConvOpts::split_k2) on long-K layers whose per-IMAGE tile count is small (stage 3, the 20^2 PAN level, enc.proj.2 at
// 640 px; most layers at smaller inputs).  The slice count depends on per-image extents and K only: every batch size runs the same
// arithmetic (the plan builder sizes the workspace from conv_split_slab_bytes() for its own batch).  Measured on R50 640^2 (stage-3 3x3,
// K = 4608: 144 K-steps): batch 1 54 -> 35 us per layer (16 blocks -> 32), batch 8 57 -> 62 us; shorter K loops (stage-3 1x1, the 20^2 PAN
// level) lose at batch 8 and gain nothing at batch 1, hence the 128-step floor.
static int split_k2_slices(const ConvOpts& o, const ConvArgs& a, int OH, int OW) {
  const bool dual = a.x2.p != nullptr;
  if (!o.split_k2 || dual || a.x_up2 || a.y.c % 8 || a.next_y.p) return 1;
  if (sx_shape_ok(a)) return 1;
  const long long tiles_img = (((long long)OH * OW + 127) / 128) * ((a.y.c + 127) / 128);
  const int groups = a.x.c / SPLIT_GROUP, nk_total = a.KH * a.KW * groups;
  if (nk_total >= 128 && tiles_img <= 8 && groups % 4 == 0) return 4;
  if (nk_total >= 128 && tiles_img <= 16 && groups % 2 == 0) return 2;
  return 1;
}
size_t conv_split_slab_bytes(const ConvArgs& a) {
  if (a.x.dt != F16X2) return 0;
  const int up = a.x_up2 ? 2 : 1;
  const int OH = (a.x.h * up + 2 * a.pad - a.KH) / a.stride + 1, OW = (a.x.w * up + 2 * a.pad - a.KW) / a.stride + 1;
  const int S = split_k2_slices(opts_of(a), a, OH, OW);
  return S > 1 ? (size_t)S * (size_t)a.x.n * OH * OW * a.y.c * 4 : 0;
}

static void launch_conv_split(const ConvArgs& a, hipStream_t s) {
  const ConvOpts& o = opts_of(a);
  const Tensor& x = a.x;
  const Tensor& y = a.y;
  RTD_CHECK(conv_split_supported(a), 1, "conv (f16x3): shape / layout not supported by the pair kernels (channels % 32, 16-byte alignment)");
  const int up = a.x_up2 ? 2 : 1;
  const int OH = (x.h * up + 2 * a.pad - a.KH) / a.stride + 1;
  const int OW = (x.w * up + 2 * a.pad - a.KW) / a.stride + 1;
  RTD_CHECK(OH == y.h && OW == y.w && x.n == y.n, 1, "conv: output shape mismatch");
  const bool dual = a.x2.p != nullptr;
  if (dual) RTD_CHECK(a.x2.n == x.n && a.x2.h == OH && a.x2.w == OW, 1, "conv: second input shape");
  const int K = a.KH * a.KW * x.c + (dual ? a.x2.c : 0);
  RTD_CHECK(a.Kpad == conv_kpad_split(K) && a.Npad >= y.c && a.Npad % 128 == 0, 1, "conv: pair filter padding");
  RTD_CHECK((long long)x.n * OH * OW < (1ll << 31), 1, "conv: M overflow");
  ConvG g;
  ConvK& k = g.k;
  k.x = x.p; k.w = a.w; k.bias = a.bias; k.y = y.p;
  k.res = a.res_mode != RES_NONE ? a.res.p : nullptr;
  k.M = x.n * OH * OW; k.H = x.h * up; k.W = x.w * up;
  k.Cin = 2 * x.c; k.ldx = 2 * x.ld; k.x_bstride = 2 * x.bstride;                    // 16-bit elements
  k.x_up2 = a.x_up2;
  k.next_w = a.next_w; k.next_bias = a.next_bias; k.next_y = a.next_y.p; k.next_ldy = a.next_y.ld; k.next_y_bstride = a.next_y.bstride;   // channels
  k.next_kpad = a.next_kpad; k.next_act = a.next_act;
  k.OH = OH; k.OW = OW; k.OHW = OH * OW;
  k.N = y.c; k.Kreal = 2 * K; k.Kpad = a.Kpad;
  k.KH = a.KH; k.KW = a.KW; k.stride = a.stride; k.pad = a.pad;
  k.ldy = y.ld; k.y_bstride = y.bstride;                                            // channels (fp32 or F16X2 alike)
  k.ldr = 0; k.r_bstride = 0; k.res_f32 = 0;
  if (a.res_mode != RES_NONE) {
    RTD_CHECK(a.res.n == y.n && a.res.h == y.h && a.res.w == y.w && a.res.c == y.c, 1, "conv: residual shape");
    k.ldr = a.res.ld; k.r_bstride = a.res.bstride; k.res_f32 = a.res.dt == F32;
  }
  k.act = a.act; k.res_mode = a.res_mode; k.y_f32 = y.dt == F32;
  k.split = 1; k.y_split = y.dt == F16X2; k.res_split = a.res_mode != RES_NONE && a.res.dt == F16X2;
  k.x2 = nullptr; k.ldx2 = 0; k.x2_bstride = 0; k.k2_start = 0;
  long long x2_bytes = 0;
  if (dual) {
    k.x2 = a.x2.p; k.ldx2 = 2 * a.x2.ld; k.x2_bstride = 2 * a.x2.bstride; k.k2_start = 2 * a.KH * a.KW * x.c;
    x2_bytes = ((long long)(a.x2.n - 1) * a.x2.bstride + ((long long)a.x2.h * a.x2.w - 1) * a.x2.ld + a.x2.c) * 4;
  }
  k.reg_epi = 1;
  k.prefer256 = 0;
  k.pf = o.prefetch ? a.pf : nullptr;
  k.pf_bytes = (o.prefetch && a.pf && a.pf_bytes < (1ull << 31)) ? (unsigned)a.pf_bytes : 0u;
  const long long x_bytes = ((long long)(x.n - 1) * x.bstride + ((long long)x.h * x.w - 1) * x.ld + x.c) * 4;
  const long long w_bytes = (long long)a.Npad * a.Kpad * 2;
  RTD_CHECK(x_bytes < (1ll << 31) && x2_bytes < (1ll << 31) && w_bytes < (1ll << 31), 1, "conv (f16x3): operand larger than a buffer descriptor (2 GiB)");
  g.probe = o.glds_drop & ~32; g.splitk = 1; g.slab = nullptr; g.y_bytes = 0;      // timing-only probes (rtd_debug_option "glds_drop")
  g.x_bytes = (o.glds_drop & 1) ? 0u : (unsigned)x_bytes; g.w_bytes = (o.glds_drop & 2) ? 0u : (unsigned)w_bytes; g.x2_bytes = (unsigned)x2_bytes;
  // thin 1x1 expand convs on wide grids (stage-0 / stage-1 c3): the streaming kernel, with the next block's reduce conv riding on it
  if (dispatch_sx(k, a, x_bytes, x2_bytes, s)) { HIP_CHECK(hipGetLastError()); return; }
  RTD_CHECK(a.next_y.p == nullptr, 1, "conv (f16x3): a fused following conv exists in the streaming kernel only");
  RTD_CHECK(a.avg_y.p == nullptr, 1, "conv (f16x3): the fused vd-shortcut average exists in the streaming kernel only (the plan must fall back to the avg-pool launch)");
  // Direct 3x3 kernels for the narrow layers on wide maps.  Chosen on the per-IMAGE tile count: every batch size runs the same arithmetic
  // (batch invariance is bit-exact).  32 input channels (stem.1, stem.2): every input pixel is staged once instead of nine times
  if (o.conv_reg && a.KH == 3 && a.KW == 3 && a.stride == 1 && a.pad == 1 && !dual && a.res_mode == RES_NONE && x.c == 32 && (y.c == 32 || y.c == 64) &&
      y.dt == F16X2) {
    const long long y_bytes = ((long long)(y.n - 1) * y.bstride + ((long long)y.h * y.w - 1) * y.ld + y.c) * 4;
    const int tiles_x = (x.w + 31) / 32, tiles_y = (x.h + 7) / 8;
    const long long ntiles = (long long)x.n * tiles_x * tiles_y;
    if (tiles_x * tiles_y >= 32 && ntiles < (1ll << 30) && y_bytes < (1ll << 31)) {
      const unsigned gx = (unsigned)std::min<long long>(ntiles, 256);        // persistent, one block per CU
      if (y.c == 32) rtd_launch((conv3x3_reg_split_kernel<1>), dim3(gx), dim3(512), 0, s, k, (unsigned)x_bytes, (unsigned)y_bytes, tiles_x, tiles_y, (int)ntiles, PoolOut{});
      else rtd_launch((conv3x3_reg_split_kernel<2>), dim3(gx), dim3(512), 0, s, k, (unsigned)x_bytes, (unsigned)y_bytes, tiles_x, tiles_y, (int)ntiles, PoolOut{});
      HIP_CHECK(hipGetLastError());
      return;
    }
  }
  // 64 -> 64 channels (stage-0 c2): direct kernel on 8 x 16 tiles
  if ((o.conv_reg & 2) && a.KH == 3 && a.KW == 3 && a.stride == 1 && a.pad == 1 && !dual && a.res_mode == RES_NONE && x.c == 64 && y.c == 64 && y.dt == F16X2) {
    const long long y_bytes = ((long long)(y.n - 1) * y.bstride + ((long long)y.h * y.w - 1) * y.ld + y.c) * 4;
    const int tiles_x = (x.w + 15) / 16, tiles_y = (x.h + 7) / 8;
    const long long ntiles = (long long)x.n * tiles_x * tiles_y;
    if (tiles_x * tiles_y >= 128 && ntiles < (1ll << 30) && y_bytes < (1ll << 31)) {
      const unsigned gx = (unsigned)std::min<long long>(ntiles, 256);        // persistent, one block per CU
      rtd_launch(conv3x3_reg_split64_kernel, dim3(gx), dim3(512), 0, s, k, (unsigned)x_bytes, (unsigned)y_bytes, tiles_x, tiles_y, (int)ntiles);
      HIP_CHECK(hipGetLastError());
      return;
    }
  }
  // ---- two-pass split-K: S slices of the channel groups, bare fp32 partial sums into the workspace slab, k_splitk_reduce finishes
  const int S = split_k2_slices(o, a, OH, OW);
  if (S > 1) RTD_CHECK(a.ws.slab && (size_t)S * (size_t)k.M * (size_t)k.N * 4 <= a.ws.slab_bytes, 1,
                       "conv (f16x3): the launch needs a split-K workspace of conv_split_slab_bytes() bytes (ConvArgs::ws)");
  const ConvK korig = k;
  if (S > 1) {
    k.raw = 1; k.act = ACT_NONE; k.res_mode = RES_NONE; k.res = nullptr; k.y = a.ws.slab; k.y_f32 = 1; k.y_split = 0; k.res_split = 0;
    k.ldy = k.N; k.y_bstride = (long long)k.OHW * k.N; g.splitk = S;
  }
  auto finish = [&]() {
    HIP_CHECK(hipGetLastError());
    if (S > 1) {
      const long long items = (long long)korig.M * (korig.N >> 3);
      rtd_launch(k_splitk_reduce, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, s, korig, (const float*)a.ws.slab, S);
      HIP_CHECK(hipGetLastError());
    }
  };
  const long long mt = (k.M + 127) / 128, ntn = (k.N + 127) / 128, ntn64 = (k.N + 63) / 64;
  // ---- flexible tile height (conv_igemm_wsf_kernel) on grids that leave CUs idle with 128-pixel tiles (20^2 maps at batch 8, most layers at
  // batch 1: R50 bs-1 latency 2.49 -> 2.36 ms): the tile whose grid fills whole rounds of the chip
  if (o.split_flex && k.Kpad / 64 / S >= o.split_flex_min_nk && k.N >= 64 && mt * ntn * S <= o.split_flex_small_max) {
    int best_mt = 0, best_bn = 0, best_st = 0;
    double best = 1e30;
    for (int bn = (k.N > 64 ? 128 : 64); bn >= 64; bn -= 64) {
      const long long ntb = (k.N + bn - 1) / bn;
      for (int mtc = 4; mtc <= 13; ++mtc) {
        const long long blocks = ((k.M + 16 * mtc - 1) / (16 * mtc)) * ntb * S;
        const long long rounds = (blocks + 255) / 256;
        const bool two = blocks > 256 && mtc <= 8;                 // 2 stages, two blocks per CU share the MFMA pipes; else 3 stages, one block per CU
        double eff = (bn == 128 ? 1.0 : 0.78) * (1.0 - 0.6 / mtc);  // small tiles: more LDS reads and barriers per MFMA
        if (!two && blocks > 256) eff *= 0.93;                     // nothing hides a lone block's barrier bubbles
        const double cost = (double)rounds * mtc * bn / eff;
        if (cost < best) { best = cost; best_mt = mtc; best_bn = bn; best_st = two ? 2 : 3; }
      }
    }
    k.ntn = (int)((k.N + best_bn - 1) / best_bn);
    const long long blocks = ((k.M + 16 * best_mt - 1) / (16 * best_mt)) * k.ntn * S;
    const dim3 grid((unsigned)blocks), blk(512);
    bool launched = true;
#define RTD_WSF(ST, BNN, MTT) rtd_launch((conv_igemm_wsf_kernel<ST, BNN, MTT>), grid, blk, 0, s, g)
#define RTD_WSF_MT(ST, BNN)                                                                                     \
    switch (best_mt) {                                                                                         \
      case 4: RTD_WSF(ST, BNN, 4); break; case 5: RTD_WSF(ST, BNN, 5); break; case 6: RTD_WSF(ST, BNN, 6); break;   \
      case 7: RTD_WSF(ST, BNN, 7); break; case 8: RTD_WSF(ST, BNN, 8); break; default: launched = false; break; }
    if (best_st == 2) {
      if (best_bn == 128) { RTD_WSF_MT(2, 128) } else launched = false;
    } else if (best_bn == 64) {
      RTD_WSF_MT(3, 64)
    } else {
      switch (best_mt) {
        case 4: RTD_WSF(3, 128, 4); break; case 5: RTD_WSF(3, 128, 5); break; case 6: RTD_WSF(3, 128, 6); break; case 7: RTD_WSF(3, 128, 7); break;
        case 8: RTD_WSF(3, 128, 8); break; case 9: RTD_WSF(3, 128, 9); break; case 10: RTD_WSF(3, 128, 10); break; case 11: RTD_WSF(3, 128, 11); break;
        case 12: RTD_WSF(3, 128, 12); break; case 13: RTD_WSF(3, 128, 13); break; default: launched = false; break;
      }
    }
#undef RTD_WSF_MT
#undef RTD_WSF
    if (launched) { finish(); return; }
    k.ntn = 1;
  }
  // ---- grids of more than one round: (MTA + MTB) x 16 pixel tiles, one block per CU (conv_igemm_wsq_kernel).  The tile height minimises
  // rounds x (K loop of the taller wave half + epilogue), in MFMA-pipe cycles
  if (o.split_wsq && S == 1 && k.N > 64 && k.Kpad / 64 >= o.split_wsq_min_nk && (o.split_wsq == 2 || mt * ntn >= o.split_wsq_min_blocks)) {
    static const int cand[7][2] = {{5, 5}, {6, 5}, {6, 6}, {7, 6}, {7, 7}, {8, 7}, {8, 8}};
    const long long nk = k.Kpad / 64;
    int best = -1;
    double best_cost = 1e30;
    for (int c = 0; c < 7; ++c) {
      const int bm = 16 * (cand[c][0] + cand[c][1]);
      const long long blocks = ((k.M + bm - 1) / bm) * ntn;
      const double cost = (double)((blocks + 255) / 256) * ((double)nk * cand[c][0] * 192.0 + 1500.0 + 10.0 * bm);
      if (cost < best_cost) { best_cost = cost; best = c; }
    }
    const bool stamped = (o.glds_drop & 32) && a.ws.slab && a.ws.slab_bytes >= (size_t)4096 * 64;
    if (stamped && best != 6) best = 4;                        // the diagnostic build exists for 7 + 7 and 8 + 8 tiles only
    const int bm = 16 * (cand[best][0] + cand[best][1]);
    k.ntn = (int)ntn;
    const dim3 grid((unsigned)(((k.M + bm - 1) / bm) * ntn)), blk(512);
    if (stamped) {     // diagnostic build: block stamps
      g.slab = a.ws.slab;
      if (best == 4) rtd_launch((conv_igemm_wsq_kernel<7, 7, true>), grid, blk, 0, s, g);
      else rtd_launch((conv_igemm_wsq_kernel<8, 8, true>), grid, blk, 0, s, g);
      finish();
      return;
    }
    switch (best) {
      case 0: rtd_launch((conv_igemm_wsq_kernel<5, 5>), grid, blk, 0, s, g); break;
      case 1: rtd_launch((conv_igemm_wsq_kernel<6, 5>), grid, blk, 0, s, g); break;
      case 2: rtd_launch((conv_igemm_wsq_kernel<6, 6>), grid, blk, 0, s, g); break;
      case 3: rtd_launch((conv_igemm_wsq_kernel<7, 6>), grid, blk, 0, s, g); break;
      case 4: rtd_launch((conv_igemm_wsq_kernel<7, 7>), grid, blk, 0, s, g); break;
      case 5: rtd_launch((conv_igemm_wsq_kernel<8, 7>), grid, blk, 0, s, g); break;
      default: rtd_launch((conv_igemm_wsq_kernel<8, 8>), grid, blk, 0, s, g); break;
    }
    finish();
    return;
  }
  // ---- fixed 128 x 128 / 128 x 64 tiles (conv_igemm_wsx_kernel): 4 stages at one block per CU below split_ws2_min_blocks blocks, 2 stages from there
  const bool n64 = k.N <= 64 || (mt * ntn < o.split_ws64_max_blocks && ntn64 > ntn);
  k.ntn = (int)(n64 ? ntn64 : ntn);
  const long long blocks = mt * k.ntn * S;
  const bool four = blocks < o.split_ws2_min_blocks;
  const dim3 grid((unsigned)blocks), blk(512);
  if (n64) { if (four) rtd_launch((conv_igemm_wsx_kernel<4, 64>), grid, blk, 0, s, g); else rtd_launch((conv_igemm_wsx_kernel<2, 64>), grid, blk, 0, s, g); }
  else { if (four) rtd_launch((conv_igemm_wsx_kernel<4, 128>), grid, blk, 0, s, g); else rtd_launch((conv_igemm_wsx_kernel<2, 128>), grid, blk, 0, s, g); }
  finish();
}

// ---- stem.2 -> max-pool in one pass (conv3x3_reg_split_kernel<2, true> + k_pool_fixup) ----------------------------------------------
// a = the 3x3 / stride 1 / pad 1 conv 32 -> 64 channels with ReLU whose output a.y would be pooled 3x3 / stride 2 / pad 1 into `pooled`;
// a.y.p is not touched (the plan does not allocate it).  `side` = conv_pool_side_bytes(a) bytes of scratch.
bool conv_pool_supported(const ConvArgs& a, const Tensor& pooled) {
  const Tensor& x = a.x;
  const Tensor& y = a.y;
  if (x.dt != F16X2 || y.dt != F16X2 || pooled.dt != F16X2) return false;
  if (!(a.KH == 3 && a.KW == 3 && a.stride == 1 && a.pad == 1 && !a.x2.p && !a.next_y.p && a.res_mode == RES_NONE && a.act == ACT_RELU)) return false;
  if (x.c != 32 || y.c != 64 || pooled.c != 64 || x.ld % SPLIT_GROUP || pooled.ld % SPLIT_GROUP) return false;
  if (y.h != x.h || y.w != x.w || (y.h & 1) || (y.w & 1) || pooled.h != y.h / 2 || pooled.w != y.w / 2 || pooled.n != x.n) return false;
  if (((uintptr_t)x.p & 15) || ((uintptr_t)pooled.p & 15) || ((uintptr_t)a.w & 15)) return false;
  const int tiles_x = (x.w + 31) / 32, tiles_y = (x.h + 7) / 8;
  return tiles_x * tiles_y >= 32 && opts_of(a).conv_reg != 0;
}
size_t conv_pool_side_bytes(const ConvArgs& a) {
  const long long ntiles = (long long)a.x.n * ((a.x.w + 31) / 32) * ((a.x.h + 7) / 8);
  return (size_t)ntiles * (8192 + 1024) + 256;
}
void launch_conv_pool(const ConvArgs& a, const Tensor& pooled, void* side, hipStream_t s) {
  RTD_CHECK(conv_pool_supported(a, pooled) && side, 1, "conv + max-pool: shape not supported");
  const Tensor& x = a.x;
  ConvK k;
  k.x = x.p; k.w = a.w; k.bias = a.bias; k.y = nullptr; k.res = nullptr;
  k.H = x.h; k.W = x.w; k.M = x.n * x.h * x.w;
  k.Cin = 2 * x.c; k.ldx = 2 * x.ld; k.x_bstride = 2 * x.bstride;
  k.OH = x.h; k.OW = x.w; k.OHW = x.h * x.w;
  k.N = 64; k.Kreal = 2 * 9 * x.c; k.Kpad = a.Kpad;
  k.KH = 3; k.KW = 3; k.stride = 1; k.pad = 1;
  k.ldy = 0; k.y_bstride = 0; k.ldr = 0; k.r_bstride = 0; k.res_f32 = 0;
  k.act = a.act; k.res_mode = RES_NONE; k.y_f32 = 0; k.split = 1; k.y_split = 1; k.res_split = 0;
  k.x2 = nullptr; k.ldx2 = 0; k.x2_bstride = 0; k.k2_start = 0; k.ntn = 1; k.reg_epi = 1; k.prefer256 = 0; k.pf = nullptr; k.pf_bytes = 0; k.x_up2 = 0;
  k.next_w = nullptr; k.next_bias = nullptr; k.next_y = nullptr; k.next_ldy = 0; k.next_y_bstride = 0; k.next_kpad = 0; k.next_act = 0;
  const long long x_bytes = ((long long)(x.n - 1) * x.bstride + ((long long)x.h * x.w - 1) * x.ld + x.c) * 4;
  const long long p_bytes = ((long long)(pooled.n - 1) * pooled.bstride + ((long long)pooled.h * pooled.w - 1) * pooled.ld + pooled.c) * 4;
  const int tiles_x = (x.w + 31) / 32, tiles_y = (x.h + 7) / 8;
  const long long ntiles = (long long)x.n * tiles_x * tiles_y;
  RTD_CHECK(x_bytes < (1ll << 31) && p_bytes < (1ll << 31) && ntiles * 8192 < (1ll << 31), 1, "conv + max-pool: operand larger than a buffer descriptor");
  PoolOut po;
  po.y = (sp16*)pooled.p; po.ldy = pooled.ld; po.bstride = pooled.bstride; po.OH = pooled.h; po.OW = pooled.w; po.y_bytes = (unsigned)p_bytes;
  po.side_row = (sp16*)side; po.row_bytes = (unsigned)(ntiles * 8192);
  po.side_col = (sp16*)((char*)side + ntiles * 8192); po.col_bytes = (unsigned)(ntiles * 1024);
  const unsigned gx = (unsigned)std::min<long long>(ntiles, 256);
  rtd_launch((conv3x3_reg_split_kernel<2, true>), dim3(gx), dim3(512), 0, s, k, (unsigned)x_bytes, 0u, tiles_x, tiles_y, (int)ntiles, po);
  HIP_CHECK(hipGetLastError());
  const long long items = ntiles * 19 * (64 / 8);
  rtd_launch(k_pool_fixup, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, s, po, tiles_x, tiles_y, (int)ntiles, 64);
  HIP_CHECK(hipGetLastError());
}

void launch_conv(const ConvArgs& a, hipStream_t s) {
  const Tensor& x = a.x;
  const Tensor& y = a.y;
  if (x.dt == F16X2) { launch_conv_split(a, s); return; }
  const ConvOpts& o = opts_of(a);
  RTD_CHECK(x.dt == BF16 || x.dt == F32, 1, "conv: input dtype");
  RTD_CHECK(y.dt == BF16 || y.dt == F32 || (y.dt == F16X2 && x.dt == F32), 1, "conv: output dtype");
  if (y.dt == F16X2) RTD_CHECK(y.c % SPLIT_GROUP == 0 && y.ld % SPLIT_GROUP == 0 && a.res_mode == RES_NONE && !a.x2.p && !a.next_y.p && ((uintptr_t)y.p & 15) == 0, 1,
                                "conv: fp32 -> F16X2 output needs 32-channel groups and no residual / second input");
  const int epc = x.dt == BF16 ? 8 : 4;
  const int up = a.x_up2 ? 2 : 1;
  if (a.x_up2) RTD_CHECK(a.KH == 1 && a.KW == 1 && a.stride == 1 && a.pad == 0 && a.x2.p, 1, "conv: x_up2 needs a 1x1 conv with a second input");
  const int OH = (x.h * up + 2 * a.pad - a.KH) / a.stride + 1;
  const int OW = (x.w * up + 2 * a.pad - a.KW) / a.stride + 1;
  RTD_CHECK(OH == y.h && OW == y.w && x.n == y.n, 1, "conv: output shape mismatch");
  RTD_CHECK(x.c % epc == 0 && x.ld % epc == 0, 1, "conv: Cin / pixel stride must be a multiple of one 16-byte chunk");
  RTD_CHECK(((uintptr_t)x.p & 15) == 0 && ((uintptr_t)a.w & 15) == 0, 1, "conv: 16-byte alignment");
  RTD_CHECK(y.c % 4 == 0 && y.ld % 4 == 0, 1, "conv: Cout / output stride must be multiples of 4");
  RTD_CHECK(((uintptr_t)y.p & (y.dt == BF16 ? 7 : 15)) == 0, 1, "conv: output alignment");
  const bool dual = a.x2.p != nullptr;
  const int K = a.KH * a.KW * x.c + (dual ? a.x2.c : 0);
  if (dual) RTD_CHECK(conv_dual_supported(a), 1, "conv: second input not supported for this shape (see conv_dual_supported)");
  RTD_CHECK(a.Kpad == conv_kpad(K) && a.Npad >= y.c && a.Npad % 128 == 0, 1, "conv: filter padding");
  RTD_CHECK((long long)x.n * OH * OW < (1ll << 31), 1, "conv: M overflow");
  ConvK k;
  k.x = x.p; k.w = a.w; k.bias = a.bias; k.y = y.p;
  k.res = a.res_mode != RES_NONE ? a.res.p : nullptr;
  k.M = x.n * OH * OW; k.H = x.h * up; k.W = x.w * up; k.Cin = x.c;
  k.x_up2 = a.x_up2;
  k.next_w = a.next_w; k.next_bias = a.next_bias; k.next_y = a.next_y.p; k.next_ldy = a.next_y.ld; k.next_y_bstride = a.next_y.bstride;
  k.next_kpad = a.next_kpad; k.next_act = a.next_act;
  if (a.next_y.p) RTD_CHECK(a.next_y.n == y.n && a.next_y.h == y.h && a.next_y.w == y.w, 1, "conv: fused next conv output shape");
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
  k.y_split = y.dt == F16X2;
  k.x2 = nullptr; k.ldx2 = 0; k.x2_bstride = 0; k.k2_start = 0;
  long long x2_bytes = 0;
  if (dual) {
    k.x2 = a.x2.p; k.ldx2 = a.x2.ld; k.x2_bstride = a.x2.bstride; k.k2_start = a.KH * a.KW * x.c;
    x2_bytes = ((long long)(a.x2.n - 1) * a.x2.bstride + ((long long)a.x2.h * a.x2.w - 1) * a.x2.ld + a.x2.c) * (long long)dtype_size(x.dt);
  }
  k.ntn = 1;
  k.reg_epi = o.reg_epilogue;
  k.prefer256 = a.prefer256;
  k.pf = o.prefetch ? a.pf : nullptr;
  k.pf_bytes = (o.prefetch && a.pf && a.pf_bytes < (1ull << 31)) ? (unsigned)a.pf_bytes : 0u;
  const bool smallc = (x.c % 32) != 0;
  const int bk2 = x.dt == BF16 ? 64 : 32;
  bool tile_ok = (x.c % bk2 == 0) && (y.c % 8 == 0) && (y.ld % 8 == 0) && (((uintptr_t)y.p & 15) == 0);
  if (a.res_mode != RES_NONE) tile_ok = tile_ok && (a.res.ld % 8 == 0) && (((uintptr_t)a.res.p & 15) == 0);
  bool done = false;
  if (k.y_split) {                        // fp32 input, F16X2 output (the split engine's stem.0): the register-staged kernel's epilogue writes it
    dispatch<float>(k, smallc, s);
    HIP_CHECK(hipGetLastError());
    return;
  }
  {
    const long long es = (long long)dtype_size(x.dt);
    const long long x_bytes = ((long long)(x.n - 1) * x.bstride + ((long long)x.h * x.w - 1) * x.ld + x.c) * es;
    const long long w_bytes = (long long)a.Npad * a.Kpad * es;
    RTD_CHECK(!a.next_y.p, 1, "conv: a fused following conv exists on F16X2 operands only (see conv_next_supported)");
    {
      const long long yb = ((long long)(y.n - 1) * y.bstride + ((long long)y.h * y.w - 1) * y.ld + y.c) * (long long)dtype_size(y.dt);
      const unsigned y_bytes = yb < (1ll << 31) ? (unsigned)yb : 0u;
      if (x.dt == BF16) done = dispatch_glds<bf16>(o, k, tile_ok, a.prefer256 != 0, x_bytes, w_bytes, y_bytes, (unsigned)x2_bytes, a.ws, s);
      else done = dispatch_glds<float>(o, k, tile_ok, a.prefer256 != 0, x_bytes, w_bytes, y_bytes, (unsigned)x2_bytes, a.ws, s);
    }
  }
  RTD_CHECK(done || !dual, 1, "conv: no kernel took the dual-input launch");
  RTD_CHECK(done || !a.next_y.p, 1, "conv: no kernel took the launch with a fused following conv (see conv_next_supported)");
  if (!done) {
    if (x.dt == BF16) dispatch<bf16>(k, smallc, s);
    else dispatch<float>(k, smallc, s);
  }
  HIP_CHECK(hipGetLastError());
}

}  // namespace rtd
