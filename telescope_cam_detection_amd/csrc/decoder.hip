// decoder.hip - one fused kernel per RT-DETR decoder layer (everything except the 300x300 self-attention).
//
// HF:rt_detr_v2/modeling_rt_detr_v2.py:339-431 (layer), :603-661 (loop), :119-225 (MS-deformable attention).
// Every op of a decoder layer other than self-attention is independent per query row, so a block
// owns 16 query rows of one image and carries them through the whole layer with the activations in
// LDS (fp32) and the weights streamed from L2 straight into MFMA B-fragments:
//
//   att -> o_proj -> +hs -> LN1 -> (+qpos) offsets|weights -> MS-deformable sampling -> out_proj -> +res -> LN2
//       -> FFN(relu) -> +res -> LN3 -> bbox MLP -> ref = sigmoid(delta + logit(ref))
//       -> [next layer]  qpos = MLP(ref), q|k = (hs+qpos) Wqk, v = hs Wv        (or the class head on the last layer)
//
// This replaces ~19 launches per layer (2400 x 256 GEMMs that were latency-bound at ~20 us each) by one.
// The arithmetic is exact fp32: v_mfma_f32_16x16x4_f32 (an fp32 fma chain) - the decoder carries the box
// refinement chain and the class logits, which is where the 1e-3 / 1e-2 px parity bar is decided.
#include "common.h"

namespace rtd {

typedef float f32x4_ __attribute__((ext_vector_type(4)));

constexpr int DR = 16;            // rows per block
constexpr int NW = 8;             // waves per block (2 per SIMD: one wave's weight-load latency hides under the other's MFMAs)
constexpr int NT = NW * 64;

__device__ __forceinline__ float dsig(float x) { return 1.f / (1.f + __expf(-x)); }
__device__ __forceinline__ float dinv_sig(float x) {
  x = fminf(fmaxf(x, 0.f), 1.f);
  return __logf(fmaxf(x, 1e-5f) / fmaxf(1.f - x, 1e-5f));
}

// Ys[16][N] = act(Xs[16][K] @ W^T + b (+ Rs)).   Xs / Ys / Rs live in LDS; K % 64 == 0 (zero padded).
// The 4 waves take the N/16 column tiles round-robin; a lane's A fragment is Xs[row = lane & 15][k0 + 4 (lane >> 4) .. +3],
// its B fragment W[n0 + (lane & 15)][same k] - each 16-byte load feeds 4 MFMAs (the k order inside a
// step only has to agree between A and B).  The weights are stored FRAGMENT-MAJOR (engine.hip pack_fragments):
// block (tile t, 16-wide k chunk c) is 64 lanes x 16 bytes in lane order, so one wave load is one contiguous
// 1 KiB read.  (Row-major weights put adjacent lanes on different rows: 4x the cache-line requests, and the
// layer ran 7x slower than its MFMA time.)
//
// `rot` (the block's row-tile index inside its image) rotates the order in which the block walks its column-tile passes and
// its K steps.  All blocks of a launch run the same GEMM at the same time and an XCD's blocks share one L2: in lockstep they
// all wait for the same L2 miss on every step (one 64 KiB round of unique bytes in flight per XCD); rotated, the XCD's blocks
// request the whole filter in their first round and later rounds hit in L2.  The fp32 summation order of a row depends only on
// its tile index, not on the batch size (batch invariance holds bit for bit).
template <int ACT>
__device__ void row_gemm(const float* Xs, int ldx, const DecLin& L, float* Ys, int ldy, const float* Rs, int ldr, int wave, int lane, int rot) {
  const int ntiles = (L.N + 15) >> 4;
  const int r16 = lane & 15, q = lane >> 4;
  const float* xrow = Xs + r16 * ldx + 4 * q;
  const int nsteps = L.K >> 6;
  const int npass = (ntiles - wave + 2 * NW - 1) / (2 * NW);   // passes of this wave (wave < ntiles for every layer here)
  // two column tiles (t, t+NW) per pass share the A fragment and give the MFMA pipe two independent accumulators
  for (int ps = 0; ps < npass; ++ps) {
    const int t = wave + 2 * NW * ((ps + rot) % npass);
    const bool has2 = t + NW < ntiles;                          // wave-uniform
    const int n0 = t << 4, n1 = has2 ? (t + NW) << 4 : n0;
    const int kc = L.K >> 4;                                   // 16-wide chunks per tile
    const float* w0 = L.w + (size_t)t * kc * 256 + lane * 4;
    const float* w1 = L.w + (size_t)(has2 ? t + NW : t) * kc * 256 + lane * 4;
    f32x4_ acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    f32x4_ c0[4], c1[4], p0[4], p1[4];
    // every load is unconditional (K % 64 == 0; the prefetch after the last step re-reads a valid step):
    // a per-element "load or zero" select makes hipcc branch and drain vmcnt(0) around each load.
    int st = (rot + ps) % nsteps;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      c0[j] = *(const f32x4_*)(w0 + (st << 2) * 256 + 256 * j);
      c1[j] = *(const f32x4_*)(w1 + (st << 2) * 256 + 256 * j);
    }
    for (int it = 0; it < nsteps; ++it) {
      const int k0 = st << 6;
      st = st + 1 == nsteps ? 0 : st + 1;
      const int kn = st << 6;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        p0[j] = *(const f32x4_*)(w0 + (kn >> 4) * 256 + 256 * j);
        p1[j] = *(const f32x4_*)(w1 + (kn >> 4) * 256 + 256 * j);
      }
      // pin the order: the next chunk's 8 loads are in flight under this chunk's 32 MFMAs (left alone, hipcc
      // sinks each load to just before its use and the L2 latency is exposed every 4 MFMAs)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4_ x4 = *(const f32x4_*)(xrow + k0 + 16 * j);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x4[u], c0[j][u], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x4[u], c1[j][u], acc1, 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j) { c0[j] = p0[j]; c1[j] = p1[j]; }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (h == 1 && !has2) break;
      const int col = (h == 0 ? n0 : n1) + r16;
      if (col < L.N) {
        const float bv = L.b[col];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = q * 4 + r;
          float v = (h == 0 ? acc0[r] : acc1[r]) + bv;
          if (Rs) v += Rs[row * ldr + col];
          if (ACT == ACT_RELU) v = fmaxf(v, 0.f);
          if (ACT == ACT_GELU) v = 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
          Ys[row * ldy + col] = v;
        }
      }
    }
  }
}

// fp32 rows in LDS -> fp16 hi / lo rows in LDS (x = hi + lo to 2^-22 relative / 2^-25 absolute, common.h sp16), K % 8 == 0
__device__ __forceinline__ void split_rows(const float* Xs, int ldx, int K, sp16* Xh, sp16* Xl, int ldb, int tid) {
  const int k8 = K >> 3;
  for (int e = tid; e < DR * k8; e += NT) {
    const int r = e / k8, c = (e - r * k8) << 3;
    const f32x4_ v0 = *(const f32x4_*)(Xs + r * ldx + c), v1 = *(const f32x4_*)(Xs + r * ldx + c + 4);
    sp16x8 h, l;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float x = j < 4 ? v0[j] : v1[j - 4];
      const sp16 hi = (sp16)x;
      h[j] = hi;
      l[j] = (sp16)(x - (float)hi);
    }
    *(sp16x8*)(Xh + r * ldb + c) = h;
    *(sp16x8*)(Xl + r * ldb + c) = l;
  }
}

// The split form of row_gemm: A = (Ah + Al) [16][K] bf16 in LDS, W = hi + lo bf16 fragments (DecLin, split layout), three
// v_mfma_f32_16x16x32_f16 per 32-deep chunk (lo*hi, hi*lo, hi*hi; fp32 accumulate) instead of 8 v_mfma_f32_16x16x4_f32:
// 5x less MFMA time for the same filter bytes, products exact to ~2^-17.  A lane's fragments: A[row = lane & 15][32c + 8 (lane >> 4) .. +7],
// W[n0 + (lane & 15)][same k].  Output: fp32 Ys and / or a hi/lo split (Yh, Yl) for a following GEMM.
template <int ACT, bool WLO = true>   // WLO = false: the filter's lo half is neither loaded nor multiplied (hi-only filter, split activations)
__device__ void row_gemm_split(const sp16* Ah, const sp16* Al, int lda, const DecLin& L, float* Ys, int ldy, const float* Rs, int ldr,
                               sp16* Yh, sp16* Yl, int ldyb, int wave, int lane, int rot, int probe = 0) {
  constexpr int PF = 2;   // K steps of filter fragments in flight per wave (round 2: 3 changed nothing; round 4: 4 needs 128 VGPRs of fragments - 256 + 704 B of scratch, 140 us per layer)
  const int ntiles = (L.N + 15) >> 4;
  const int r16 = lane & 15, q = lane >> 4;
  const sp16* ah = Ah + r16 * lda + 8 * q;
  const sp16* al = Al + r16 * lda + 8 * q;
  const int kc = L.K >> 5;                                     // 32-deep chunks per tile, 2 KiB each (hi 1 KiB | lo 1 KiB)
  const int nsteps = L.K >> 6;
  const int npass = (ntiles - wave + 2 * NW - 1) / (2 * NW);
  for (int ps = 0; ps < npass; ++ps) {                          // pass / K-step rotation: see row_gemm
    const int t = wave + 2 * NW * ((ps + rot) % npass);
    const bool has2 = t + NW < ntiles;                          // wave-uniform
    const int n0 = t << 4, n1 = has2 ? (t + NW) << 4 : n0;
    const char* w0 = (const char*)L.w + (size_t)t * kc * 2048 + lane * 16;
    const char* w1 = (const char*)L.w + (size_t)(has2 ? t + NW : t) * kc * 2048 + lane * 16;
    f32x4_ acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    // the pass's two bias values are requested with its first filter fragments (read in the epilogue they cost an exposed L2 round trip per pass)
    const float bias0 = (n0 + r16 < L.N) ? L.b[n0 + r16] : 0.f, bias1 = (n1 + r16 < L.N) ? L.b[n1 + r16] : 0.f;
    asm volatile("" ::: "memory");                              // (keeps the two loads up here: hipcc sinks a load to its use otherwise)
    // PF register sets, each one 64-deep K step of both tiles (8 x 16-byte loads per lane); a set is reloaded with the step PF
    // ahead right after its MFMAs were issued, so 8 PF loads per wave stay in flight (the loop is L2-latency bound: one step in
    // flight per wave was 0.9 us per step).
    sp16x8 wh0[PF][2], wl0[PF][2], wh1[PF][2], wl1[PF][2];
    int st = (rot + ps) % nsteps;
    auto nxt = [&](int v) { return v + 1 == nsteps ? 0 : v + 1; };
    auto load = [&](sp16x8 (&h0)[2], sp16x8 (&l0)[2], sp16x8 (&h1)[2], sp16x8 (&l1)[2], int step) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        h0[j] = *(const sp16x8*)(w0 + (step << 1) * 2048 + 2048 * j);
        if (WLO) l0[j] = *(const sp16x8*)(w0 + (step << 1) * 2048 + 2048 * j + 1024);
        h1[j] = *(const sp16x8*)(w1 + (step << 1) * 2048 + 2048 * j);
        if (WLO) l1[j] = *(const sp16x8*)(w1 + (step << 1) * 2048 + 2048 * j + 1024);
      }
    };
    auto mma = [&](const sp16x8 (&h0)[2], const sp16x8 (&l0)[2], const sp16x8 (&h1)[2], const sp16x8 (&l1)[2], int step) {
      const int k0 = step << 6;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const sp16x8 xh = *(const sp16x8*)(ah + k0 + 32 * j), xl = *(const sp16x8*)(al + k0 + 32 * j);
        acc0 = mfma_pair16(xl, h0[j], acc0);
        acc1 = mfma_pair16(xl, h1[j], acc1);
        if (WLO) {
          acc0 = mfma_pair16(xh, l0[j], acc0);
          acc1 = mfma_pair16(xh, l1[j], acc1);
        }
        acc0 = mfma_pair16(xh, h0[j], acc0);
        acc1 = mfma_pair16(xh, h1[j], acc1);
      }
    };
    int cur = st, pf = st;                                       // step of the next MFMA group / of the next load
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      if (u < nsteps) { load(wh0[u], wl0[u], wh1[u], wl1[u], pf); pf = nxt(pf); }
    }
    for (int it = 0; it < nsteps; it += PF) {                    // (wave-uniform guards: no load is issued past the last step)
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        if (it + u < nsteps) {
          __builtin_amdgcn_sched_barrier(0);
          if (!(probe & 1)) mma(wh0[u], wl0[u], wh1[u], wl1[u], cur);
          else { acc0[0] += (float)wh0[u][1][7] + (float)wl1[u][1][7]; acc1[0] += (float)wh1[u][1][7] + (float)wl0[u][1][7]; }   // probe: waits for the step's loads, no MFMA
          cur = nxt(cur);
          __builtin_amdgcn_sched_barrier(0);
          if (it + u + PF < nsteps && !(probe & 2)) { load(wh0[u], wl0[u], wh1[u], wl1[u], pf); pf = nxt(pf); }
        }
      }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (h == 1 && !has2) break;
      const int col = (h == 0 ? n0 : n1) + r16;
      if (col < L.N) {
        const float bv = h == 0 ? bias0 : bias1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = q * 4 + r;
          float v = (h == 0 ? acc0[r] : acc1[r]) + bv;
          if (Rs) v += Rs[row * ldr + col];
          if (ACT == ACT_RELU) v = fmaxf(v, 0.f);
          if (ACT == ACT_GELU) v = 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
          if (Ys) Ys[row * ldy + col] = v;
          if (Yh) {
            const sp16 hi = (sp16)v;
            Yh[row * ldyb + col] = hi;
            Yl[row * ldyb + col] = (sp16)(v - (float)hi);
          }
        }
      }
    }
  }
}

// in-place LayerNorm of Xs[16][D] (D <= 256, D % 4 == 0); each wave owns DR / NW rows.  The affine parameters come in registers: the caller
// requests them (ln_fetch) BEFORE the GEMM that produces the rows, so their L2 round trip (~1.5 us of this 2.5 us phase when it was issued
// here, tools/dec_stamps.py) runs under that GEMM.  Same arithmetic, same order: bit-identical rows.
struct LNRegs { float g[4], b[4]; };
__device__ __forceinline__ LNRegs ln_fetch(const DecLN& P, int D, int lane) {
  LNRegs R;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane + 64 * i;
    R.g[i] = c < D ? P.g[c] : 0.f;
    R.b[i] = c < D ? P.b[c] : 0.f;
  }
  return R;
}
__device__ __forceinline__ void row_ln(float* Xs, int ldx, int D, const LNRegs& P, int wave, int lane) {
  constexpr int RPW = DR / NW;                                  // rows per wave: their reductions are independent and interleave
  float v[RPW][4], s[RPW];
#pragma unroll
  for (int j = 0; j < RPW; ++j) {
    const int r = wave * RPW + j;
    s[j] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = lane + 64 * i;
      v[j][i] = c < D ? Xs[r * ldx + c] : 0.f;
      s[j] += v[j][i];
    }
  }
  float mean[RPW], sq[RPW];
#pragma unroll
  for (int j = 0; j < RPW; ++j) mean[j] = wave_sum64(s[j]) / (float)D;
#pragma unroll
  for (int j = 0; j < RPW; ++j) {
    sq[j] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = lane + 64 * i;
      const float d = c < D ? v[j][i] - mean[j] : 0.f;
      sq[j] += d * d;
    }
  }
#pragma unroll
  for (int j = 0; j < RPW; ++j) {
    const int r = wave * RPW + j;
    const float rstd = rsqrtf(wave_sum64(sq[j]) / (float)D + 1e-5f);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = lane + 64 * i;
      if (c < D) Xs[r * ldx + c] = (v[j][i] - mean[j]) * rstd * P.g[i] + P.b[i];
    }
  }
}


// MS-deformable sampling of the block's 16 rows x 8 heads = 128 (row, head) items; 3 levels x 4 points.
// FOUR lanes own one item, each 8 of its 32 channels (one 16-byte load per tap), so the 256 threads cover 64
// items per pass (2 passes).  All 16 bilinear taps of a level are loaded UNCONDITIONALLY (clamped address, zero
// weight when the tap is outside the map = grid_sample's zero padding) so they are in flight together: this phase
// is pure HBM/MALL latency (random 64-byte rows of a ~200 MB tensor) and was 50 % of the kernel with one
// 2-byte load per lane and 8 items per pass.
template <typename TV>
__device__ void sample_rows(const DecArgs& a, const float* sO, int LDO, const float* sR, int LDR, float* sA, int LDH, int b,
                            int nvalid, int tid) {
  constexpr int NL = 3, NP = 4, LP = NL * NP;
  constexpr int V = 16 / (int)sizeof(TV);                      // channels per 16-byte load: 8 (bf16) / 4 (fp32)
  typedef TV VT __attribute__((ext_vector_type(V)));
  const int c8 = tid & 3;
  const float pscale = 1.f / (float)NP;
  for (int item = tid >> 2; item < DR * a.heads; item += NT / 4) {
    const int r = item / a.heads, head = item - r * a.heads;
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
    if (r < nvalid) {
      const float* offs = sO + r * LDO + head * LP * 2;
      const float* awl = sO + r * LDO + a.heads * LP * 2 + head * LP;
      float lg[LP];
      float mx = -INFINITY;
#pragma unroll
      for (int i = 0; i < LP; ++i) { lg[i] = awl[i]; mx = fmaxf(mx, lg[i]); }
      float den = 0.f;
#pragma unroll
      for (int i = 0; i < LP; ++i) { lg[i] = __expf(lg[i] - mx); den += lg[i]; }
      const float inv_den = 1.f / den;
      const float rx = sR[r * LDR + 0], ry = sR[r * LDR + 1], rw = sR[r * LDR + 2], rh = sR[r * LDR + 3];
      const TV* vb = (const TV*)a.value + (long long)b * a.S * a.value_ld + a.value_coff + head * 32 + c8 * 8;
#pragma unroll
      for (int l = 0; l < NL; ++l) {
        const int H = a.lvl[l * 3 + 0], W = a.lvl[l * 3 + 1], start = a.lvl[l * 3 + 2];
        float wt[NP * 4];
        int of[NP * 4];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          const int i = l * NP + p;
          const float aw = lg[i] * inv_den;
          const float lx = rx + offs[i * 2 + 0] * pscale * rw * a.offset_scale;
          const float ly = ry + offs[i * 2 + 1] * pscale * rh * a.offset_scale;
          const float gx = 2.f * lx - 1.f, gy = 2.f * ly - 1.f;
          const float ix = ((gx + 1.f) * (float)W - 1.f) * 0.5f;
          const float iy = ((gy + 1.f) * (float)H - 1.f) * 0.5f;
          const float fx = floorf(ix), fy = floorf(iy);
          // clamp before the int conversion: masked-anchor boxes (ref = 1) with large offsets can be far outside
          const int x0 = (int)fminf(fmaxf(fx, -2.f), (float)W + 1.f), y0 = (int)fminf(fmaxf(fy, -2.f), (float)H + 1.f);
          const int x1 = x0 + 1, y1 = y0 + 1;
          const float wx1 = ix - fx, wy1 = iy - fy, wx0 = (fx + 1.f) - ix, wy0 = (fy + 1.f) - iy;
          const bool okx0 = (unsigned)x0 < (unsigned)W, okx1 = (unsigned)x1 < (unsigned)W;
          const bool oky0 = (unsigned)y0 < (unsigned)H, oky1 = (unsigned)y1 < (unsigned)H;
          const int cx0 = min(max(x0, 0), W - 1), cx1 = min(max(x1, 0), W - 1);
          const int cy0 = min(max(y0, 0), H - 1), cy1 = min(max(y1, 0), H - 1);
          wt[p * 4 + 0] = (oky0 && okx0) ? wx0 * wy0 * aw : 0.f;
          wt[p * 4 + 1] = (oky0 && okx1) ? wx1 * wy0 * aw : 0.f;
          wt[p * 4 + 2] = (oky1 && okx0) ? wx0 * wy1 * aw : 0.f;
          wt[p * 4 + 3] = (oky1 && okx1) ? wx1 * wy1 * aw : 0.f;
          of[p * 4 + 0] = (start + cy0 * W + cx0) * a.value_ld;
          of[p * 4 + 1] = (start + cy0 * W + cx1) * a.value_ld;
          of[p * 4 + 2] = (start + cy1 * W + cx0) * a.value_ld;
          of[p * 4 + 3] = (start + cy1 * W + cx1) * a.value_ld;
        }
        VT vv[NP * 4][8 / V];
#pragma unroll
        for (int t = 0; t < NP * 4; ++t)
#pragma unroll
          for (int u = 0; u < 8 / V; ++u) vv[t][u] = *(const VT*)(vb + of[t] + u * V);
#pragma unroll
        for (int t = 0; t < NP * 4; ++t)
#pragma unroll
          for (int k = 0; k < 8; ++k) acc[k] = fmaf((float)vv[t][k / V][k % V], wt[t], acc[k]);
      }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) sA[r * LDH + head * 32 + c8 * 8 + k] = acc[k];
  }
}


// Self-attention of the block's 16 query rows (HF:v2.py:246-336, no mask), one head per wave, exact fp32 MFMA.
// Computed TRANSPOSED so nothing crosses LDS: per 16-key tile  S^T = K Q^T  (keys on accumulator rows, the
// lane's query on the column), so the softmax reductions over keys are 4 registers + two xor-shuffles (16, 32),
// and the accumulator tile of P^T is already the B-fragment of the next product  O^T += V^T P^T
// (MI355X guide: "an accumulator tile as the next MFMA's operand").  K and V were written by the previous launch
// in fragment order, so each operand fetch is one contiguous 1 KiB wave load.
__device__ void self_attention_rows(const DecArgs& a, const float* sQ, int ldq, float* sA, int LDH, int b, int tiles, int wave, int lane) {
  const int head = wave;                                        // NW == heads
  const int r16 = lane & 15, q = lane >> 4;
  const float scale = rsqrtf(32.f);
  f32x4_ qa[2];                                                 // Q[query r16][32 head + 16 c + 4 q + u] * scale
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    qa[c] = *(const f32x4_*)(sQ + r16 * ldq + head * 32 + 16 * c + 4 * q);
#pragma unroll
    for (int u = 0; u < 4; ++u) qa[c][u] *= scale;
  }
  float m = -INFINITY, l = 0.f;                                 // running max / sum of this lane's query
  f32x4_ O[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};   // O^T: dims 16 d + 4 q + r of query r16
  const int tp = (tiles + 1) & ~1;                              // fragment buffers are strided by an even tile count (the split form stores V per tile PAIR)
  const float* kb = a.kfrag_in + ((size_t)(b * a.heads + head) * tp) * 512 + lane * 4;
  const float* vb = a.vfrag_in + ((size_t)(b * a.heads + head) * tp) * 512 + lane * 4;
  f32x4_ kc[2], vc[2], kn[2], vn[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) { kc[c] = *(const f32x4_*)(kb + c * 256); vc[c] = *(const f32x4_*)(vb + c * 256); }
  for (int t = 0; t < tiles; ++t) {
    const int tn = min(t + 1, tiles - 1);                       // unconditional prefetch (clamped)
#pragma unroll
    for (int c = 0; c < 2; ++c) { kn[c] = *(const f32x4_*)(kb + (size_t)tn * 512 + c * 256); vn[c] = *(const f32x4_*)(vb + (size_t)tn * 512 + c * 256); }
    __builtin_amdgcn_sched_barrier(0);
    f32x4_ S = {0.f, 0.f, 0.f, 0.f};                            // S^T[key 4 q + r][query r16]
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int u = 0; u < 4; ++u) S = __builtin_amdgcn_mfma_f32_16x16x4f32(kc[c][u], qa[c][u], S, 0, 0, 0);
    float sv[4];
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      sv[r] = (t * 16 + 4 * q + r < a.Q) ? S[r] : -INFINITY;
      mx = fmaxf(mx, sv[r]);
    }
    mx = rows4_max(mx);
    const float mn = fmaxf(m, mx);
    const float alpha = __expf(m - mn);
    f32x4_ p;
    float rs = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) { p[r] = __expf(sv[r] - mn); rs += p[r]; }
    rs = rows4_sum(rs);
    l = l * alpha + rs;
    m = mn;
#pragma unroll
    for (int d = 0; d < 2; ++d) {
#pragma unroll
      for (int r = 0; r < 4; ++r) O[d][r] *= alpha;
#pragma unroll
      for (int u = 0; u < 4; ++u) O[d] = __builtin_amdgcn_mfma_f32_16x16x4f32(vc[d][u], p[u], O[d], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < 2; ++c) { kc[c] = kn[c]; vc[c] = vn[c]; }
  }
  const float inv = 1.f / l;
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    f32x4_ o = {O[d][0] * inv, O[d][1] * inv, O[d][2] * inv, O[d][3] * inv};
    *(f32x4_*)(sA + r16 * LDH + head * 32 + 16 * d + 4 * q) = o;
  }
}

// The pair form (bf16 and f16x3 engines): the same transposed scheme on v_mfma_f32_16x16x32_f16 with hi/lo fp16 splits of every operand (x = hi + lo to
// 2^-22: hi*hi + hi*lo + lo*hi, fp32 accumulate), 12 16-bit MFMAs of 16 cycles per PAIR of key tiles instead of 32 fp32 MFMAs of 32 - the
// phase is MFMA-issue bound (two waves per SIMD; prefetch depth and tile-level parallelism changed nothing).
//   S^T_e = K_e Q^T           : one MFMA covers the whole head dim (32); K fragments [hi | lo][lane][8]: key lane & 15, dims 8 (lane >> 4) ..
//   O^T  += V_pair^T P_pair^T : the 32-deep contraction runs over the pair's keys in the order (tile e, key 4 kq + r) -> position
//                               8 kq + 4 e + r, which is exactly how a lane's S accumulators of the two tiles are laid out (no shuffle);
//                               V fragments [d][hi | lo][lane][8] per pair, written half by each tile's producer block.
__device__ void self_attention_rows_split(const DecArgs& a, const float* sQ, int ldq, float* sA, int LDH, int b, int tiles, int wave, int lane) {
  const int head = wave;
  const int r16 = lane & 15, q = lane >> 4;
  const float scale = rsqrtf(32.f);
  sp16x8 qh, ql;                                               // Q[query r16][32 head + 8 q + j] * scale
  {
    const f32x4_ q0 = *(const f32x4_*)(sQ + r16 * ldq + head * 32 + 8 * q), q1 = *(const f32x4_*)(sQ + r16 * ldq + head * 32 + 8 * q + 4);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float x = (j < 4 ? q0[j] : q1[j - 4]) * scale;
      const sp16 hi = (sp16)x;
      qh[j] = hi;
      ql[j] = (sp16)(x - (float)hi);
    }
  }
  float m = -INFINITY, l = 0.f;
  f32x4_ O[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  const int tp = (tiles + 1) & ~1, npairs = tp >> 1;
  const char* kb = (const char*)a.kfrag_in + ((size_t)(b * a.heads + head) * tp) * 2048 + lane * 16;
  const char* vb = (const char*)a.vfrag_in + ((size_t)(b * a.heads + head) * tp) * 2048 + lane * 16;
  sp16x8 kc[2][2], vc[2][2], kn[2][2], vn[2][2];               // K: [tile of the pair][hi, lo]; V: [d][hi, lo]
  auto ld = [&](sp16x8 (&k)[2][2], sp16x8 (&v)[2][2], int pr) {
    const int t1 = min(2 * pr + 1, tiles - 1);                  // odd tile count: the second K of the last pair re-reads the last tile (masked)
#pragma unroll
    for (int hl = 0; hl < 2; ++hl) {
      k[0][hl] = *(const sp16x8*)(kb + (size_t)(2 * pr) * 2048 + hl * 1024);
      k[1][hl] = *(const sp16x8*)(kb + (size_t)t1 * 2048 + hl * 1024);
#pragma unroll
      for (int d = 0; d < 2; ++d) v[d][hl] = *(const sp16x8*)(vb + (size_t)pr * 4096 + (d * 2 + hl) * 1024);
    }
  };
  ld(kc, vc, 0);
  for (int pr = 0; pr < npairs; ++pr) {
    ld(kn, vn, min(pr + 1, npairs - 1));
    __builtin_amdgcn_sched_barrier(0);
    f32x4_ S[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      S[e] = mfma_pair16(kc[e][1], qh, S[e]);
      S[e] = mfma_pair16(kc[e][0], ql, S[e]);
      S[e] = mfma_pair16(kc[e][0], qh, S[e]);
    }
    float sv[2][4];
    float mx = -INFINITY;
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int t = 2 * pr + e;
        sv[e][r] = (t < tiles && t * 16 + 4 * q + r < a.Q) ? S[e][r] : -INFINITY;
        mx = fmaxf(mx, sv[e][r]);
      }
    mx = rows4_max(mx);
    const float mn = fmaxf(m, mx);
    const float alpha = __expf(m - mn);
    sp16x8 ph, pl;
    float rs = 0.f;
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pv = __expf(sv[e][r] - mn);
        rs += pv;
        const sp16 hi = (sp16)pv;
        ph[4 * e + r] = hi;
        pl[4 * e + r] = (sp16)(pv - (float)hi);
      }
    rs = rows4_sum(rs);
    l = l * alpha + rs;
    m = mn;
#pragma unroll
    for (int d = 0; d < 2; ++d) {
#pragma unroll
      for (int r = 0; r < 4; ++r) O[d][r] *= alpha;
      O[d] = mfma_pair16(vc[d][1], ph, O[d]);
      O[d] = mfma_pair16(vc[d][0], pl, O[d]);
      O[d] = mfma_pair16(vc[d][0], ph, O[d]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int hl = 0; hl < 2; ++hl) { kc[e][hl] = kn[e][hl]; vc[e][hl] = vn[e][hl]; }
  }
  const float inv = 1.f / l;
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    f32x4_ o = {O[d][0] * inv, O[d][1] * inv, O[d][2] * inv, O[d][3] * inv};
    *(f32x4_*)(sA + r16 * LDH + head * 32 + 16 * d + 4 * q) = o;
  }
}

// diagnostic phase stamps (100 MHz wall clock), enabled only when a.stamps != nullptr (rtd_debug_option "dec_stamps")
#define DEC_STAMP(i)                                                                         \
  do {                                                                                       \
    if (a.stamps && tid == 0) a.stamps[(size_t)blockIdx.x * 16 + (i)] = (float)(long long)(__builtin_amdgcn_s_memrealtime() - t_start); \
  } while (0)

typedef __attribute__((address_space(3))) void* dec_lds_ptr_t;
// Ask L2 for a later GEMM's filter: one dword per 128-byte line, this block's 1/nparts share, LDS-DMA into a dummy (no VGPR,
// nothing waits).  The blocks of an XCD together request the whole filter one phase before they stream it, so the stream
// hits in L2 instead of every block waiting on the same misses (tools/dec_stamps.py: 137 -> 115 us per layer came from
// de-synchronising the blocks; this removes the remaining first-touch misses).
__device__ __forceinline__ void touch_weights(const DecLin& L, int part, int nparts, int tid, char* dummy) {
  if (!L.w) return;
  const unsigned bytes = (unsigned)((L.N + 15) >> 4) * (unsigned)L.K * 64u;
  const unsigned lines = bytes >> 7;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)L.w, 0, bytes, 0x00020000);
  for (unsigned l = part + nparts * tid; l < lines; l += nparts * NT) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (dec_lds_ptr_t)dummy, 4, l << 7, 0, 0, 0);
}

// SPLIT: the linear layers take hi/lo fp16 splits of their operands (row_gemm_split); every fp32 A operand of K <= 256 is split
// into the sXh/sXl staging rows right before its GEMM, the two wide ones (FFN hidden 1024, qpos hidden 512) are written as
// splits by the producing GEMM straight into the sF region (same bytes as the fp32 rows they replace).
// (The phase functions above are plain __device__ functions that hipcc inlines here.  When the inliner's budget runs out - any experiment that
// grows this kernel - it leaves one of them as a CALL, the by-reference DecArgs must then live in memory, and the whole 704-byte argument
// struct is copied to scratch at kernel entry: round 4 read that as register spilling ("256 registers + 704 B of scratch"); it is an
// inlining artefact, and __forceinline__ on the phase functions removes it.  They are NOT force-inlined in this build: with the attribute
// hipcc schedules the same code 2.5 % slower (0.770 vs 0.747 ms per step for the nine launches, same box, round 5).)
template <int SPLIT>
__global__ __launch_bounds__(NT, 1) void dec_layer_kernel(const DecArgs a) {
  // LDS (floats).  Row strides are (cols + 4): ds_read_b128 of 16 rows x 4 k-groups is conflict-free.
  constexpr int LDH = 260, LDF = 1028, LDQ = 516, LDO = 292, LDR = 68;
  constexpr int LDX = 264, LDFB = 1032, LDQB = 520;             // sp16 row strides (16-byte rows, 4-bank skew per row)
  __shared__ __attribute__((aligned(16))) float sH[DR * LDH];   // hs / running activation x
  __shared__ __attribute__((aligned(16))) float sP[DR * LDH];   // query_pos of this layer, later of the next
  __shared__ __attribute__((aligned(16))) float sA[DR * LDH];   // attention out / sampled values / scratch
  __shared__ __attribute__((aligned(16))) float sF[DR * LDF + 64];   // FFN hidden (1024); SPLIT: sp16 hi rows | lo rows
  __shared__ __attribute__((aligned(16))) sp16 sXh[SPLIT ? DR * LDX : 8], sXl[SPLIT ? DR * LDX : 8];
  __shared__ __attribute__((aligned(16))) char sDummy[256];     // touch_weights target
  float* const sT = sF;                                         // 512-wide scratch (bbox hidden, qpos hidden, q|k): live only while sF is dead
  sp16* const sFh = (sp16*)sF;
  sp16* const sFl = sFh + DR * LDFB;
  sp16* const sQh = (sp16*)sF;                                  // qpos hidden (512) as a split, live between qp0 and qp1
  sp16* const sQl = sQh + DR * LDQB;
  static_assert(2 * DR * LDFB * 2 <= (DR * LDF + 64) * 4, "split FFN hidden must fit the fp32 region");
  __shared__ __attribute__((aligned(16))) float sO[DR * LDO];   // sampling offsets | attention logits
  __shared__ __attribute__((aligned(16))) float sR[DR * LDR];   // ref boxes (cols 0..3), zero padded to 64 (K of qpos.0)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // Y[16][N] = act(X[16][K] W^T + b (+ R)); X fp32 rows in LDS.  SPLIT: X is first split into sXh/sXl (all waves must have
  // left the previous GEMM: every call site below sits behind a __syncthreads()).
#define DEC_TOUCH(LW) touch_weights(LW, rot, tiles, tid, sDummy)
#define DEC_GEMM(ACT, X, LDXS, LW, Y, LDY, R, LDRS, NEXT)                                                   \
  do {                                                                                                       \
    if (SPLIT) {                                                                                             \
      split_rows(X, LDXS, (LW).K, sXh, sXl, LDX, tid);                                                       \
      __syncthreads();                                                                                       \
      row_gemm_split<ACT, (SPLIT != 2) >(sXh, sXl, LDX, LW, Y, LDY, R, LDRS, nullptr, nullptr, 0, wave, lane, rot, a.probe);         \
    } else {                                                                                                 \
      row_gemm<ACT>(X, LDXS, LW, Y, LDY, R, LDRS, wave, lane, rot);                                          \
    }                                                                                                        \
    DEC_TOUCH(NEXT);                                                                                         \
  } while (0)
  const unsigned long long t_start = a.stamps ? __builtin_amdgcn_s_memrealtime() : 0ull;
  const int tiles = (a.Q + DR - 1) / DR;
  const int b = blockIdx.x / tiles;
  const int q0 = (blockIdx.x - b * tiles) * DR;
  const int rot = q0 / DR;                                       // GEMM pass / K-step rotation (row_gemm)
  const DecLin no_next = {nullptr, nullptr, 0, 0, 0};
  const int D = a.D;
  const long long row0 = (long long)b * a.Q + q0;
  const int nvalid = min(DR, a.Q - q0);
  // modes: 0 decoder prologue | 1 decoder layer | 2 last decoder layer | 3 AIFI prologue (x + pos -> q,k,v) |
  //        4 AIFI layer (self-attention, o-proj + LN, GELU FFN + LN; HF:v2.py:838-904)
  const bool has_attn = a.mode == 1 || a.mode == 2 || a.mode == 4;
  const bool has_cross = a.mode == 1 || a.mode == 2;

  // ---- load the block's rows (16-byte loads, every load of a thread issued before the first use: the phase is pure latency) --
  {
    constexpr int PER = DR * 256 / 4 / NT;                       // float4 per thread and array (D == 256)
    f32x4_ vh[PER], vq[PER], vp[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int e = (tid + i * NT) * 4, r = e >> 8, c = e & 255;
      const int rr = min(r, nvalid - 1);                          // rows past the end re-read a valid row; zeroed below
      vh[i] = *(const f32x4_*)(a.hs_in + (row0 + rr) * D + c);
      if (has_attn) vq[i] = *(const f32x4_*)(a.q_in + (row0 + rr) * D + c);
      if (has_cross) vp[i] = *(const f32x4_*)(a.qpos_in + (row0 + rr) * D + c);
      if (a.mode == 3) vp[i] = *(const f32x4_*)(a.qpos_in + (long long)(q0 + rr) * D + c);   // AIFI: one sin-cos table for every image
    }
    // first filters of this launch (the later ones are requested one GEMM ahead, DEC_GEMM)
    if (has_attn) { DEC_TOUCH(a.o); DEC_TOUCH(a.fc1); }
    if (has_cross) DEC_TOUCH(a.offaw);
    if (a.mode == 4) DEC_TOUCH(a.fc2);
    if (a.mode == 0) { DEC_TOUCH(a.bb0); DEC_TOUCH(a.bb1); }
    if (a.mode == 3) { DEC_TOUCH(a.qk); DEC_TOUCH(a.v); }
    const f32x4_ z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int e = (tid + i * NT) * 4, r = e >> 8, c = e & 255;
      const bool ok = r < nvalid;
      *(f32x4_*)(sH + r * LDH + c) = ok ? vh[i] : z4;
      if (has_attn) *(f32x4_*)(sT + r * LDQ + c) = ok ? vq[i] : z4;
      if (has_cross || a.mode == 3) *(f32x4_*)(sP + r * LDH + c) = ok ? vp[i] : z4;
    }
  }
  for (int e = tid; e < DR * LDR; e += NT) {
    const int r = e / LDR, c = e - r * LDR;
    sR[e] = (has_cross && r < nvalid && c < 4) ? a.ref8[(row0 + r) * 8 + c] : 0.f;
  }
  __syncthreads();
  DEC_STAMP(0);   // rows loaded

  if (has_attn) {
    // ---- self-attention (q in sT, K/V fragments from the previous launch) -> sA ----------------------------
    for (int head = wave; head < a.heads; head += NW) {          // one head per wave and turn (NW == heads: one turn)
      if (SPLIT && a.attn_split) self_attention_rows_split(a, sT, LDQ, sA, LDH, b, tiles, head, lane);
      else self_attention_rows(a, sT, LDQ, sA, LDH, b, tiles, head, lane);
    }
    __syncthreads();
    DEC_STAMP(12);  // self-attention
    // ---- self-attention output projection + residual + LN1 (HF:v2.py:395-405) ---------------------
    const LNRegs ln1 = ln_fetch(a.ln1, D, lane);
    DEC_GEMM(ACT_NONE, sA, LDH, a.o, sH, LDH, sH, LDH, a.op);      // x = hs + att @ Wo   (in place: each element read then written by one lane)
    __syncthreads();
    DEC_STAMP(1);   // o_proj
    row_ln(sH, LDH, D, ln1, wave, lane);
    __syncthreads();
    DEC_STAMP(2);   // ln1
  }
  if (has_cross) {
    // ---- cross attention: (x + qpos) -> offsets | weights (HF:v2.py:170-186) -------------------------
    for (int e = tid; e < DR * D; e += NT) {
      const int r = e / D, c = e - r * D;
      sA[r * LDH + c] = sH[r * LDH + c] + sP[r * LDH + c];
    }
    __syncthreads();
    DEC_GEMM(ACT_NONE, sA, LDH, a.offaw, sO, LDO, nullptr, 0, a.fc1);
    __syncthreads();
    DEC_STAMP(3);   // add + offaw
    // ---- MS-deformable sampling (HF:v2.py:44-115,203-221): 32 lanes = one (row, head) ---------------
    if (a.value_f32) sample_rows<float>(a, sO, LDO, sR, LDR, sA, LDH, b, nvalid, tid);
    else sample_rows<bf16>(a, sO, LDO, sR, LDR, sA, LDH, b, nvalid, tid);
    __syncthreads();
    DEC_STAMP(4);   // sampling
    // ---- output projection + residual + LN2 (HF:v2.py:221,418-421) ----------------------------------
    const LNRegs ln2 = ln_fetch(a.ln2, D, lane);
    DEC_GEMM(ACT_NONE, sA, LDH, a.op, sH, LDH, sH, LDH, a.fc2);
    __syncthreads();
    row_ln(sH, LDH, D, ln2, wave, lane);
    __syncthreads();
    DEC_STAMP(5);   // op + ln2
  }
  if (has_attn) {
    // ---- FFN + residual + LN3 (HF:v2.py:423-428; AIFI: GELU, :888-896) ---------------------------------
    if (SPLIT) {
      split_rows(sH, LDH, a.fc1.K, sXh, sXl, LDX, tid);
      __syncthreads();
      if (a.mode == 4) row_gemm_split<ACT_GELU, (SPLIT != 2) >(sXh, sXl, LDX, a.fc1, nullptr, 0, nullptr, 0, sFh, sFl, LDFB, wave, lane, rot, a.probe);
      else row_gemm_split<ACT_RELU, (SPLIT != 2) >(sXh, sXl, LDX, a.fc1, nullptr, 0, nullptr, 0, sFh, sFl, LDFB, wave, lane, rot, a.probe);
    } else {
      if (a.mode == 4) row_gemm<ACT_GELU>(sH, LDH, a.fc1, sF, LDF, nullptr, 0, wave, lane, rot);
      else row_gemm<ACT_RELU>(sH, LDH, a.fc1, sF, LDF, nullptr, 0, wave, lane, rot);
    }
    DEC_TOUCH(a.bb0);
    __syncthreads();
    DEC_STAMP(6);   // fc1
    const LNRegs ln3 = ln_fetch(a.ln3, D, lane);
    if (SPLIT) row_gemm_split<ACT_NONE, (SPLIT != 2) >(sFh, sFl, LDFB, a.fc2, sH, LDH, sH, LDH, nullptr, nullptr, 0, wave, lane, rot, a.probe);
    else row_gemm<ACT_NONE>(sF, LDF, a.fc2, sH, LDH, sH, LDH, wave, lane, rot);
    DEC_TOUCH(a.bb1);
    __syncthreads();
    row_ln(sH, LDH, D, ln3, wave, lane);
    __syncthreads();
    DEC_STAMP(7);   // fc2 + ln3
  }
  if (a.mode == 4) {
    // AIFI output tokens in the trunk's storage type (they feed the CCFF convolutions)
    for (int e = tid; e < DR * D; e += NT) {
      const int r = e / D, c = e - r * D;
      if (r < nvalid) {
        if (a.out_bf16) ((bf16*)a.out_bf16)[(row0 + r) * D + c] = (bf16)sH[r * LDH + c];
        else a.hs_out[(row0 + r) * D + c] = sH[r * LDH + c];
      }
    }
    return;
  }

  if (a.mode != 3) {
  // ---- box head: mode 0 = enc_bbox_head(target) + anchors (HF:v2.py:1588-1599), else bbox_embed[i] + logit(ref) (:636-639)
  DEC_GEMM(ACT_RELU, sH, LDH, a.bb0, sA, LDH, nullptr, 0, a.qp0);
  DEC_TOUCH(a.bb2);
  __syncthreads();
  DEC_GEMM(ACT_RELU, sA, LDH, a.bb1, sT, LDQ, nullptr, 0, a.qp1);
  DEC_TOUCH(a.cls);
  __syncthreads();
  DEC_GEMM(ACT_NONE, sT, LDQ, a.bb2, sO, LDO, nullptr, 0, a.qk);    // [16][4] deltas in sO cols 0..3
  __syncthreads();
  if (tid < DR * 4) {
    const int r = tid >> 2, c = tid & 3;
    if (r < nvalid) {
      float v;
      if (a.mode == 0) {
        int t = a.tk_idx[row0 + r];
        t = min(max(t, 0), a.S - 1);
        const float u = sO[r * LDO + c] + a.anchors[(long long)t * 4 + c];
        a.ref_unact8[(row0 + r) * 8 + c] = u;
        a.ref_unact8[(row0 + r) * 8 + 4 + c] = 0.f;
        v = dsig(u);
      } else {
        v = dsig(sO[r * LDO + c] + dinv_sig(sR[r * LDR + c]));
      }
      sR[r * LDR + c] = v;
      a.ref8[(row0 + r) * 8 + c] = v;
      a.ref8[(row0 + r) * 8 + 4 + c] = 0.f;
    }
  }
  DEC_STAMP(8);   // bbox head + refine
  // hidden state of this layer
  if (a.mode != 0) {
    for (int e = tid; e < DR * D; e += NT) {
      const int r = e / D, c = e - r * D;
      if (r < nvalid) a.hs_out[(row0 + r) * D + c] = sH[r * LDH + c];
    }
  }
  __syncthreads();
  }   // mode != 3

  if (a.mode == 2) {
    // ---- class head of the last layer (HF:v2.py:644-646,1880) ----------------------------------------
    DEC_GEMM(ACT_NONE, sH, LDH, a.cls, sT, LDQ, nullptr, 0, no_next);
    __syncthreads();
    for (int e = tid; e < DR * a.C; e += NT) {
      const int r = e / a.C, c = e - r * a.C;
      if (r < nvalid) a.logits[(row0 + r) * a.C + c] = sT[r * LDQ + c];
    }
    return;
  }

  // ---- projections for the NEXT layer: qpos = MLP(ref) (HF:v2.py:613), q|k = (hs+qpos) Wqk, v = hs Wv ----
  if (a.mode != 3) {
    if (SPLIT) {
      split_rows(sR, LDR, a.qp0.K, sXh, sXl, LDX, tid);
      __syncthreads();
      row_gemm_split<ACT_RELU, (SPLIT != 2) >(sXh, sXl, LDX, a.qp0, nullptr, 0, nullptr, 0, sQh, sQl, LDQB, wave, lane, rot, a.probe);
      DEC_TOUCH(a.v);
      __syncthreads();
      row_gemm_split<ACT_NONE, (SPLIT != 2) >(sQh, sQl, LDQB, a.qp1, sP, LDH, nullptr, 0, nullptr, nullptr, 0, wave, lane, rot, a.probe);
    } else {
      row_gemm<ACT_RELU>(sR, LDR, a.qp0, sT, LDQ, nullptr, 0, wave, lane, rot);
      DEC_TOUCH(a.v);
      __syncthreads();
      row_gemm<ACT_NONE>(sT, LDQ, a.qp1, sP, LDH, nullptr, 0, wave, lane, rot);
    }
    __syncthreads();
  }
  DEC_STAMP(9);   // hs store + qpos MLP
  for (int e = tid; e < DR * D; e += NT) {
    const int r = e / D, c = e - r * D;
    sA[r * LDH + c] = sH[r * LDH + c] + sP[r * LDH + c];
    if (a.mode != 3 && r < nvalid) a.qpos_out[(row0 + r) * D + c] = sP[r * LDH + c];
  }
  __syncthreads();
  DEC_GEMM(ACT_NONE, sA, LDH, a.qk, sT, LDQ, nullptr, 0, no_next);
  if (SPLIT) __syncthreads();                                   // every wave is out of the q|k GEMM before its staging rows are re-split
  DEC_GEMM(ACT_NONE, sH, LDH, a.v, sO, LDO, nullptr, 0, no_next);
  __syncthreads();
  DEC_STAMP(10);  // qk + v
  // q rows, and K / V of these 16 rows (= key tile `tile`) in the fragment order self_attention_rows() reads
  for (int e = tid; e < DR * D; e += NT) {
    const int r = e / D, c = e - r * D;
    if (r < nvalid) a.q_out[(row0 + r) * D + c] = sT[r * LDQ + c];
  }
  if (SPLIT && a.attn_split) {
    // split form (self_attention_rows_split): K of this tile as [hi | lo][lane][8] bf16, V into this tile's half of its pair's
    // [d][hi | lo][lane][8] fragments (8-byte pieces); the last tile of an odd count also zeroes the missing partner's half
    const int tile = q0 / DR, tp = (tiles + 1) & ~1, pr = tile >> 1, eh = tile & 1;
    const bool zero_partner = (tile == tiles - 1) && (tiles & 1);
    for (int e = tid; e < a.heads * 128; e += NT) {              // K: one 16-byte piece per (head, hi/lo, lane)
      const int h = e >> 7, hl = (e >> 6) & 1, ln = e & 63;
      const int kr = ln & 15;
      sp16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float x = kr < nvalid ? sT[kr * LDQ + D + h * 32 + 8 * (ln >> 4) + j] : 0.f;
        const sp16 hi = (sp16)x;
        o[j] = hl ? (sp16)(x - (float)hi) : hi;
      }
      *(sp16x8*)((char*)a.kfrag_out + ((size_t)(b * a.heads + h) * tp + tile) * 2048 + hl * 1024 + ln * 16) = o;
    }
    for (int e = tid; e < a.heads * 256; e += NT) {              // V: one 8-byte piece per (head, d, hi/lo, lane)
      const int h = e >> 8, d = (e >> 7) & 1, hl = (e >> 6) & 1, ln = e & 63;
      sp16x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int vr = 4 * (ln >> 4) + r;
        const float x = vr < nvalid ? sO[vr * LDO + h * 32 + 16 * d + (ln & 15)] : 0.f;
        const sp16 hi = (sp16)x;
        o[r] = hl ? (sp16)(x - (float)hi) : hi;
      }
      char* dst = (char*)a.vfrag_out + ((size_t)(b * a.heads + h) * tp + 2 * pr) * 2048 + (d * 2 + hl) * 1024 + ln * 16;
      *(sp16x4*)(dst + eh * 8) = o;
      if (zero_partner) *(sp16x4*)(dst + 8) = sp16x4{(sp16)0.f, (sp16)0.f, (sp16)0.f, (sp16)0.f};
    }
  } else {
    const int tile = q0 / DR;
    const int tp = (tiles + 1) & ~1;
    for (int e = tid; e < a.heads * 512; e += NT) {
      const int h = e >> 9, rem = e & 511, c = rem >> 8, ln = (rem & 255) >> 2, u = rem & 3;
      const size_t dst = ((size_t)(b * a.heads + h) * tp + tile) * 512 + rem;
      const int kr = ln & 15;                                   // K: key = lane & 15, dim = 32h + 16c + 4 (lane>>4) + u
      a.kfrag_out[dst] = kr < nvalid ? sT[kr * LDQ + D + h * 32 + 16 * c + 4 * (ln >> 4) + u] : 0.f;
      const int vr = 4 * (ln >> 4) + u;                         // V: key = 4 (lane>>4) + u, dim = 32h + 16c + (lane & 15)
      a.vfrag_out[dst] = vr < nvalid ? sO[vr * LDO + h * 32 + 16 * c + (ln & 15)] : 0.f;
    }
  }
  DEC_STAMP(11);  // stores
}

// ---- query selection scores: LayerNorm + enc_score_head + class max in one launch (exact fp32 MFMA) ----------------------------
// One WAVE owns a 16-row tile end to end (no block-level barrier): load 16 x 256 fp32 (load i of a lane = row i, columns
// 4 lane..+3), LayerNorm in registers, rows to the wave's LDS slab as the MFMA A operand, score GEMM with the filter fragments
// streamed from L2 one 16-deep chunk ahead, + bias, max over the classes straight from the accumulators.  The chunk order is
// rotated by the tile's index inside its image (de-synchronises the waves' filter streams; a row's summation order then depends
// only on its position in the image - batch order invariance).
constexpr int SEL_WAVES = 4;
__global__ __launch_bounds__(64 * SEL_WAVES, 2) void select_score_kernel(const SelArgs a) {
  constexpr int LDH = 260;
  __shared__ __attribute__((aligned(16))) float sX[SEL_WAVES][DR * LDH];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long tile = (long long)blockIdx.x * SEL_WAVES + wave;
  const long long row0 = tile * DR;
  if (row0 >= a.rows) return;
  const int nvalid = (int)min((long long)DR, (long long)a.rows - row0);
  float* xs = sX[wave];
  {
    f32x4_ v[DR];
#pragma unroll
    for (int i = 0; i < DR; ++i) v[i] = *(const f32x4_*)(a.x + (row0 + min(i, nvalid - 1)) * a.ldx + lane * 4);
    const f32x4_ g4 = *(const f32x4_*)(a.ln.g + lane * 4), b4 = *(const f32x4_*)(a.ln.b + lane * 4);
#pragma unroll
    for (int i = 0; i < DR; ++i) {
      const float s = wave_sum64(v[i][0] + v[i][1] + v[i][2] + v[i][3]);
      const float mean = s / 256.f;
      float sq = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) sq += (v[i][j] - mean) * (v[i][j] - mean);
      sq = wave_sum64(sq);
      const float rstd = rsqrtf(sq / 256.f + 1e-5f);
      f32x4_ o4;
#pragma unroll
      for (int j = 0; j < 4; ++j) o4[j] = (v[i][j] - mean) * rstd * g4[j] + b4[j];
      *(f32x4_*)(xs + i * LDH + lane * 4) = o4;
    }
  }
  __builtin_amdgcn_wave_barrier();
  constexpr int MAXT = 5;                                         // <= 80 classes per pass; more classes: further passes
  const int ntiles = (a.C + 15) >> 4;
  const int r16 = lane & 15, q = lane >> 4;
  const float* xrow = xs + r16 * LDH + 4 * q;
  const int kc = a.score.K >> 4;                                  // 16 chunks
  const int rot = (a.rows_per_image % DR == 0) ? (int)((row0 % a.rows_per_image) / DR) % kc : 0;
  float best[4] = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
  for (int t0 = 0; t0 < ntiles; t0 += MAXT) {
    const int nt = min(MAXT, ntiles - t0);
    f32x4_ acc[MAXT];
#pragma unroll
    for (int t = 0; t < MAXT; ++t) acc[t] = f32x4_{0.f, 0.f, 0.f, 0.f};
    // filter fragments two chunks ahead (a chunk's 20 MFMAs are ~0.3 us, an L2 round trip under load 0.5-1 us)
    f32x4_ cur[MAXT], n1[MAXT], n2[MAXT];
    auto wrap = [&](int v) { return v >= kc ? v - kc : v; };
    auto ldw = [&](f32x4_ (&dst)[MAXT], int chunk) {
#pragma unroll
      for (int t = 0; t < MAXT; ++t) dst[t] = *(const f32x4_*)(a.score.w + ((size_t)(t0 + min(t, nt - 1)) * kc + chunk) * 256 + lane * 4);
    };
    int c = rot;
    ldw(cur, c);
    ldw(n1, wrap(c + 1));
    for (int it = 0; it < kc; ++it) {
      const int k0 = c << 4;
      ldw(n2, wrap(c + 2));
      c = wrap(c + 1);
      __builtin_amdgcn_sched_barrier(0);
      const f32x4_ x4 = *(const f32x4_*)(xrow + k0);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int t = 0; t < MAXT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(x4[u], cur[t][u], acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < MAXT; ++t) { cur[t] = n1[t]; n1[t] = n2[t]; }
    }
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
      const int col = (t0 + t) * 16 + r16;
      if (t < nt && col < a.C) {
        const float bv = a.score.b[col];
#pragma unroll
        for (int r = 0; r < 4; ++r) best[r] = fmaxf(best[r], acc[t][r] + bv);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float v = row16_max(best[r]);                            // the 16 lanes of this q group = the 16 columns of a tile
    if (r16 == 0 && q * 4 + r < nvalid) a.mx[row0 + q * 4 + r] = v;
  }
}

void launch_select_score(const SelArgs& a, hipStream_t s) {
  RTD_CHECK(a.score.K == 256 && a.score.N == a.C && a.rows_per_image > 0, 1, "select_score: 256-wide rows");
  const long long tiles = (a.rows + DR - 1) / DR;
  rtd_launch(select_score_kernel, dim3((unsigned)((tiles + SEL_WAVES - 1) / SEL_WAVES)), dim3(64 * SEL_WAVES), 0, s, a);
  HIP_CHECK(hipGetLastError());
}

__global__ void k_gather_ln(const float* __restrict__ x, long long ldx, int rows_per_image, const int32_t* __restrict__ idx, int total, int Q,
                            DecLN ln, float* __restrict__ dst, long long ldd) {
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);          // one wave per selected row
  const int lane = threadIdx.x & 63;
  if (t >= total) return;
  const int b = t / Q;
  int r = idx[t];
  r = min(max(r, 0), rows_per_image - 1);
  const f32x4_ v = *(const f32x4_*)(x + ((long long)b * rows_per_image + r) * ldx + lane * 4);
  const float s = wave_sum64(v[0] + v[1] + v[2] + v[3]);
  const float mean = s / 256.f;
  float sq = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) sq += (v[j] - mean) * (v[j] - mean);
  sq = wave_sum64(sq);
  const float rstd = rsqrtf(sq / 256.f + 1e-5f);
  f32x4_ o4;
#pragma unroll
  for (int j = 0; j < 4; ++j) o4[j] = (v[j] - mean) * rstd * ln.g[lane * 4 + j] + ln.b[lane * 4 + j];
  *(f32x4_*)(dst + (long long)t * ldd + lane * 4) = o4;
}
void launch_gather_ln(const float* x, int64_t ldx, int rows_per_image, const int32_t* idx, int B, int Q, const DecLN& ln, float* dst, int64_t ldd,
                      hipStream_t s) {
  const int total = B * Q;
  rtd_launch(k_gather_ln, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, s, x, (long long)ldx, rows_per_image, idx, total, Q, ln, dst, (long long)ldd);
  HIP_CHECK(hipGetLastError());
}

void launch_dec_layer(const DecArgs& a, hipStream_t s) {
  RTD_CHECK(a.D == 256 && a.D / a.heads == 32 && a.ffn <= 1024 && a.C <= 512, 1, "fused decoder: d_model 256, head dim 32, ffn <= 1024");
  RTD_CHECK(a.n_levels == 3 && a.n_points == 4 && a.heads == NW, 1, "fused decoder: 3 levels x 4 points, one head per wave");
  const int tiles = (a.Q + DR - 1) / DR;
  if (a.split == 2) rtd_launch(dec_layer_kernel<2>, dim3(a.B * tiles), dim3(NT), 0, s, a);
  else if (a.split) rtd_launch(dec_layer_kernel<1>, dim3(a.B * tiles), dim3(NT), 0, s, a);
  else rtd_launch(dec_layer_kernel<0>, dim3(a.B * tiles), dim3(NT), 0, s, a);
  HIP_CHECK(hipGetLastError());
}

}  // namespace rtd
