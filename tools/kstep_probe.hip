// kstep_probe.hip - where do the cycles of a pair-kernel K-step go?  One block of 8 waves per CU as in conv_igemm_wsq_kernel (4 MFMA waves,
// 4 waves that only join the barrier); the MFMA waves run MT x 4 accumulator tiles x 3 products per step in several variants:
//   bit 0: fragment reads from LDS (2 MT + 8 ds_read_b128 per step, conflict-free rows) instead of operands held in registers
//   bit 1: one s_barrier per step (all 8 waves)
//   bit 2: product-major MFMA order (hh for the four channel tiles, then hl, then lh) instead of chain-major (hh, hl, lh per tile)
//   bit 3: the second wave of every SIMD issues LDS-DMA like a loader (11 x 1 KiB per step from a small L2-resident buffer)
// Prints shader clocks per K-step (s_memtime, median wave) against MT x 12 x 16.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/_bin/kstep_probe tools/kstep_probe.hip && tools/_bin/kstep_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <type_traits>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MT, int V>
__global__ __launch_bounds__(512, 2) void k_step(const unsigned* __restrict__ src, unsigned src_bytes, int steps, long long* __restrict__ out, float* __restrict__ sink) {
  constexpr bool LDS = V & 1, BAR = V & 2, PMAJ = V & 4, DMA = V & 8;
  constexpr int STAGE = (MT * 2 * 16 + 128) * 128;
  __shared__ __attribute__((aligned(16))) char smem[3 * STAGE + 256];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const bool loader = wv >= 4;
  const int w4 = wv & 3, wm = w4 & 1, wn = w4 >> 1;
  for (int i = tid; i < 3 * STAGE / 4; i += 512) ((unsigned*)smem)[i] = src[i & 4095];
  __syncthreads();
  f4 acc[4][MT];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < MT; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
  const int r16 = lane & 15, c4 = lane >> 4, sw = (r16 >> 1) & 7;
  const int foh = r16 * 128 + ((c4 ^ sw) << 4), fol = r16 * 128 + (((c4 + 4) ^ sw) << 4);
  long long c0 = 0, c1 = 0;
  if (loader) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, src_bytes, 0x00020000);
    for (int ks = 0; ks < steps; ++ks) {
      if (DMA) {
        asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
      }
      if (BAR) __builtin_amdgcn_s_barrier();
      if (DMA) {
        char* sa = smem + (ks % 3) * STAGE + w4 * 1024;
#pragma unroll
        for (int i = 0; i < 11; ++i)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(sa + i * 4096), 16, (unsigned)(((ks * 11 + i) * 4096 + w4 * 1024 + lane * 16) & (src_bytes - 1)), 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    h8 wh[4], wl[4], xh[2], xl[2];
    for (int i = 0; i < 4; ++i) { wh[i] = *(const h8*)(smem + (MT * 32 + wn * 64) * 128 + i * 2048 + foh); wl[i] = *(const h8*)(smem + (MT * 32 + wn * 64) * 128 + i * 2048 + fol); }
    xh[0] = xh[1] = *(const h8*)(smem + foh); xl[0] = xl[1] = *(const h8*)(smem + fol);
    c0 = __builtin_amdgcn_s_memtime();
    for (int ks = 0; ks < steps; ++ks) {
      if (BAR) __builtin_amdgcn_s_barrier();
      const char* sa = smem + (ks % 3) * STAGE + wm * MT * 16 * 128;
      const char* sb = smem + (ks % 3) * STAGE + (MT * 32 + wn * 64) * 128;
      if (LDS) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { wh[i] = *(const h8*)(sb + i * 2048 + foh); wl[i] = *(const h8*)(sb + i * 2048 + fol); }
        xh[0] = *(const h8*)(sa + foh); xl[0] = *(const h8*)(sa + fol);
      }
#pragma unroll
      for (int j = 0; j < MT; ++j) {
        if (LDS && j + 1 < MT) { xh[(j + 1) & 1] = *(const h8*)(sa + (j + 1) * 2048 + foh); xl[(j + 1) & 1] = *(const h8*)(sa + (j + 1) * 2048 + fol); }
        __builtin_amdgcn_sched_barrier(0);
        if (PMAJ) {
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[i], xh[j & 1], acc[i][j], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[i], xl[j & 1], acc[i][j], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[i], xh[j & 1], acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[i], xh[j & 1], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[i], xl[j & 1], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[i], xh[j & 1], acc[i][j], 0, 0, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_s_waitcnt(0);
    c1 = __builtin_amdgcn_s_memtime();
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < MT; ++j) s += acc[i][j][0] + acc[i][j][2];
  if (s == 123.456f) sink[0] = s;
  if (lane == 0 && !loader) out[blockIdx.x * 4 + wv] = c1 - c0;
}

// The candidate schedule: the LAST tile's 12 MFMAs are deferred past the next barrier, where they are interleaved one to one with the new
// step's first ten fragment reads (filter fragments double-buffered by step parity); the two reads of pixel tile j + 1 sit after the
// first and the fifth MFMA of tile j: never more than one ds_read_b128 per MFMA gap.
template <int MT>
__global__ __launch_bounds__(512, 2) void k_step_il(const unsigned* __restrict__ src, unsigned src_bytes, int steps, long long* __restrict__ out, float* __restrict__ sink) {
  constexpr int STAGE = (MT * 2 * 16 + 128) * 128;
  __shared__ __attribute__((aligned(16))) char smem[3 * STAGE + 256];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const bool loader = wv >= 4;
  const int w4 = wv & 3, wm = w4 & 1, wn = w4 >> 1;
  for (int i = tid; i < 3 * STAGE / 4; i += 512) ((unsigned*)smem)[i] = src[i & 4095];
  __syncthreads();
  f4 acc[4][MT];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < MT; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
  const int r16 = lane & 15, c4 = lane >> 4, sw = (r16 >> 1) & 7;
  const int foh = r16 * 128 + ((c4 ^ sw) << 4), fol = r16 * 128 + (((c4 + 4) ^ sw) << 4);
  long long c0 = 0, c1 = 0;
  if (loader) {
    for (int ks = 0; ks < steps; ++ks) __builtin_amdgcn_s_barrier();
  } else {
    h8 wh[2][4], wl[2][4], xh[2], xl[2], ph, pl;
    for (int i = 0; i < 4; ++i) { wh[1][i] = *(const h8*)(smem + (MT * 32 + wn * 64) * 128 + i * 2048 + foh); wl[1][i] = *(const h8*)(smem + (MT * 32 + wn * 64) * 128 + i * 2048 + fol); }
    ph = *(const h8*)(smem + foh); pl = *(const h8*)(smem + fol);
    c0 = __builtin_amdgcn_s_memtime();
#define SB() __builtin_amdgcn_sched_barrier(0)
#define MF(A, B, C) C = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, C, 0, 0, 0)
    auto step = [&](auto parc, const int ks) __attribute__((always_inline)) {
      constexpr int P = decltype(parc)::value;
      __builtin_amdgcn_s_barrier();
      const char* sa = smem + (ks % 3) * STAGE + wm * MT * 16 * 128;
      const char* sb = smem + (ks % 3) * STAGE + (MT * 32 + wn * 64) * 128;
      // deferred tile of the previous step (registers only), one new read per gap
      wh[P][0] = *(const h8*)(sb + foh); xh[0] = *(const h8*)(sa + foh); SB();
      MF(wh[P ^ 1][0], ph, acc[0][MT - 1]); wh[P][1] = *(const h8*)(sb + 2048 + foh); SB();
      MF(wh[P ^ 1][1], ph, acc[1][MT - 1]); wh[P][2] = *(const h8*)(sb + 4096 + foh); SB();
      MF(wh[P ^ 1][2], ph, acc[2][MT - 1]); wh[P][3] = *(const h8*)(sb + 6144 + foh); SB();
      MF(wh[P ^ 1][3], ph, acc[3][MT - 1]); xl[0] = *(const h8*)(sa + fol); SB();
      MF(wh[P ^ 1][0], pl, acc[0][MT - 1]); wl[P][0] = *(const h8*)(sb + fol); SB();
      MF(wh[P ^ 1][1], pl, acc[1][MT - 1]); wl[P][1] = *(const h8*)(sb + 2048 + fol); SB();
      MF(wh[P ^ 1][2], pl, acc[2][MT - 1]); wl[P][2] = *(const h8*)(sb + 4096 + fol); SB();
      MF(wh[P ^ 1][3], pl, acc[3][MT - 1]); wl[P][3] = *(const h8*)(sb + 6144 + fol); SB();
      MF(wl[P ^ 1][0], ph, acc[0][MT - 1]); MF(wl[P ^ 1][1], ph, acc[1][MT - 1]); MF(wl[P ^ 1][2], ph, acc[2][MT - 1]); MF(wl[P ^ 1][3], ph, acc[3][MT - 1]); SB();
#pragma unroll
      for (int j = 0; j < MT - 1; ++j) {
        MF(wh[P][0], xh[j & 1], acc[0][j]); xh[(j + 1) & 1] = *(const h8*)(sa + (j + 1) * 2048 + foh); SB();
        MF(wh[P][1], xh[j & 1], acc[1][j]); MF(wh[P][2], xh[j & 1], acc[2][j]); MF(wh[P][3], xh[j & 1], acc[3][j]); SB();
        MF(wh[P][0], xl[j & 1], acc[0][j]); xl[(j + 1) & 1] = *(const h8*)(sa + (j + 1) * 2048 + fol); SB();
        MF(wh[P][1], xl[j & 1], acc[1][j]); MF(wh[P][2], xl[j & 1], acc[2][j]); MF(wh[P][3], xl[j & 1], acc[3][j]); SB();
        MF(wl[P][0], xh[j & 1], acc[0][j]); MF(wl[P][1], xh[j & 1], acc[1][j]); MF(wl[P][2], xh[j & 1], acc[2][j]); MF(wl[P][3], xh[j & 1], acc[3][j]); SB();
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      ph = xh[(MT - 1) & 1]; pl = xl[(MT - 1) & 1];
    };
    for (int ks = 0; ks + 1 < steps; ks += 2) { step(std::integral_constant<int, 0>{}, ks); step(std::integral_constant<int, 1>{}, ks + 1); }
#undef SB
#undef MF
    __builtin_amdgcn_s_waitcnt(0);
    c1 = __builtin_amdgcn_s_memtime();
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < MT; ++j) s += acc[i][j][0] + acc[i][j][2];
  if (s == 123.456f) sink[0] = s;
  if (lane == 0 && !loader) out[blockIdx.x * 4 + wv] = c1 - c0;
}

template <int MT>
static void run_il(const unsigned* src, unsigned src_bytes, long long* out, float* sink, int cus) {
  const int steps = 2000;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((k_step_il<MT>), dim3(cus), dim3(512), 0, 0, src, src_bytes, steps, out, sink);
    CHECK(hipDeviceSynchronize());
  }
  std::vector<long long> h((size_t)cus * 4);
  CHECK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
  std::sort(h.begin(), h.end());
  printf("MT %d  %-58s %7.0f clocks per K-step (MFMA %d)\n", MT, "LDS reads + barrier, deferred last tile, one read per gap", (double)h[h.size() / 2] / steps, MT * 12 * 16);
}

template <int MT, int V>
static void run(const unsigned* src, unsigned src_bytes, long long* out, float* sink, int cus, const char* what) {
  const int steps = 2000;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((k_step<MT, V>), dim3(cus), dim3(512), 0, 0, src, src_bytes, steps, out, sink);
    CHECK(hipDeviceSynchronize());
  }
  std::vector<long long> h((size_t)cus * 4);
  CHECK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
  std::sort(h.begin(), h.end());
  printf("MT %d  %-58s %7.0f clocks per K-step (MFMA %d)\n", MT, what, (double)h[h.size() / 2] / steps, MT * 12 * 16);
}

int main() {
  hipDeviceProp_t p;
  CHECK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  const unsigned src_bytes = 1u << 20;
  unsigned* src; long long* out; float* sink;
  CHECK(hipMalloc(&src, src_bytes)); CHECK(hipMalloc(&out, (size_t)cus * 4 * 8)); CHECK(hipMalloc(&sink, 16));
  std::vector<unsigned> hs(src_bytes / 4);
  for (size_t i = 0; i < hs.size(); ++i) {
    unsigned h = (unsigned)i * 2654435761u + 12345u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    unsigned lo = (h & 0x8000u) | ((10u + ((h >> 20) & 7u)) << 10) | (h & 0x3ffu);
    unsigned h2 = h * 2654435761u; h2 ^= h2 >> 16;
    unsigned hi = (h2 & 0x8000u) | ((10u + ((h2 >> 20) & 7u)) << 10) | (h2 & 0x3ffu);
    hs[i] = lo | (hi << 16);
  }
  CHECK(hipMemcpy(src, hs.data(), src_bytes, hipMemcpyHostToDevice));
  run<7, 0>(src, src_bytes, out, sink, cus, "registers only, chain-major");
  run<7, 4>(src, src_bytes, out, sink, cus, "registers only, product-major");
  run<7, 1>(src, src_bytes, out, sink, cus, "+ LDS fragment reads, chain-major");
  run<7, 5>(src, src_bytes, out, sink, cus, "+ LDS fragment reads, product-major");
  run<7, 3>(src, src_bytes, out, sink, cus, "+ LDS reads + barrier, chain-major");
  run<7, 7>(src, src_bytes, out, sink, cus, "+ LDS reads + barrier, product-major");
  run<7, 11>(src, src_bytes, out, sink, cus, "+ LDS reads + barrier + LDS-DMA waves, chain-major");
  run<7, 15>(src, src_bytes, out, sink, cus, "+ LDS reads + barrier + LDS-DMA waves, product-major");
  run_il<7>(src, src_bytes, out, sink, cus);
  run_il<4>(src, src_bytes, out, sink, cus);
  run_il<8>(src, src_bytes, out, sink, cus);
  run<4, 3>(src, src_bytes, out, sink, cus, "+ LDS reads + barrier, chain-major");
  run<4, 7>(src, src_bytes, out, sink, cus, "+ LDS reads + barrier, product-major");
  return 0;
}
