// mfma_rate_probe.hip - what the matrix pipes of THIS chip sustain: v_mfma_f32_16x16x32_f16 back to back on every CU, operands in
// registers (no LDS, no memory), zero vs random operand bits, 1 / 2 waves per SIMD.  Prints shader clocks per MFMA (s_memtime), the
// core clock the kernel ran at (s_memtime ticks per s_memrealtime tick, 100 MHz) and the chip-wide rate.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/mfma_rate_probe tools/mfma_rate_probe.hip && tools/_bin/mfma_rate_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// NACC independent accumulators per wave, `iters` rounds of NACC MFMAs
template <int NACC>
__global__ __launch_bounds__(512) void k_mfma(const unsigned* __restrict__ seed, int iters, long long* __restrict__ out, float* __restrict__ sink) {
  const int tid = threadIdx.x, lane = tid & 63;
  h8 a[4], b[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 8; ++j) {
      unsigned h = seed[(lane * 8 + j + i * 512) & 4095];
      unsigned short ua = (unsigned short)h, ub = (unsigned short)(h >> 16);
      a[i][j] = __builtin_bit_cast(_Float16, ua);
      b[i][j] = __builtin_bit_cast(_Float16, ub);
    }
  f4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  const long long c0 = __builtin_amdgcn_s_memtime();
  const long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
  }
  __builtin_amdgcn_s_waitcnt(0);
  const long long c1 = __builtin_amdgcn_s_memtime();
  const long long r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
  if (s == 123.456f) sink[0] = s;
  if (lane == 0) {
    const int w = blockIdx.x * (blockDim.x >> 6) + (tid >> 6);
    out[2 * w] = c1 - c0;
    out[2 * w + 1] = r1 - r0;
  }
}

int main() {
  hipDeviceProp_t p;
  CHECK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  printf("device %s, %d CUs, clockRate %d kHz\n", p.name, cus, p.clockRate);
  unsigned* seed; long long* out; float* sink;
  CHECK(hipMalloc(&seed, 4096 * 4)); CHECK(hipMalloc(&out, (size_t)cus * 8 * 2 * 8)); CHECK(hipMalloc(&sink, 16));
  std::vector<unsigned> hs(4096);
  const int iters = 40000;
  for (int rnd = 0; rnd < 2; ++rnd) {
    for (int i = 0; i < 4096; ++i) {
      unsigned h = (unsigned)i * 2654435761u + 12345u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
      // two finite fp16 values: exponents 2^-5 .. 2^2, random mantissa and sign
      unsigned lo = (h & 0x8000u) | ((10u + ((h >> 20) & 7u)) << 10) | (h & 0x3ffu);
      unsigned h2 = h * 2654435761u; h2 ^= h2 >> 16;
      unsigned hi = (h2 & 0x8000u) | ((10u + ((h2 >> 20) & 7u)) << 10) | (h2 & 0x3ffu);
      hs[i] = rnd ? (lo | (hi << 16)) : 0u;
    }
    CHECK(hipMemcpy(seed, hs.data(), 4096 * 4, hipMemcpyHostToDevice));
    for (int waves : {4, 8}) {
      const int threads = waves * 64;
      std::vector<long long> ho((size_t)cus * waves * 2);
      hipEvent_t e0, e1;
      CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
      for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_mfma<16>, dim3(cus), dim3(threads), 0, 0, seed, iters, out, sink);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
      }
      float ms = 0.f;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      CHECK(hipMemcpy(ho.data(), out, ho.size() * 8, hipMemcpyDeviceToHost));
      double cyc = 0, rt = 0;
      for (int w = 0; w < cus * waves; ++w) { cyc += ho[2 * w]; rt += ho[2 * w + 1]; }
      cyc /= cus * waves; rt /= cus * waves;
      const double mfmas_per_wave = (double)iters * 16;
      const double ghz = cyc / (rt * 10.0);                      // 100 MHz real-time ticks -> ns
      const double flops = mfmas_per_wave * cus * waves * 16.0 * 16.0 * 32.0 * 2.0;
      printf("%-7s operands, %d waves/CU: %.1f shader clocks per MFMA per wave (per SIMD: %.1f), core clock %.3f GHz, kernel %.3f ms, %.0f TFLOP/s\n",
             rnd ? "random" : "zero", waves, cyc / mfmas_per_wave, cyc / mfmas_per_wave / (waves / 4.0), ghz, ms, flops / (ms * 1e-3) / 1e12);
      if (waves == 4) {
        // dependent chains: NACC accumulators in rotation (1 = every MFMA waits for the previous one's result)
        auto chain = [&](auto kern, int nacc) {
          hipLaunchKernelGGL(kern, dim3(cus), dim3(threads), 0, 0, seed, iters, out, sink);
          CHECK(hipDeviceSynchronize());
          std::vector<long long> h2((size_t)cus * waves * 2);
          CHECK(hipMemcpy(h2.data(), out, h2.size() * 8, hipMemcpyDeviceToHost));
          double c = 0;
          for (int w = 0; w < cus * waves; ++w) c += h2[2 * w];
          printf("  %2d accumulator(s) in rotation: %.1f shader clocks per MFMA\n", nacc, c / (cus * waves) / ((double)iters * nacc));
        };
        chain(k_mfma<1>, 1); chain(k_mfma<2>, 2); chain(k_mfma<3>, 3); chain(k_mfma<4>, 4); chain(k_mfma<8>, 8);
      }
    }
  }
  return 0;
}
