// Probe: how fast can ONE CU take bytes from L2 / HBM into LDS (or registers), as a function of the issue form, the number
// of issuing waves, the bytes kept in flight and how many CUs do it at once?  The conv kernels' K-step time tracks the
// bytes staged per CU; this pins the ceiling that figure should be compared with.
// build: hipcc --offload-arch=gfx950 -O3 tools/ingest_probe.hip -o gpurun_out/ingest_probe && ./gpurun_out/ingest_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct P {
  const char* src;
  unsigned src_bytes;      // < 4 GiB (buffer descriptor)
  unsigned region;         // bytes each block cycles through (power of two)
  unsigned block_stride;   // byte distance between the regions of consecutive blocks (0 = all share one region)
  int iters;               // batches per loader wave
  int nload;               // loader waves (the first nload waves of waves 4..7 then 0..3)
  int consumers;           // 0 none, 1 = waves 0-3 loop ds_read_b128, 2 = ds_read_b128 + MFMA
  float* sink;
};

// MODE 0: buffer_load_dwordx4 ... lds (DEPTH x 1 KiB per wave per batch, vmcnt(0) between batches)
// MODE 1: global_load_dwordx4 to registers, xor-reduced (no LDS)
// MODE 2: global_load_dwordx4 to registers + ds_write_b128
// MODE 3: like 0 but rolling: keeps DEPTH-1 DMAs in flight (counted vmcnt), what a pipelined GEMM loader does
template <int MODE, int DEPTH>
__global__ __launch_bounds__(512, 2) void k(const P p) {
  __shared__ __attribute__((aligned(16))) char smem[8 * 8 * 1024 + 65536];   // 8 waves x 8 KiB slots + a consumer tile; 128 KiB -> 1 block/CU
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lw = (wv + 4) & 7;                       // loader index: waves 4..7 are loaders 0..3, waves 0..3 loaders 4..7
  const unsigned base = (unsigned)((unsigned long long)blockIdx.x * p.block_stride);
  if (lw < p.nload) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.src, 0, p.src_bytes, 0x00020000);
    char* my = smem + wv * 8192;
    unsigned off = (unsigned)(lw * 1024 + lane * 16);
    const unsigned step = (unsigned)(p.nload * 1024);
    const unsigned mask = p.region - 1;
    f32x4 accv = {0.f, 0.f, 0.f, 0.f};
    if (MODE == 0) {
      for (int it = 0; it < p.iters; ++it) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(my + (d & 7) * 1024), 16, base + (off & mask), 0, 0, 0);
          off += step;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    } else if (MODE == 3) {
      for (int it = 0; it < p.iters; ++it) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(my + (d & 7) * 1024), 16, base + (off & mask), 0, 0, 0);
          off += step;
          if (DEPTH >= 16) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
          else if (DEPTH >= 8) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      const char* g = p.src;
      for (int it = 0; it < p.iters; ++it) {
        f32x4 v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
          v[d] = *(const f32x4*)(g + base + (off & mask));
          off += step;
        }
        if (MODE == 1) {
#pragma unroll
          for (int d = 0; d < DEPTH; ++d) accv += v[d];
        } else {
#pragma unroll
          for (int d = 0; d < DEPTH; ++d) *(f32x4*)(my + (d & 7) * 1024 + lane * 16) = v[d];
        }
      }
    }
    if (MODE == 1 && accv[0] + accv[1] + accv[2] + accv[3] == 12345.678f) p.sink[tid] = accv[0];
  } else if (p.consumers && wv < 4) {
    // consumer waves: per "K-step" 16 ds_read_b128 (+16 MFMA), like the conv kernel's MFMA role
    const char* t = smem + 65536 + (lane & 31) * 128 + ((lane >> 5) << 4);
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int steps = p.iters * DEPTH / 8;           // roughly one step per 32 KiB the loaders move (4 loaders x 8 KiB)
    for (int s = 0; s < steps; ++s) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        bf16x8 a0 = *(const bf16x8*)(t + kk * 32), a1 = *(const bf16x8*)(t + 4096 + kk * 32);
        bf16x8 b0 = *(const bf16x8*)(t + 8192 + kk * 32), b1 = *(const bf16x8*)(t + 12288 + kk * 32);
        if (p.consumers == 2) {
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[1], 0, 0, 0);
          acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[2], 0, 0, 0);
          acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[3], 0, 0, 0);
        } else {
          acc[0][0] += (float)a0[0] + (float)a1[1] + (float)b0[2] + (float)b1[3];
        }
      }
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) sum += acc[i][0] + acc[i][7];
    if (sum == 12345.678f) p.sink[tid] = sum;
  }
}

template <int MODE, int DEPTH>
static float run(const P& p, int nblk) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<MODE, DEPTH>), dim3(nblk), dim3(512), 0, 0, p);      // warm (fills L2 where the region fits)
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<MODE, DEPTH>), dim3(nblk), dim3(512), 0, 0, p);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms;
}

template <int MODE, int DEPTH>
static void report(const char* what, P p, int nblk, long long bytes_per_wave) {
  p.iters = (int)(bytes_per_wave / (DEPTH * 1024));
  const float ms = run<MODE, DEPTH>(p, nblk);
  const double bytes = (double)p.iters * DEPTH * 1024.0 * p.nload * nblk;
  printf("%-34s mode=%d depth=%2d loaders=%d cons=%d blocks=%3d  %8.1f us  %7.1f GB/s/CU  %6.2f TB/s\n", what, MODE, DEPTH, p.nload, p.consumers, nblk,
         ms * 1e3, bytes / (ms * 1e-3) / nblk / 1e9, bytes / (ms * 1e-3) / 1e12);
  fflush(stdout);
}

int main() {
  const size_t big = 3ull << 30;           // 3 GiB source
  char* src; float* sink;
  CHECK(hipMalloc(&src, big)); CHECK(hipMalloc(&sink, 4096));
  CHECK(hipMemset(src, 1, big));
  P p{};
  p.src = src; p.src_bytes = (unsigned)big; p.sink = sink;
  const long long W = 4ll << 20;           // bytes per loader wave per launch

  struct Pat { const char* name; unsigned region, stride; int nblk; } pats[] = {
      {"private 64K (L2 hit)", 64u << 10, 64u << 10, 256},
      {"private 64K (L2 hit)", 64u << 10, 64u << 10, 64},
      {"shared 512K (L2 hit, weights-like)", 512u << 10, 0, 256},
      // L2 miss, Infinity-Cache (MALL, 256 MB) resident: per-XCD footprint >= 8 MB (2x its L2), total <= 128 MB
      {"private 512K (MALL hit)", 512u << 10, 512u << 10, 256},
      {"private 1M (MALL hit)", 1u << 20, 1u << 20, 128},
      {"private 1M (MALL hit)", 1u << 20, 1u << 20, 64},
      {"private 4M (MALL hit)", 4u << 20, 4u << 20, 16},
      {"shared 16M (MALL hit, all blocks same)", 16u << 20, 0, 256},
      {"shared 16M (MALL hit, all blocks same)", 16u << 20, 0, 64},
      // beyond the MALL: HBM
      {"private 8M (HBM stream)", 8u << 20, 8u << 20, 256},
      {"private 8M (HBM stream)", 8u << 20, 8u << 20, 128},
      {"private 8M (HBM stream)", 8u << 20, 8u << 20, 64},
      {"private 32M (HBM stream)", 32u << 20, 32u << 20, 16},
      {"private 32M (HBM stream)", 32u << 20, 32u << 20, 8},
  };
  for (auto& pt : pats) {
    p.region = pt.region; p.block_stride = pt.stride;
    const int nblk = pt.nblk;
    printf("---- %s, %d blocks\n", pt.name, nblk);
    for (int nl : {1, 2, 4, 8}) {
      p.nload = nl; p.consumers = 0;
      report<0, 8>(pt.name, p, nblk, W);
      if (nl == 4) {
        report<0, 2>(pt.name, p, nblk, W);
        report<0, 4>(pt.name, p, nblk, W);
        report<0, 16>(pt.name, p, nblk, W);
        report<0, 32>(pt.name, p, nblk, W);
        report<3, 16>(pt.name, p, nblk, W);
        report<1, 8>(pt.name, p, nblk, W);
        p.consumers = 2; report<3, 16>(pt.name, p, nblk, W);
        p.consumers = 0;
      }
      if (nl == 8) { report<1, 8>(pt.name, p, nblk, W); report<0, 16>(pt.name, p, nblk, W); }
    }
  }
  // consumer-only reference: the time of the ds_read + MFMA loop with no loader
  p.region = 64u << 10; p.block_stride = 64u << 10; p.nload = 0; p.consumers = 2;
  { P q = p; q.iters = (int)(W / 8192); float ms = run<0, 8>(q, 256); printf("consumers only (ds_read+MFMA), %d steps: %.1f us -> %.0f ns/step\n", q.iters, ms * 1e3, ms * 1e6 / q.iters); }
  p.consumers = 1;
  { P q = p; q.iters = (int)(W / 8192); float ms = run<0, 8>(q, 256); printf("consumers only (ds_read), %d steps: %.1f us -> %.0f ns/step\n", q.iters, ms * 1e3, ms * 1e6 / q.iters); }
  return 0;
}
