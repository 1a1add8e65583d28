// Probe: what mixed read+write HBM rate can a plain streaming kernel reach on this chip?  The 1x1 convs of the first two
// backbone stages (K = 64..128) move 120-240 MB per launch with almost no arithmetic and run at about 3 TB/s; this pins
// the ceiling such a layer should be compared with, for the read:write mixes those layers have.
// build: hipcc --offload-arch=gfx950 -O3 tools/stream_probe.hip -o tools/_bin/stream_probe && ./tools/_bin/stream_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// Every thread handles 16-byte units; unit u of the output reads unit u of `r` (when RES) and unit u / ratio of `a`.
// UNROLL units are in flight per thread before the first use.  NT = nontemporal stores.
template <int UNROLL, bool RES, bool NT>
__global__ __launch_bounds__(256) void k_stream(const f32x4* __restrict__ a, const f32x4* __restrict__ r, f32x4* __restrict__ y,
                                                long long units, int a_shift) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  long long u = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; u + (UNROLL - 1) * stride < units; u += UNROLL * stride) {
    f32x4 va[UNROLL], vr[UNROLL];
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) {
      const long long v = u + i * stride;
      va[i] = a[v >> a_shift];
      if (RES) vr[i] = r[v];
    }
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) {
      f32x4 o = va[i];
      if (RES) o += vr[i];
      if (NT) __builtin_nontemporal_store(o, &y[u + i * stride]);
      else y[u + i * stride] = o;
    }
  }
  for (; u < units; u += stride) {
    f32x4 o = a[u >> a_shift];
    if (RES) o += r[u];
    y[u] = o;
  }
}

// read-only: xor-reduce
template <int UNROLL>
__global__ __launch_bounds__(256) void k_read(const f32x4* __restrict__ a, float* sink, long long units) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  f32x4 acc = {0, 0, 0, 0};
  for (long long u = (long long)blockIdx.x * blockDim.x + threadIdx.x; u + (UNROLL - 1) * stride < units; u += UNROLL * stride) {
    f32x4 v[UNROLL];
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) v[i] = a[u + i * stride];
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) acc += v[i];
  }
  if (acc[0] == 1.2345f) *sink = acc[1] + acc[2] + acc[3];
}

// 4 units in, 1 out (a 256 -> 64 channel layer's byte mix)
__global__ __launch_bounds__(256) void k_reduce4(const f32x4* __restrict__ a, f32x4* __restrict__ y, long long units_out) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long u = (long long)blockIdx.x * blockDim.x + threadIdx.x; u < units_out; u += stride) {
    const long long g = (u >> 6) * 256 + (u & 63);       // a wave reads four consecutive 1 KiB runs
    y[u] = a[g] + a[g + 64] + a[g + 128] + a[g + 192];
  }
}

// The access shape of an MFMA accumulator stored WITHOUT an LDS transpose: a wave owns 32 pixels x 64 channels of a [pixels][256 ch] bf16
// tensor; lane (p = lane & 31, h = lane >> 5) holds 32 consecutive channels = 64 bytes of pixel p and moves them as 4 x 16 bytes, so
// one instruction touches 32 different 128-byte lines (two 16-byte pieces each).  r is read and y written in that shape; `a` (the
// K = 64 input, a quarter of the bytes) is read coalesced.  8 waves = the 4 channel groups x 2 pixel halves of a 64-pixel tile.
template <int PT, bool AFRAG = false>   // pixel sub-tiles in flight per wave; AFRAG: `a` is read as MFMA B fragments by every wave (4x redundant, 32 lines per instruction)
__global__ __launch_bounds__(512) void k_scatter(const f32x4* __restrict__ a, const f32x4* __restrict__ r, f32x4* __restrict__ y, long long pixels) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int cg = wv & 3, ph = wv >> 2;
  const long long tiles = pixels / (64 * PT);
  for (long long t = blockIdx.x; t < tiles; t += gridDim.x) {
    f32x4 vr[PT][4], va[PT];
#pragma unroll
    for (int j = 0; j < PT; ++j) {
      const long long px = (t * PT + j) * 64 + ph * 32 + (lane & 31);
      const long long u = px * 32 + cg * 8 + (lane >> 5) * 4;           // 16-byte units; 512 bytes per pixel
      if (AFRAG) {
        va[j] = a[px * 8 + (lane >> 5)];
#pragma unroll
        for (int kk = 1; kk < 4; ++kk) va[j] += a[px * 8 + kk * 2 + (lane >> 5)];
      } else {
        va[j] = a[(t * PT + j) * 64 * 8 + threadIdx.x];                 // 64 px x 128 B = 8 KiB per tile: one unit per thread
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) vr[j][s] = r[u + s];
    }
#pragma unroll
    for (int j = 0; j < PT; ++j) {
      const long long px = (t * PT + j) * 64 + ph * 32 + (lane & 31);
      const long long u = px * 32 + cg * 8 + (lane >> 5) * 4;
#pragma unroll
      for (int s = 0; s < 4; ++s) y[u + s] = vr[j][s] + va[j];
    }
  }
}

template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void k_write(f32x4* __restrict__ y, long long units) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  const f32x4 o = {1.f, 2.f, 3.f, (float)threadIdx.x};
  for (long long u = (long long)blockIdx.x * blockDim.x + threadIdx.x; u < units; u += stride) {
    if (NT) __builtin_nontemporal_store(o, &y[u]);
    else y[u] = o;
  }
}

int main(int argc, char** argv) {
  const long long MB = 1 << 20;
  const long long out_bytes = 100 * MB;                  // s0.c3: 8 x 160 x 160 x 256 bf16 = 105 MB
  // ROT buffer sets used round-robin: with ROT = 1 the 225-300 MB footprint of a launch mostly lives in the 256 MB Infinity
  // Cache from one launch to the next; ROT = 8 (2.4 GB) makes every launch go to HBM, which is what a layer inside the network
  // sees for the bytes older than a few layers
  const int ROT = argc > 1 ? atoi(argv[1]) : 1;
  char *as[16], *rs[16], *ys[16];
  float* sink;
  for (int i = 0; i < ROT; ++i) {
    CHECK(hipMalloc(&as[i], out_bytes));
    CHECK(hipMalloc(&rs[i], out_bytes));
    CHECK(hipMalloc(&ys[i], out_bytes));
    CHECK(hipMemset(as[i], 1, out_bytes));
    CHECK(hipMemset(rs[i], 1, out_bytes));
  }
  CHECK(hipMalloc(&sink, 4));
  int rot = 0;
  char *a = as[0], *r = rs[0], *y = ys[0];
  printf("buffer sets: %d\n", ROT);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const long long units = out_bytes / 16;
  auto timeit = [&](const char* name, double bytes, auto launch) {
    auto next = [&] { rot = (rot + 1) % ROT; a = as[rot]; r = rs[rot]; y = ys[rot]; };
    for (int i = 0; i < 3; ++i) { next(); launch(); }
    CHECK(hipDeviceSynchronize());
    const int reps = 20;
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) { next(); launch(); }
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1000.0 / reps;
    printf("%-44s %8.1f us  %7.2f TB/s\n", name, us, bytes / us * 1e-6);
  };
  for (int blocks : {512, 4096}) {
    printf("---- grid %d x 256\n", blocks);
    char nm[128];
    snprintf(nm, sizeof nm, "read 100 MB (unroll 8)");
    timeit(nm, (double)out_bytes, [&] { hipLaunchKernelGGL(k_read<8>, dim3(blocks), dim3(256), 0, 0, (const f32x4*)a, sink, units); });
    snprintf(nm, sizeof nm, "write 100 MB");
    timeit(nm, (double)out_bytes, [&] { hipLaunchKernelGGL((k_write<1, false>), dim3(blocks), dim3(256), 0, 0, (f32x4*)y, units); });
    snprintf(nm, sizeof nm, "write 100 MB nontemporal");
    timeit(nm, (double)out_bytes, [&] { hipLaunchKernelGGL((k_write<1, true>), dim3(blocks), dim3(256), 0, 0, (f32x4*)y, units); });
    snprintf(nm, sizeof nm, "copy 100 -> 100 MB (unroll 4)");
    timeit(nm, 2.0 * out_bytes, [&] { hipLaunchKernelGGL((k_stream<4, false, false>), dim3(blocks), dim3(256), 0, 0, (const f32x4*)a, (const f32x4*)r, (f32x4*)y, units, 0); });
    snprintf(nm, sizeof nm, "copy 100 -> 100 MB (unroll 8, nt)");
    timeit(nm, 2.0 * out_bytes, [&] { hipLaunchKernelGGL((k_stream<8, false, true>), dim3(blocks), dim3(256), 0, 0, (const f32x4*)a, (const f32x4*)r, (f32x4*)y, units, 0); });
    snprintf(nm, sizeof nm, "c3-like: read 25 + 100, write 100 (unroll 4)");
    timeit(nm, 2.25 * out_bytes, [&] { hipLaunchKernelGGL((k_stream<4, true, false>), dim3(blocks), dim3(256), 0, 0, (const f32x4*)a, (const f32x4*)r, (f32x4*)y, units, 2); });
    snprintf(nm, sizeof nm, "c3-like (unroll 8, nt)");
    timeit(nm, 2.25 * out_bytes, [&] { hipLaunchKernelGGL((k_stream<8, true, true>), dim3(blocks), dim3(256), 0, 0, (const f32x4*)a, (const f32x4*)r, (f32x4*)y, units, 2); });
    snprintf(nm, sizeof nm, "c3-like, accumulator-shaped r/y access, 1 in flight");
    timeit(nm, 2.25 * out_bytes, [&] { hipLaunchKernelGGL(k_scatter<1>, dim3(blocks), dim3(512), 0, 0, (const f32x4*)a, (const f32x4*)r, (f32x4*)y, out_bytes / 512); });
    snprintf(nm, sizeof nm, "c3-like, accumulator-shaped r/y access, 2 in flight");
    timeit(nm, 2.25 * out_bytes, [&] { hipLaunchKernelGGL(k_scatter<2>, dim3(blocks), dim3(512), 0, 0, (const f32x4*)a, (const f32x4*)r, (f32x4*)y, out_bytes / 512); });
    snprintf(nm, sizeof nm, "c3-like, all three accumulator/fragment-shaped");
    timeit(nm, 2.25 * out_bytes, [&] { hipLaunchKernelGGL((k_scatter<1, true>), dim3(blocks), dim3(512), 0, 0, (const f32x4*)a, (const f32x4*)r, (f32x4*)y, out_bytes / 512); });
    snprintf(nm, sizeof nm, "stem.2-like: read 50, write 100 (unroll 4)");
    timeit(nm, 1.5 * out_bytes, [&] { hipLaunchKernelGGL((k_stream<4, false, false>), dim3(blocks), dim3(256), 0, 0, (const f32x4*)a, (const f32x4*)r, (f32x4*)y, units, 1); });
    snprintf(nm, sizeof nm, "c1-like: read 100, write 25");
    timeit(nm, 1.25 * out_bytes, [&] { hipLaunchKernelGGL(k_reduce4, dim3(blocks), dim3(256), 0, 0, (const f32x4*)a, (f32x4*)y, units / 4); });
  }
  return 0;
}
