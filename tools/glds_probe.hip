// Probe: does an out-of-range lane of `buffer_load_dwordx4 ... lds` write ZEROS to LDS (needed for conv zero padding)?
// build: hipcc --offload-arch=gfx950 -O3 tools/glds_probe.hip -o gpurun_out/glds_probe && ./gpurun_out/glds_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
__global__ __launch_bounds__(256) void k(const unsigned* __restrict__ src, unsigned* dst, int nbytes) {
  __shared__ __attribute__((aligned(16))) char smem[4096];
  for (int i = threadIdx.x; i < 1024; i += 256) ((unsigned*)smem)[i] = 0xABABABABu;
  __syncthreads();
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
  const int wv = threadIdx.x >> 6;
  unsigned voff = threadIdx.x * 16u;
  if ((threadIdx.x % 5) == 3) voff = 0x80000000u;       // far out of range
  if ((threadIdx.x % 7) == 6) voff = nbytes - 8;        // straddles the end
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(smem + wv * 1024), 16, voff, 0, 0, 0);
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += 256) dst[i] = ((unsigned*)smem)[i];
}
int main() {
  const int n = 4096;  // bytes
  std::vector<unsigned> h(n / 4);
  for (int i = 0; i < n / 4; ++i) h[i] = 0x10000000u + i;
  unsigned *d, *o;
  hipMalloc(&d, n); hipMalloc(&o, 4096);
  hipMemcpy(d, h.data(), n, hipMemcpyHostToDevice);
  k<<<1, 256>>>(d, o, n);
  std::vector<unsigned> r(1024);
  hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost);
  int bad = 0, zero_oob = 0, stale_oob = 0, other = 0;
  for (int t = 0; t < 256; ++t) {
    const bool far = (t % 5) == 3, edge = (t % 7) == 6 && !far ? true : ((t % 7) == 6);
    for (int j = 0; j < 4; ++j) {
      unsigned v = r[t * 4 + j];
      if (!far && (t % 7) != 6) { if (v != 0x10000000u + t * 4 + j) ++bad; }
      else if (v == 0) ++zero_oob; else if (v == 0xABABABABu) ++stale_oob; else ++other;
    }
  }
  printf("in-range mismatches=%d  oob dwords: zero=%d stale=%d other=%d\n", bad, zero_oob, stale_oob, other);
  for (int t : {3, 6, 13}) printf("lane %d: %08x %08x %08x %08x\n", t, r[t*4], r[t*4+1], r[t*4+2], r[t*4+3]);
  return 0;
}
