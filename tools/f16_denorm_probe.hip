// Does gfx950's matrix pipe keep fp16 SUBNORMAL inputs (needed by an fp16 hi/lo pair: lo = x - hi is subnormal for |x| < 2^-3)?
// One wave: D = A * B with A = 2^-20 (fp16 subnormal) in every element, B = 2^10 -> each product 2^-10, K = 32 -> 2^-5 exactly.
// Also: does (_Float16)float keep subnormals on conversion (FP16 denorm mode of the kernel)?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void probe(float a_val, float b_val, float* out) {
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)a_val; b[i] = (_Float16)b_val; }
  f4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  if (threadIdx.x == 0) { out[0] = c[0]; out[1] = (float)a[0]; out[2] = (float)b[0]; }
}
int main() {
  float* d; hipMalloc(&d, 64);
  const float cases[][2] = {{9.5367431640625e-07f, 1024.f}, {1024.f, 9.5367431640625e-07f}, {5.9604644775390625e-08f, 1.f}, {3.0517578125e-05f, 3.0517578125e-05f}, {0.5f, 2.f}};
  for (auto& c : cases) {
    probe<<<1, 64>>>(c[0], c[1], d);
    float h[3]; hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
    printf("a=%.9g (as f16 %.9g) b=%.9g (as f16 %.9g): mfma sum32 = %.9g expected %.9g %s\n", c[0], h[1], c[1], h[2], h[0], 32.0 * (double)h[1] * (double)h[2],
           h[0] == (float)(32.0 * (double)h[1] * (double)h[2]) ? "OK" : "MISMATCH");
  }
  return 0;
}
