// Probe: do MFMA (one wave) and VALU / transcendental work (another wave on the same SIMD) overlap on gfx950?
// Block = 512 threads (8 waves, 2 per SIMD). mode 0: all waves MFMA; 1: all waves VALU; 2: waves 0-3 MFMA, waves 4-7 VALU;
// 3: every wave alternates MFMA/VALU instruction-by-instruction (same wave interleave).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ __launch_bounds__(512) void probe(float* out, int iters, int mode) {
  const int wave = threadIdx.x >> 6;
  f32x16 acc0 = {}, acc1 = {};
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
  float v0 = threadIdx.x * 0.01f, v1 = 1.0f, v2 = 0.5f, v3 = 0.25f;
  const bool do_mfma = mode == 0 || (mode == 2 && wave < 4) || mode == 3;
  const bool do_valu = mode == 1 || (mode == 2 && wave >= 4) || mode == 3;
  for (int it = 0; it < iters; ++it) {
    if (do_mfma) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc1, 0, 0, 0);
      }
    }
    if (do_valu) {
#pragma unroll
      for (int k = 0; k < 32; ++k) {       // 32 x (4 independent fma chains) = 128 VALU ops
        v0 = __builtin_fmaf(v0, 1.0001f, 0.5f); v1 = __builtin_fmaf(v1, 0.9999f, 0.25f);
        v2 = __builtin_fmaf(v2, 1.0002f, 0.125f); v3 = __builtin_fmaf(v3, 0.9998f, 0.0625f);
      }
    }
  }
  float s = v0 + v1 + v2 + v3;
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}
int main() {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int mode = 0; mode < 4; ++mode) {
    probe<<<256, 512>>>(out, 100, mode);
    hipDeviceSynchronize();
    hipEventRecord(e0); probe<<<256, 512>>>(out, iters, mode); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("mode %d: %.3f ms  (%.1f ns / iteration)\n", mode, ms, ms * 1e6 / iters);
  }
  return 0;
}
