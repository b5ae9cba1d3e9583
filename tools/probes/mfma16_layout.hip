// Checks the operand / result layout of v_mfma_f32_16x16x32_bf16 assumed by the generated streams:
//   A: lane l holds A[i = l % 16][k = 8 (l / 16) .. + 7],  B: lane l holds B[k = 8 (l / 16) .. + 7][j = l % 16],
//   D: lane l holds D[i = 4 (l / 16) + r][j = l % 16], r = 0..3.        Build on the GPU box: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
static float bf(float x) { unsigned u; std::memcpy(&u, &x, 4); u = (u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000u; float y; std::memcpy(&y, &u, 4); return y; }
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__global__ void k(const float* A, const float* B, float* D) {
  const int l = threadIdx.x, i16 = l % 16, g = l / 16;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)A[i16 * 32 + 8 * g + e]; b[e] = (__bf16)B[(8 * g + e) * 16 + i16]; }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  asm volatile("s_nop 7\n v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n s_nop 15\n s_nop 15\n s_nop 15" : "+v"(c) : "v"(a), "v"(b));
  for (int r = 0; r < 4; ++r) { D[(4 * g + r) * 16 + i16] = c[r]; D[256 + l * 4 + r] = c[r]; }
}
int main() {
  float hA[16 * 32], hB[32 * 16], hD[512], ref[256];
  srand(3);
  for (auto& x : hA) x = bf((float)rand() / RAND_MAX - 0.5f);
  for (auto& x : hB) x = bf((float)rand() / RAND_MAX - 0.5f);
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int kk = 0; kk < 32; ++kk) s += (double)hA[i * 32 + kk] * hB[kk * 16 + j]; ref[i * 16 + j] = (float)s; }
  float *dA, *dB, *dD;
  hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dD, sizeof(hD));
  hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
  double e = 0; for (int i = 0; i < 256; ++i) e = fmax(e, fabs(hD[i] - ref[i]));
  if (e >= 1e-4) for (int l = 0; l < 64; l += 5) for (int r = 0; r < 4; ++r) { int best = -1; for (int q = 0; q < 256; ++q) if (fabs(ref[q] - hD[256 + l * 4 + r]) < 1e-5) best = q; printf("lane %2d reg %d = %9.5f -> ref(i=%d, j=%d)\n", l, r, hD[256 + l * 4 + r], best < 0 ? -1 : best / 16, best < 0 ? -1 : best % 16); }
  printf("max |D - ref| = %g  -> layout %s\n", e, e < 1e-4 ? "CONFIRMED" : "WRONG");
  return e < 1e-4 ? 0 : 1;
}
