// How much VALU / transcendental work hides behind v_mfma_f32_32x32x16_bf16 on gfx950, at 1 and 2 waves per SIMD: s_memtime ticks per
// MFMA for streams "1 MFMA + k VALU" (two independent accumulators, VALU chains independent of the MFMAs), every CU busy.
// Build on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_mix.hip -o /tmp/mfma_mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP32(x) REP16(x) REP16(x)
#define M0 "v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n"
#define M1 "v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n"
#define V "v_add_f32 %4, %4, %5\n v_mul_f32 %5, %5, %6\n v_max_f32 %6, %6, %7\n v_add_f32 %7, %7, %4\n"
#define E "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
#define PROBE(NAME, BODY)                                                                                  \
  __global__ __launch_bounds__(512) void NAME(unsigned long long* out, float* sink, int iters) {                      \
    f32x16 c0, c1;                                                                                          \
    for (int i = 0; i < 16; ++i) { c0[i] = 0.f; c1[i] = 0.f; }                                               \
    bf16x8 a, b;                                                                                            \
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 1e-3f); b[i] = (__bf16)1.0f; }               \
    float x = threadIdx.x * 1e-3f, y = x + 1.f, z = x + 2.f, w = x + 3.f;                                    \
    __syncthreads();                                                                                        \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                  \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                      \
    for (int it = 0; it < iters; ++it)                                                                         \
      asm volatile(REP32(BODY) : "+v"(c0), "+v"(c1) : "v"(a), "v"(b), "v"(x), "v"(y), "v"(z), "v"(w));        \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                  \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                      \
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                                        \
    sink[threadIdx.x] = c0[0] + c1[3] + x + y + z + w;                                                      \
  }
PROBE(k_m, M0 M1)
PROBE(k_m_v4, M0 V M1 V)
PROBE(k_m_v8, M0 V V M1 V V)
PROBE(k_m_v12, M0 V V V M1 V V V)
PROBE(k_m_e4, M0 E M1 E)
PROBE(k_m_e2v4, M0 "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n" V M1 "v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n" V)
PROBE(k_v8, V V V V)
int main() {
  unsigned long long* out; float* sink;
  hipMalloc(&out, 8 * 4096); hipMalloc(&sink, 4 * 1024);
  static unsigned long long h[256];
#define RUN(K, WAVES, NM, NV)                                                                         \
  { hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms = 0;                                                   \
    for (int r = 0; r < 3; ++r) { hipEventRecord(e0); hipLaunchKernelGGL(K, dim3(256), dim3(64 * WAVES), 0, 0, out, sink, 4096); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); } \
    hipMemcpy(h, out, 8 * 256, hipMemcpyDeviceToHost);                                                \
    double s = 0; for (int i = 0; i < 256; ++i) s += h[i]; s /= 256;                                  \
    double nm = 4096.0 * 32 * NM;                                                                        \
    printf("%-10s waves/SIMD %d: %7.1f ticks, %7.2f ns per MFMA per wave (%d VALU per MFMA); tick rate %.0f MHz; chip %.0f TFLOP/s\n", #K, WAVES / 4, s / nm, ms * 1e6 / nm, NV, s / (ms * 1e3), 256.0 * WAVES * nm * 32768 / (ms * 1e-3) / 1e12); }
  RUN(k_m, 4, 2, 0) RUN(k_m, 8, 2, 0)
  RUN(k_m_v4, 4, 2, 4) RUN(k_m_v4, 8, 2, 4)
  RUN(k_m_v8, 4, 2, 8) RUN(k_m_v8, 8, 2, 8)
  RUN(k_m_v12, 4, 2, 12) RUN(k_m_v12, 8, 2, 12)
  RUN(k_m_e4, 4, 2, 4) RUN(k_m_e4, 8, 2, 4)
  RUN(k_m_e2v4, 4, 2, 6) RUN(k_m_e2v4, 8, 2, 6)
  RUN(k_v8, 4, 1, 16) RUN(k_v8, 8, 1, 16)
  return 0;
}
