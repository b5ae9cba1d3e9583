// Issue cost of single VALU instructions on gfx950, one wave per SIMD: cycles per instruction from s_memtime around 256 independent
// copies (4 chains x 64).  Build: hipcc --offload-arch=gfx950 -O3 tools/probes/valu_rate.hip -o gpurun_out/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))
#define PROBE(NAME, BODY)                                                                                  \
  __global__ void NAME(unsigned long long* out, float* sink) {                                             \
    float a = threadIdx.x * 1e-3f, b = a + 1.f, c = a + 2.f, d = a + 3.f;                                   \
    float2 p = {a, b}, q = {c, d};                                                                          \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                  \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                      \
    asm volatile(REP64(BODY) : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(p), "+v"(q));                          \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                  \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                      \
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                                        \
    sink[threadIdx.x] = a + b + c + d + p.x + p.y + q.x + q.y;                                              \
  }
PROBE(k_exp, "v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n")
PROBE(k_add, "v_add_f32 %0, %0, %1\n v_add_f32 %1, %1, %2\n v_add_f32 %2, %2, %3\n v_add_f32 %3, %3, %0\n")
PROBE(k_mul, "v_mul_f32 %0, %0, %1\n v_mul_f32 %1, %1, %2\n v_mul_f32 %2, %2, %3\n v_mul_f32 %3, %3, %0\n")
PROBE(k_cvt, "v_cvt_pk_bf16_f32 %0, %0, %1\n v_cvt_pk_bf16_f32 %1, %1, %2\n v_cvt_pk_bf16_f32 %2, %2, %3\n v_cvt_pk_bf16_f32 %3, %3, %0\n")
PROBE(k_pkmul, "v_pk_mul_f32 %4, %4, %5\n v_pk_mul_f32 %5, %5, %4\n v_pk_mul_f32 %4, %4, %5\n v_pk_mul_f32 %5, %5, %4\n")
PROBE(k_pkadd, "v_pk_add_f32 %4, %4, %5\n v_pk_add_f32 %5, %5, %4\n v_pk_add_f32 %4, %4, %5\n v_pk_add_f32 %5, %5, %4\n")
PROBE(k_max, "v_max_f32 %0, %0, %1\n v_max_f32 %1, %1, %2\n v_max_f32 %2, %2, %3\n v_max_f32 %3, %3, %0\n")
PROBE(k_fma, "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %2, %2, %3, %0\n v_fma_f32 %3, %3, %0, %1\n")
PROBE(k_exp16, "v_exp_f16 %0, %0\n v_exp_f16 %1, %1\n v_exp_f16 %2, %2\n v_exp_f16 %3, %3\n")
PROBE(k_mov, "v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0\n")
int main() {
  unsigned long long* out; float* sink;
  hipMalloc(&out, 8 * 1024); hipMalloc(&sink, 4 * 1024);
  unsigned long long h[4];
#define RUN(K, WAVES)                                                                         \
  for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(K, dim3(1), dim3(64 * WAVES), 0, 0, out, sink); hipDeviceSynchronize(); } \
  hipMemcpy(h, out, 8, hipMemcpyDeviceToHost);                                                \
  printf("%-8s waves/CU %2d: %6.2f memtime-ticks per instruction per wave\n", #K, WAVES, (double)h[0] / 256.0);
  RUN(k_mov, 4) RUN(k_add, 4) RUN(k_mul, 4) RUN(k_max, 4) RUN(k_fma, 4) RUN(k_cvt, 4) RUN(k_pkmul, 4) RUN(k_pkadd, 4) RUN(k_exp, 4) RUN(k_exp16, 4)
  RUN(k_add, 8) RUN(k_exp, 8) RUN(k_pkmul, 8)
  return 0;
}
