// v_mfma_f32_32x32x16_bf16 vs v_mfma_f32_16x16x32_bf16 on RANDOM operands at equal flop (1 vs 2 instructions), with the LDS reads and VALU
// work of an attention-backward tile step beside them (per 32x32x16-equivalent: 2 ds_read_b128 + 3 VALU + 1 exp2), 1 and 2 waves per SIMD.
// The chip lowers its clock under load; the question is which shape it lets run faster.  Build on the GPU box: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
// operands: 0 c (16 regs), 1 d (16 regs), 2 r0, 3 r1 (LDS results), 4..7 x y z w, 8 a, 9 b, 10 addr
#define BIG0 "v_mfma_f32_32x32x16_bf16 %0, %8, %9, %0\n"
#define BIG1 "v_mfma_f32_32x32x16_bf16 %1, %8, %9, %1\n"
#define SIDE "ds_read_b128 %2, %10\n v_add_f32 %4, %4, %5\n v_exp_f32 %5, %5\n ds_read_b128 %3, %10 offset:4096\n v_mul_f32 %6, %6, %7\n v_max_f32 %7, %7, %4\n"
#define PROBE(NAME, BODY)                                                                                   \
  __global__ __launch_bounds__(512) void NAME(const float* rnd, float* sink, int iters) {                   \
    __shared__ char buf[32768];                                                                              \
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) ((float*)buf)[i] = rnd[i];                           \
    __syncthreads();                                                                                        \
    f32x16 c, d;                                                                                            \
    for (int i = 0; i < 16; ++i) { c[i] = rnd[threadIdx.x + i]; d[i] = rnd[1024 + threadIdx.x + i]; }         \
    bf16x8 a, b;                                                                                            \
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)rnd[2048 + 8 * threadIdx.x + i]; b[i] = (__bf16)rnd[6144 + 8 * (threadIdx.x & 63) + i]; } \
    f32x4 r0, r1;                                                                                           \
    float x = rnd[threadIdx.x], y = rnd[threadIdx.x + 1] * 1e-3f, z = 1.0f + 1e-4f * rnd[threadIdx.x + 2], w = rnd[threadIdx.x + 3]; \
    unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)buf + (threadIdx.x & 63) * 16; \
    for (int it = 0; it < iters; ++it) {                                                                    \
      asm volatile(REP16(BODY) "s_waitcnt lgkmcnt(0)\n" : "+v"(c), "+v"(d), "=v"(r0), "=v"(r1), "+v"(x), "+v"(y), "+v"(z), "+v"(w) : "v"(a), "v"(b), "v"(addr)); \
      for (int i = 0; i < 16; ++i) { c[i] *= 1e-3f; d[i] *= 1e-3f; }     /* keep the accumulators finite and changing */ \
    }                                                                                                       \
    sink[threadIdx.x] = c[0] + d[3] + r0[0] + r1[1] + x + y + z + w;                                         \
  }
// the 16x16x32 form of the same flop: 2 instructions per 32x32x16; accumulator tuples of 4 registers inside c / d are addressed as
// sub-tuples, which inline asm cannot name -> separate f32x4 operands
#define PROBE16(NAME, SIDE_)                                                                                \
  __global__ __launch_bounds__(512) void NAME(const float* rnd, float* sink, int iters) {                   \
    __shared__ char buf[32768];                                                                              \
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) ((float*)buf)[i] = rnd[i];                           \
    __syncthreads();                                                                                        \
    f32x4 c0, c1, c2, c3;                                                                                   \
    for (int i = 0; i < 4; ++i) { c0[i] = rnd[threadIdx.x + i]; c1[i] = rnd[64 + threadIdx.x + i]; c2[i] = rnd[128 + threadIdx.x + i]; c3[i] = rnd[192 + threadIdx.x + i]; } \
    bf16x8 a, b;                                                                                            \
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)rnd[2048 + 8 * threadIdx.x + i]; b[i] = (__bf16)rnd[6144 + 8 * (threadIdx.x & 63) + i]; } \
    f32x4 r0, r1;                                                                                           \
    float x = rnd[threadIdx.x], y = rnd[threadIdx.x + 1] * 1e-3f, z = 1.0f + 1e-4f * rnd[threadIdx.x + 2], w = rnd[threadIdx.x + 3]; \
    unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)buf + (threadIdx.x & 63) * 16; \
    for (int it = 0; it < iters; ++it) {                                                                    \
      asm volatile(REP16("v_mfma_f32_16x16x32_bf16 %0, %10, %11, %0\n v_mfma_f32_16x16x32_bf16 %1, %10, %11, %1\n" SIDE_                  \
                         "v_mfma_f32_16x16x32_bf16 %2, %10, %11, %2\n v_mfma_f32_16x16x32_bf16 %3, %10, %11, %3\n" SIDE_) "s_waitcnt lgkmcnt(0)\n" \
                   : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "=v"(r0), "=v"(r1), "+v"(x), "+v"(y), "+v"(z), "+v"(w) : "v"(a), "v"(b), "v"(addr)); \
      for (int i = 0; i < 4; ++i) { c0[i] *= 1e-3f; c1[i] *= 1e-3f; c2[i] *= 1e-3f; c3[i] *= 1e-3f; }            \
    }                                                                                                       \
    sink[threadIdx.x] = c0[0] + c1[3] + c2[1] + c3[2] + r0[0] + r1[1] + x + y + z + w;                        \
  }
#define SIDE16 "ds_read_b128 %4, %12\n v_add_f32 %6, %6, %7\n v_exp_f32 %7, %7\n ds_read_b128 %5, %12 offset:4096\n v_mul_f32 %8, %8, %9\n v_max_f32 %9, %9, %6\n"
PROBE(big_bare, BIG0 BIG1)
PROBE(big_side, BIG0 SIDE BIG1 SIDE)
PROBE16(small_bare, "")
PROBE16(small_side, SIDE16)
int main() {
  float *rnd, *sink;
  hipMalloc(&rnd, 4 * 16384); hipMalloc(&sink, 4 * 1024);
  static float h[16384];
  srand(1);
  for (int i = 0; i < 16384; ++i) h[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
  hipMemcpy(rnd, h, sizeof(h), hipMemcpyHostToDevice);
#define RUN(K, WAVES, NOTE)                                                                                \
  { hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms = 0, best = 1e9;                  \
    for (int r = 0; r < 4; ++r) { hipEventRecord(e0); hipLaunchKernelGGL(K, dim3(256), dim3(64 * WAVES), 0, 0, rnd, sink, 8192); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); if (r && ms < best) best = ms; } \
    double eq = 8192.0 * 16 * 2;                                                                             \
    printf("%-10s waves/SIMD %d: %7.2f ns per 32x32x16-equivalent per wave, chip %6.0f TFLOP/s  (%s)\n", #K, WAVES / 4, best * 1e6 / eq, 256.0 * WAVES * eq * 32768 / (best * 1e-3) / 1e12, NOTE); }
  for (int rep = 0; rep < 2; ++rep) {
    RUN(big_bare, 4, "32x32x16, MFMA only") RUN(small_bare, 4, "16x16x32, MFMA only")
    RUN(big_bare, 8, "32x32x16, MFMA only") RUN(small_bare, 8, "16x16x32, MFMA only")
    RUN(big_side, 4, "32x32x16 + 2 LDS reads, 3 VALU, 1 exp2 each") RUN(small_side, 4, "16x16x32 pair + the same")
    RUN(big_side, 8, "32x32x16 + 2 LDS reads, 3 VALU, 1 exp2 each") RUN(small_side, 8, "16x16x32 pair + the same")
  }
  return 0;
}
