// Do LDS reads hide behind v_mfma_f32_32x32x16_bf16 on gfx950?  ns per MFMA for streams "1 MFMA + k LDS reads", accumulators in VGPRs
// ("v") or AGPRs ("a"), 1 and 2 waves per SIMD, every CU busy.  Build on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_lds.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define M0 "v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n"
#define M1 "v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n"
#define L128a "ds_read_b128 %4, %8\n"
#define L128b "ds_read_b128 %5, %8 offset:4096\n"
#define L64a "ds_read_b64_tr_b16 %6, %8 offset:8192\n"
#define L64b "ds_read_b64_tr_b16 %7, %8 offset:12288\n"
#define W "s_waitcnt lgkmcnt(0)\n"
#define PROBE(NAME, ACC, BODY)                                                                              \
  __global__ __launch_bounds__(512) void NAME(float* sink, int iters) {                                     \
    __shared__ char buf[32768];                                                                              \
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) ((float*)buf)[i] = i;                                \
    __syncthreads();                                                                                        \
    f32x16 c0, c1;                                                                                          \
    for (int i = 0; i < 16; ++i) { c0[i] = 0.f; c1[i] = 0.f; }                                               \
    bf16x8 a, b;                                                                                            \
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 1e-3f); b[i] = (__bf16)1.0f; }               \
    f32x4 r0, r1; f32x2 r2, r3;                                                                             \
    unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)buf + (threadIdx.x & 63) * 16; \
    for (int it = 0; it < iters; ++it)                                                                      \
      asm volatile(REP16(BODY) W : "+" ACC(c0), "+" ACC(c1), "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(a), "v"(b), "v"(addr)); \
    sink[threadIdx.x] = c0[0] + c1[3] + r0[0] + r1[1] + r2[0] + r3[1];                                       \
  }
// operand numbering: 0 c0, 1 c1, 2 r0, 3 r1, 4 r2, 5 r3, 6 a, 7 b, 8 addr  -> redefine the macros with these numbers
#undef M0
#undef M1
#undef L128a
#undef L128b
#undef L64a
#undef L64b
#define M0 "v_mfma_f32_32x32x16_bf16 %0, %6, %7, %0\n"
#define M1 "v_mfma_f32_32x32x16_bf16 %1, %6, %7, %1\n"
#define L128a "ds_read_b128 %2, %8\n"
#define L128b "ds_read_b128 %3, %8 offset:4096\n"
#define L64a "ds_read_b64_tr_b16 %4, %8 offset:8192\n"
#define L64b "ds_read_b64_tr_b16 %5, %8 offset:12288\n"
PROBE(v_m, "v", M0 M1)
PROBE(a_m, "a", M0 M1)
PROBE(v_m_l1, "v", M0 L128a M1 L128b)
PROBE(a_m_l1, "a", M0 L128a M1 L128b)
PROBE(v_m_l2, "v", M0 L128a L128b M1 L128a L128b)
PROBE(a_m_l2, "a", M0 L128a L128b M1 L128a L128b)
PROBE(v_m_t2, "v", M0 L64a L64b M1 L64a L64b)
PROBE(a_m_t2, "a", M0 L64a L64b M1 L64a L64b)
PROBE(v_l2, "v", L128a L128b L128a L128b)
int main() {
  float* sink;
  hipMalloc(&sink, 4 * 1024);
#define RUN(K, WAVES, NM, NOTE)                                                                         \
  { hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms = 0;                                                   \
    for (int r = 0; r < 3; ++r) { hipEventRecord(e0); hipLaunchKernelGGL(K, dim3(256), dim3(64 * WAVES), 0, 0, sink, 8192); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); } \
    double nm = 8192.0 * 16 * NM;                                                                        \
    printf("%-8s waves/SIMD %d: %7.2f ns per MFMA per wave  (%s)\n", #K, WAVES / 4, ms * 1e6 / nm, NOTE); }
  RUN(v_m, 4, 2, "MFMA only, acc in VGPRs") RUN(v_m, 8, 2, "MFMA only, acc in VGPRs")
  RUN(a_m, 4, 2, "MFMA only, acc in AGPRs") RUN(a_m, 8, 2, "MFMA only, acc in AGPRs")
  RUN(v_m_l1, 4, 2, "+1 ds_read_b128 per MFMA, VGPR acc") RUN(v_m_l1, 8, 2, "+1 ds_read_b128 per MFMA, VGPR acc")
  RUN(a_m_l1, 4, 2, "+1 ds_read_b128 per MFMA, AGPR acc") RUN(a_m_l1, 8, 2, "+1 ds_read_b128 per MFMA, AGPR acc")
  RUN(v_m_l2, 4, 2, "+2 ds_read_b128 per MFMA, VGPR acc") RUN(v_m_l2, 8, 2, "+2 ds_read_b128 per MFMA, VGPR acc")
  RUN(a_m_l2, 4, 2, "+2 ds_read_b128 per MFMA, AGPR acc") RUN(a_m_l2, 8, 2, "+2 ds_read_b128 per MFMA, AGPR acc")
  RUN(v_m_t2, 4, 2, "+2 ds_read_b64_tr_b16 per MFMA, VGPR acc") RUN(v_m_t2, 8, 2, "+2 ds_read_b64_tr_b16 per MFMA, VGPR acc")
  RUN(a_m_t2, 4, 2, "+2 ds_read_b64_tr_b16 per MFMA, AGPR acc") RUN(a_m_t2, 8, 2, "+2 ds_read_b64_tr_b16 per MFMA, AGPR acc")
  RUN(v_l2, 4, 2, "no MFMA: 2 ds_read_b128 per slot") RUN(v_l2, 8, 2, "no MFMA: 2 ds_read_b128 per slot")
  return 0;
}
