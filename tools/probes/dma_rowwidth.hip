// How fast can eight waves per CU stream a row-major [M][K] bf16 operand into LDS by LDS-DMA, as a function of the bytes each 1-KiB
// request takes from one row (128 B x 8 rows, 256 B x 4 rows, 512 B x 2 rows, 1024 B x 1 row)?  Same cadence as the NT GEMM ring: 3 slots
// of 48 KiB, requests of the stage after next issued after the barrier, counted vmcnt wait.  One 512-thread workgroup per CU, each walks
// row panels of 384 rows (48 KiB per stage = 384 rows x 128 B, or fewer rows x wider).  Build on the GPU box: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;
template <int W>   // bytes per row per request
__global__ __launch_bounds__(512) void stream(const char* A, long ldab, int npanels, int K2 /* bytes per row */, float* sink, int shared) {
  extern __shared__ char smem[];
  constexpr int RPR = 1024 / W;            // rows per request
  constexpr int ROWS = 48 * RPR;           // rows per 48-KiB stage
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nst = K2 / W;                  // stages per panel
  long total = (long)((npanels - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x) * nst;
  long issued = 0, done = 0;
  auto issue = [&](long g, int slot) {
    long pan = blockIdx.x + (g / nst) * gridDim.x;
    if (shared > 0) pan %= shared;                        // every workgroup walks the same few panels: the stream comes out of L2
    const int st = (int)(g % nst);
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int piece = wave * 6 + j;                      // 48 pieces per stage
      const int row = piece * RPR + lane / (W / 16);
      const char* src = A + (pan * ROWS + row) * ldab + (long)st * W + (lane % (W / 16)) * 16;
      __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)(smem + slot * 49152 + piece * 1024), 16, 0, 0);
    }
  };
  if (total > 0) issue(0, 0);
  if (total > 1) issue(1, 1);
  issued = total > 1 ? 2 : total;
  float acc = 0.f;
  for (long g = 0; g < total; ++g) {
    if (issued - g >= 2) asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if (issued < total) { issue(issued, (int)(issued % 3)); ++issued; }
    acc += *reinterpret_cast<float*>(smem + (g % 3) * 49152 + threadIdx.x * 4);     // touch the stage
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc == 123.456f) sink[threadIdx.x] = acc;
}
int main() {
  const long M = 196608, K = 3072;                       // the d_up operand of cfg2: 1.2 GB
  char* A; float* sink;
  hipMalloc(&A, M * K * 2 + 4096); hipMalloc(&sink, 4096);
  hipMemset(A, 1, M * K * 2);
#define RUN(W)                                                                                        \
  { hipFuncSetAttribute(reinterpret_cast<const void*>(stream<W>), hipFuncAttributeMaxDynamicSharedMemorySize, 147456);  \
    const int rows = 48 * (1024 / W), npan = (int)(M / rows);                                             \
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms = 0, best = 1e9;                  \
    for (int r = 0; r < 4; ++r) { hipEventRecord(e0); hipLaunchKernelGGL(stream<W>, dim3(256), dim3(512), 147456, 0, A, K * 2, npan, (int)(K * 2), sink, SHARED); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); if (r && ms < best) best = ms; } \
    printf("%s %4d bytes per row and request (%d rows per request, %4d-row panels): %7.1f us, %5.2f TB/s = %5.1f GB/s per CU\n", SHARED ? "L2 " : "HBM", W, 1024 / W, rows, best * 1e3, (double)npan * rows * K * 2 / (best * 1e-3) / 1e12, (double)npan * rows * K * 2 / (best * 1e-3) / 256e9); }
  { const int SHARED = 0; RUN(128) RUN(256) RUN(512) RUN(1024) }
  { const int SHARED = 1; RUN(128) RUN(1024) }       // one 2.3-MB panel read by everybody
  return 0;
}
