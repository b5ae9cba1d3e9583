// A stand-in for a collective's channels: `wgs` workgroups of 256 threads that each hold `lds_bytes` of LDS and keep their CU slot for
// `ticks` ticks of the 100-MHz wall clock (bounded spin), doing no memory traffic.  Built on the GPU box by tools/occupant_probe.py:
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/probes/occupant.hip -o /tmp/libfk_occupant.so
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void occupant_kernel(long long ticks, int* sink) {
  extern __shared__ int lds[];
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  const uint64_t t0 = wall_clock64();
  int acc = 0;
  for (int it = 0; it < (1 << 24); ++it) {                    // bounded: every wave leaves after at most 2^24 polls
    if ((long long)(wall_clock64() - t0) >= ticks) break;
    __builtin_amdgcn_s_sleep(32);
    acc += lds[(threadIdx.x + it) & 255];
  }
  if (acc == 0x7fffffff) sink[0] = acc;
}

extern "C" int fk_occupy(int wgs, int lds_bytes, long long ticks, int* sink, void* stream) {
  if (lds_bytes > 65536)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(occupant_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  hipLaunchKernelGGL(occupant_kernel, dim3(wgs), dim3(256), (size_t)lds_bytes, (hipStream_t)stream, ticks, sink);
  return (int)hipGetLastError();
}
