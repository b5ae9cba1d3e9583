// conv.hip — causal 1-D convolutions of the VQ-VAE ("SoundStream") tokenizer, channels-last, as GEMMs:
//   fk_im2col1d   x [B, T, Cin] -> cols [B*Tout, K*Cin]: cols[b, t, k, :] = x[b, t*stride + k*dil - pad, :] (0 outside [0, T))
//                 pad = dil*(K-1) is the CausalConv1d left padding (models/vq_brain.py:22-28); Tout = (T - 1) / stride + 1.
//                 y = cols * W'^T with W'[o, k*Cin + c] = W[o, c, k] is nn.Conv1d; CausalConvTranspose1d(kernel 2s, stride s,
//                 :31-45) is the same with K = 2 taps (x[j-1], x[j]) and N = s*Cout phase-major outputs viewed as [B, s*T, Cout].
//   fk_col2im1d   the adjoint (gather form, no atomics): dx[b, tau, :] = sum_{t,k : t*stride + k*dil - pad = tau} dcols[b, t, k, :]
//   fk_elu_fwd/bwd  nn.ELU(alpha = 1)                                        (:57,72-77,...)
//   fk_argmax_rows  index of the row maximum (first on ties) — nearest code of the cosine-similarity VQ lookup
// All HBM-bound, 16-byte vectorised along the channel dimension.
#include "fk_common.h"

namespace {

constexpr int TPB = 256;
inline unsigned grid_for(int64_t work, int64_t cap = 1 << 20) {
  int64_t b = fk_cdiv(work, TPB);
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (unsigned)b;
}
template <typename T> struct VW { static constexpr int N = 16 / sizeof(T); };

template <typename T>
__global__ void im2col1d_kernel(const T* x, T* cols, int64_t B, int64_t Tin, int64_t Tout, int Cin, int K, int stride, int dil, int pad) {
  constexpr int N = VW<T>::N;
  const int cv = Cin / N;
  const int64_t total = B * Tout * K * cv;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % cv);
    const int k = (int)((i / cv) % K);
    const int64_t t = (i / ((int64_t)cv * K)) % Tout, b = i / ((int64_t)cv * K * Tout);
    const int64_t ts = t * stride + (int64_t)k * dil - pad;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (ts >= 0 && ts < Tin) v = *reinterpret_cast<const u32x4*>(x + ((b * Tin + ts) * Cin + (int64_t)c * N));
    *reinterpret_cast<u32x4*>(cols + (((b * Tout + t) * K + k) * Cin + (int64_t)c * N)) = v;
  }
}

template <typename T>
__global__ void col2im1d_kernel(const T* dcols, T* dx, int64_t B, int64_t Tin, int64_t Tout, int Cin, int K, int stride, int dil, int pad) {
  constexpr int N = VW<T>::N;
  const int cv = Cin / N;
  const int64_t total = B * Tin * cv;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % cv);
    const int64_t tau = (i / cv) % Tin, b = i / ((int64_t)cv * Tin);
    float acc[N];
#pragma unroll
    for (int e = 0; e < N; ++e) acc[e] = 0.0f;
    for (int k = 0; k < K; ++k) {
      const int64_t num = tau + pad - (int64_t)k * dil;
      if (num < 0 || num % stride != 0) continue;
      const int64_t t = num / stride;
      if (t >= Tout) continue;
      const T* p = dcols + (((b * Tout + t) * K + k) * Cin + (int64_t)c * N);
      if constexpr (sizeof(T) == 2) {
        bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
        for (int e = 0; e < N; ++e) acc[e] += (float)v[e];
      } else {
        f32x4 v = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
        for (int e = 0; e < N; ++e) acc[e] += v[e];
      }
    }
    T* q = dx + ((b * Tin + tau) * Cin + (int64_t)c * N);
    if constexpr (sizeof(T) == 2) {
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < N; ++e) o[e] = (bf16_t)acc[e];
      *reinterpret_cast<bf16x8*>(q) = o;
    } else {
      *reinterpret_cast<f32x4*>(q) = f32x4{acc[0], acc[1], acc[2], acc[3]};
    }
  }
}

template <typename T>
__global__ void elu_fwd_kernel(const T* x, T* y, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = to_f32<T>(x[i]);
    y[i] = from_f32<T>(v > 0.0f ? v : expm1f(v));
  }
}
template <typename T>
__global__ void elu_bwd_kernel(const T* x, const T* dy, T* dx, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = to_f32<T>(x[i]);
    dx[i] = from_f32<T>(to_f32<T>(dy[i]) * (v > 0.0f ? 1.0f : __expf(v)));
  }
}

// one wave per row: lanes stride the columns keeping (max, first index), then a wave reduction
template <typename T>
__global__ void argmax_rows_kernel(const T* x, int64_t ld, int64_t* idx, int64_t rows, int cols) {
  const int lane = threadIdx.x & 63;
  const int64_t row = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* p = x + row * ld;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int c = lane; c < cols; c += 64) {
    const float v = to_f32<T>(p[c]);
    if (v > best) { best = v; bi = c; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  if (lane == 0) idx[row] = bi == 0x7fffffff ? 0 : bi;
}

}  // namespace

extern "C" {

#define FK_CONV_CHECK(name)                                                                                                    \
  FK_CHECK_ARG(dtype == FK_F32 || dtype == FK_BF16, name ": bad dtype %d", dtype);                                             \
  const int vec = dtype == FK_BF16 ? 8 : 4;                                                                                    \
  FK_CHECK_ARG(B > 0 && T > 0 && Cin > 0 && K > 0 && stride > 0 && dil > 0 && Cin % vec == 0, name ": bad shape (Cin %% %d)", vec); \
  const int64_t Tout = (T - 1) / stride + 1, pad = dil * (K - 1)

int fk_im2col1d(const void* x, void* cols, int64_t B, int64_t T, int64_t Cin, int64_t K, int64_t stride, int64_t dil, int dtype, void* stream) {
  FK_CONV_CHECK("fk_im2col1d");
  FK_CHECK_ARG(x && cols && ((uintptr_t)x & 15) == 0 && ((uintptr_t)cols & 15) == 0, "fk_im2col1d: pointers must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const int64_t work = B * Tout * K * (Cin / vec);
  if (dtype == FK_BF16) hipLaunchKernelGGL(im2col1d_kernel<bf16_t>, dim3(grid_for(work, 65536)), dim3(TPB), 0, s, (const bf16_t*)x, (bf16_t*)cols, B, T, Tout, (int)Cin, (int)K, (int)stride, (int)dil, (int)pad);
  else hipLaunchKernelGGL(im2col1d_kernel<float>, dim3(grid_for(work, 65536)), dim3(TPB), 0, s, (const float*)x, (float*)cols, B, T, Tout, (int)Cin, (int)K, (int)stride, (int)dil, (int)pad);
  FK_CHECK_LAUNCH("fk_im2col1d");
  return FK_OK;
}

int fk_col2im1d(const void* dcols, void* dx, int64_t B, int64_t T, int64_t Cin, int64_t K, int64_t stride, int64_t dil, int dtype, void* stream) {
  FK_CONV_CHECK("fk_col2im1d");
  FK_CHECK_ARG(dcols && dx && ((uintptr_t)dx & 15) == 0 && ((uintptr_t)dcols & 15) == 0, "fk_col2im1d: pointers must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const int64_t work = B * T * (Cin / vec);
  if (dtype == FK_BF16) hipLaunchKernelGGL(col2im1d_kernel<bf16_t>, dim3(grid_for(work, 65536)), dim3(TPB), 0, s, (const bf16_t*)dcols, (bf16_t*)dx, B, T, Tout, (int)Cin, (int)K, (int)stride, (int)dil, (int)pad);
  else hipLaunchKernelGGL(col2im1d_kernel<float>, dim3(grid_for(work, 65536)), dim3(TPB), 0, s, (const float*)dcols, (float*)dx, B, T, Tout, (int)Cin, (int)K, (int)stride, (int)dil, (int)pad);
  FK_CHECK_LAUNCH("fk_col2im1d");
  return FK_OK;
}

int fk_elu_fwd(const void* x, void* y, int64_t n, int dtype, void* stream) {
  FK_CHECK_ARG((dtype == FK_F32 || dtype == FK_BF16) && x && y && n > 0, "fk_elu_fwd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  if (dtype == FK_BF16) hipLaunchKernelGGL(elu_fwd_kernel<bf16_t>, dim3(grid_for(n, 65536)), dim3(TPB), 0, s, (const bf16_t*)x, (bf16_t*)y, n);
  else hipLaunchKernelGGL(elu_fwd_kernel<float>, dim3(grid_for(n, 65536)), dim3(TPB), 0, s, (const float*)x, (float*)y, n);
  FK_CHECK_LAUNCH("fk_elu_fwd");
  return FK_OK;
}
int fk_elu_bwd(const void* x, const void* dy, void* dx, int64_t n, int dtype, void* stream) {
  FK_CHECK_ARG((dtype == FK_F32 || dtype == FK_BF16) && x && dy && dx && n > 0, "fk_elu_bwd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  if (dtype == FK_BF16) hipLaunchKernelGGL(elu_bwd_kernel<bf16_t>, dim3(grid_for(n, 65536)), dim3(TPB), 0, s, (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)dx, n);
  else hipLaunchKernelGGL(elu_bwd_kernel<float>, dim3(grid_for(n, 65536)), dim3(TPB), 0, s, (const float*)x, (const float*)dy, (float*)dx, n);
  FK_CHECK_LAUNCH("fk_elu_bwd");
  return FK_OK;
}

int fk_argmax_rows(const void* x, int64_t ld, int64_t* idx, int64_t rows, int64_t cols, int dtype, void* stream) {
  FK_CHECK_ARG((dtype == FK_F32 || dtype == FK_BF16) && x && idx && rows > 0 && cols > 0 && cols < (1LL << 31), "fk_argmax_rows: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const unsigned nb = (unsigned)fk_cdiv(rows, 4);
  if (dtype == FK_BF16) hipLaunchKernelGGL(argmax_rows_kernel<bf16_t>, dim3(nb), dim3(TPB), 0, s, (const bf16_t*)x, ld, idx, rows, (int)cols);
  else hipLaunchKernelGGL(argmax_rows_kernel<float>, dim3(nb), dim3(TPB), 0, s, (const float*)x, ld, idx, rows, (int)cols);
  FK_CHECK_LAUNCH("fk_argmax_rows");
  return FK_OK;
}

}  // extern "C"
