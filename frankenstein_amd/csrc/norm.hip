// norm.hip — LayerNorm / RMSNorm forward + backward (HBM-bound).
//
// Replaces nn.LayerNorm(dim) (models/brainformer.py:237,239,252,254,287,500), F.layer_norm
// (models/gpt2_model.py:27) and RMSNorm (models/brainformer.py:221-232) of the reference.
// One wave per row (64 lanes x 16-byte vectors; the row stays in L1 between the three sweeps),
// wave-shuffle reductions, fp32 statistics saved for the backward.  The backward fuses the
// residual-branch gradient add (dx = dres + norm_bwd(dy)).  dgamma/dbeta are column reductions
// through a [chunks, 2, dim] fp32 workspace (deterministic).
#include "fk_common.h"

namespace {

#ifndef FK_NT_STORES_NORM
#define FK_NT_STORES_NORM 0
#endif
template <typename T> struct VecIO;
template <> struct VecIO<bf16_t> {
  static constexpr int N = 8;
  FK_DEV static void load(const bf16_t* p, float (&v)[8]) {
    bf16x8 a = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)a[i];
  }
  FK_DEV static void store(bf16_t* p, const float (&v)[8]) {
    bf16x8 a;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (bf16_t)v[i];
    fk_st<FK_NT_STORES_NORM != 0>(reinterpret_cast<bf16x8*>(p), a);
  }
};
template <> struct VecIO<float> {
  static constexpr int N = 4;
  FK_DEV static void load(const float* p, float (&v)[4]) {
    f32x4 a = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = a[i];
  }
  FK_DEV static void store(float* p, const float (&v)[4]) {
    f32x4 a;
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = v[i];
    *reinterpret_cast<f32x4*>(p) = a;
  }
};

template <typename T>
__global__ __launch_bounds__(256) void norm_fwd_kernel(const T* x, const float* gamma, const float* beta, T* y,
                                                        float* mean, float* rstd, int64_t rows, int dim, float eps, int kind) {
  constexpr int N = VecIO<T>::N;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t wstride = (int64_t)gridDim.x * 4;
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += wstride) {
    const T* xr = x + row * dim;
    float s = 0.0f;
    float v[N];
    if (kind == FK_NORM_LAYER) {
      for (int c = lane * N; c < dim; c += 64 * N) {
        VecIO<T>::load(xr + c, v);
#pragma unroll
        for (int i = 0; i < N; ++i) s += v[i];
      }
    }
    const float mu = kind == FK_NORM_LAYER ? wave_sum(s) / dim : 0.0f;
    float q = 0.0f;
    for (int c = lane * N; c < dim; c += 64 * N) {
      VecIO<T>::load(xr + c, v);
#pragma unroll
      for (int i = 0; i < N; ++i) q += (v[i] - mu) * (v[i] - mu);
    }
    const float rs = rsqrtf(wave_sum(q) / dim + eps);
    if (lane == 0) {
      if (mean) mean[row] = mu;
      rstd[row] = rs;
    }
    T* yr = y + row * dim;
    for (int c = lane * N; c < dim; c += 64 * N) {
      VecIO<T>::load(xr + c, v);
      float o[N];
#pragma unroll
      for (int i = 0; i < N; ++i) {
        float t = (v[i] - mu) * rs;
        if (kind == FK_NORM_RMS && sizeof(T) == 2) t = (float)(bf16_t)t;   // reference: _norm(x.float()).type_as(x) * weight
        o[i] = t * gamma[c + i] + (beta ? beta[c + i] : 0.0f);
      }
      VecIO<T>::store(yr + c, o);
    }
  }
}

// Fast path: LPR lanes per row (16 -> 4 rows per wave, or 64), the row lives in registers (read once), exact two-pass
// variance, LPR-wide shuffle reductions.  lane-in-row l owns columns l*N + it*LPR*N.
template <int LPR> FK_DEV float row_sum(float v) {
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
template <typename T, int LPR, int MAXIT>
__global__ __launch_bounds__(256) void norm_fwd_fast_kernel(const T* x, const float* gamma, const float* beta, T* y,
                                                             float* mean, float* rstd, int64_t rows, int dim, float eps, int kind) {
  constexpr int N = VecIO<T>::N, RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l = lane % LPR, sub = lane / LPR;
  // gamma / beta: dim % N == 0 on this path, so a whole N-vector is in or out of range: 16-byte loads, one predicate per vector
  // (48 branchy scalar loads per wave made the prologue as long as the three row iterations it served)
  float gm[MAXIT][N], bt[MAXIT][N];
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) {
    const int c = l * N + it * LPR * N;
    const bool in = c < dim;
#pragma unroll
    for (int i = 0; i < N; i += 4) {
      const f32x4 g4 = in ? *reinterpret_cast<const f32x4*>(gamma + c + i) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      const f32x4 b4 = (in && beta) ? *reinterpret_cast<const f32x4*>(beta + c + i) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int e = 0; e < 4; ++e) { gm[it][i + e] = g4[e]; bt[it][i + e] = b4[e]; }
    }
  }
  const float inv = 1.0f / dim;
  const int64_t stride = (int64_t)gridDim.x * 4 * RPW;
  for (int64_t base = ((int64_t)blockIdx.x * 4 + wave) * RPW; base < rows; base += stride) {   // wave-uniform trip count
    const int64_t row = base + sub;
    const bool ok = row < rows;
    float v[MAXIT][N];
    float s = 0.0f;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      const int c = l * N + it * LPR * N;
      if (ok && c < dim) VecIO<T>::load(x + row * dim + c, v[it]);
      else {
#pragma unroll
        for (int i = 0; i < N; ++i) v[it][i] = 0.0f;
      }
#pragma unroll
      for (int i = 0; i < N; ++i) s += v[it][i];
    }
    const float mu = kind == FK_NORM_LAYER ? row_sum<LPR>(s) * inv : 0.0f;
    float q = 0.0f;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      const int c = l * N + it * LPR * N;
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const float d = (c + i < dim) ? v[it][i] - mu : 0.0f;
        q += d * d;
      }
    }
    const float rs = rsqrtf(row_sum<LPR>(q) * inv + eps);
    if (ok) {
      if (l == 0) {
        if (mean) mean[row] = mu;
        rstd[row] = rs;
      }
#pragma unroll
      for (int it = 0; it < MAXIT; ++it) {
        const int c = l * N + it * LPR * N;
        if (c < dim) {
          float o[N];
#pragma unroll
          for (int i = 0; i < N; ++i) {
            float t = (v[it][i] - mu) * rs;
            if (kind == FK_NORM_RMS && sizeof(T) == 2) t = (float)(bf16_t)t;
            o[i] = t * gm[it][i] + bt[it][i];
          }
          VecIO<T>::store(y + row * dim + c, o);
        }
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void norm_bwd_dx_kernel(const T* dy, const T* x, const float* gamma, const float* mean,
                                                           const float* rstd, const T* dres, T* dx, int64_t rows, int dim, int kind) {
  constexpr int N = VecIO<T>::N;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t wstride = (int64_t)gridDim.x * 4;
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += wstride) {
    const T* xr = x + row * dim;
    const T* gr = dy + row * dim;
    const float mu = (kind == FK_NORM_LAYER) ? mean[row] : 0.0f, rs = rstd[row];
    float s1 = 0.0f, s2 = 0.0f;
    float xv[N], gv[N];
    for (int c = lane * N; c < dim; c += 64 * N) {
      VecIO<T>::load(xr + c, xv);
      VecIO<T>::load(gr + c, gv);
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const float dg = gv[i] * gamma[c + i];
        s1 += dg;
        s2 += dg * (xv[i] - mu) * rs;
      }
    }
    const float c1 = (kind == FK_NORM_LAYER) ? wave_sum(s1) / dim : 0.0f;
    const float c2 = wave_sum(s2) / dim;
    for (int c = lane * N; c < dim; c += 64 * N) {
      VecIO<T>::load(xr + c, xv);
      VecIO<T>::load(gr + c, gv);
      float o[N], rv[N];
      if (dres) VecIO<T>::load(dres + row * dim + c, rv);
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const float xh = (xv[i] - mu) * rs;
        o[i] = rs * (gv[i] * gamma[c + i] - c1 - xh * c2) + (dres ? rv[i] : 0.0f);
      }
      VecIO<T>::store(dx + row * dim + c, o);
    }
  }
}

// Fused backward: dx (+ residual-branch gradient) AND this block's partial dgamma/dbeta in one sweep; each row is read
// once into registers (LPR lanes per row, lane-in-row l owns columns l*N + it*LPR*N), column partials live in registers
// across the block's rows and are combined across row groups (shuffles) and the 4 waves (LDS).
// part layout: [gridDim.x][2][dim].
template <typename T, int LPR, int MAXIT>
__global__ __launch_bounds__(256) void norm_bwd_fused_kernel(const T* dy, const T* x, const float* gamma, const float* mean,
                                                              const float* rstd, const T* dres, T* dx, float* part,
                                                              int64_t rows, int dim, int kind) {
  constexpr int N = VecIO<T>::N, RPW = 64 / LPR;
  extern __shared__ float red[];   // [4 waves][2][dim]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l = lane % LPR, sub = lane / LPR;
  float ag[MAXIT][N], ab[MAXIT][N], gm[MAXIT][N];
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) {
    const int c = l * N + it * LPR * N;
#pragma unroll
    for (int i = 0; i < N; ++i) { ag[it][i] = 0.0f; ab[it][i] = 0.0f; }
    const bool in = c < dim;                              // dim % N == 0: whole vectors in or out, 16-byte gamma loads
#pragma unroll
    for (int i = 0; i < N; i += 4) {
      const f32x4 g4 = in ? *reinterpret_cast<const f32x4*>(gamma + c + i) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int e = 0; e < 4; ++e) gm[it][i + e] = g4[e];
    }
  }
  const float inv = 1.0f / dim;
  const int64_t stride = (int64_t)gridDim.x * 4 * RPW;
  for (int64_t base = ((int64_t)blockIdx.x * 4 + wave) * RPW; base < rows; base += stride) {
    const int64_t row = base + sub;
    const bool ok = row < rows;
    const float mu = (ok && kind == FK_NORM_LAYER) ? mean[row] : 0.0f, rs = ok ? rstd[row] : 0.0f;
    float xv[MAXIT][N], gv[MAXIT][N], rv[MAXIT][N];
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      const int c = l * N + it * LPR * N;
      if (ok && c < dim) {
        VecIO<T>::load(x + row * dim + c, xv[it]);
        VecIO<T>::load(dy + row * dim + c, gv[it]);
        if (dres) VecIO<T>::load(dres + row * dim + c, rv[it]);      // issued with the other loads: its latency hides under the sums
      } else {
#pragma unroll
        for (int i = 0; i < N; ++i) { xv[it][i] = mu; gv[it][i] = 0.0f; }
      }
#pragma unroll
      for (int i = 0; i < N; ++i) {
        xv[it][i] = (xv[it][i] - mu) * rs;             // xhat
        const float dg = gv[it][i] * gm[it][i];
        s1 += dg;
        s2 += dg * xv[it][i];
      }
    }
    const float c1 = (kind == FK_NORM_LAYER) ? row_sum<LPR>(s1) * inv : 0.0f;
    const float c2 = row_sum<LPR>(s2) * inv;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      const int c = l * N + it * LPR * N;
      if (ok && c < dim) {
        float o[N];
#pragma unroll
        for (int i = 0; i < N; ++i) {
          o[i] = rs * (gv[it][i] * gm[it][i] - c1 - xv[it][i] * c2) + (dres ? rv[it][i] : 0.0f);
          ag[it][i] += gv[it][i] * xv[it][i];
          ab[it][i] += gv[it][i];
        }
        VecIO<T>::store(dx + row * dim + c, o);
      }
    }
  }
  if (part == nullptr) return;
#pragma unroll
  for (int it = 0; it < MAXIT; ++it)
#pragma unroll
    for (int i = 0; i < N; ++i) {
#pragma unroll
      for (int o = LPR; o < 64; o <<= 1) {           // combine the wave's row groups (same columns)
        ag[it][i] += __shfl_xor(ag[it][i], o, 64);
        ab[it][i] += __shfl_xor(ab[it][i], o, 64);
      }
    }
  if (sub == 0) {
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      const int c = l * N + it * LPR * N;
      if (c < dim) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
          red[(wave * 2 + 0) * dim + c + i] = ag[it][i];
          red[(wave * 2 + 1) * dim + c + i] = ab[it][i];
        }
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * dim; i += blockDim.x)
    part[(int64_t)blockIdx.x * 2 * dim + i] = red[i] + red[2 * dim + i] + red[4 * dim + i] + red[6 * dim + i];
}
// out[which][c] = sum_k part[k][which][c]; block = 16 columns x 64 slices: 48 blocks at dim = 384 and 16 loads per thread for 1024 partial
// rows (the 64-column x 16-slice form before it ran 12 blocks with 64 dependent loads per thread: 16 us for 3 MB).  Fixed summation
// tree: slice sums, then 4 x 16, then 4 -> deterministic.
__global__ __launch_bounds__(1024) void norm_bwd_fused_final(const float* part, float* dgamma, float* dbeta, int dim, int nblk, int accumulate) {
  __shared__ float sh[64][17];
  __shared__ float sh2[4][16];
  const int cl = threadIdx.x & 15, sl = threadIdx.x >> 4, c = blockIdx.x * 16 + cl, which = blockIdx.y;
  float s = 0.0f;
  if (c < dim)
    for (int k = sl; k < nblk; k += 64) s += part[((int64_t)k * 2 + which) * dim + c];
  sh[sl][cl] = s;
  __syncthreads();
  if (threadIdx.x < 64) {
    float t = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += sh[sl * 16 + i][cl];
    sh2[sl][cl] = t;
  }
  __syncthreads();
  if (threadIdx.x < 16 && c < dim) {
    float* out = which == 0 ? dgamma : dbeta;
    if (out) {
      const float t = (sh2[0][cl] + sh2[1][cl]) + (sh2[2][cl] + sh2[3][cl]);
      out[c] = accumulate ? out[c] + t : t;
    }
  }
}

// partial[chunk][0][c] = sum_r dy*xhat, partial[chunk][1][c] = sum_r dy  over the chunk's rows
template <typename T>
__global__ void norm_bwd_param_partial(const T* dy, const T* x, const float* mean, const float* rstd, float* part,
                                       int64_t rows, int dim, int rows_per_chunk, int kind) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= dim) return;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
  float sg = 0.0f, sb = 0.0f;
  for (int64_t r = r0; r < r1; ++r) {
    const float g = to_f32<T>(dy[r * dim + c]);
    const float mu = (kind == FK_NORM_LAYER) ? mean[r] : 0.0f;
    sg += g * (to_f32<T>(x[r * dim + c]) - mu) * rstd[r];
    sb += g;
  }
  part[((int64_t)blockIdx.y * 2) * dim + c] = sg;
  part[((int64_t)blockIdx.y * 2 + 1) * dim + c] = sb;
}
__global__ void norm_bwd_param_final(const float* part, float* dgamma, float* dbeta, int dim, int nchunk, int accumulate) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= dim) return;
  float sg = 0.0f, sb = 0.0f;
  for (int k = 0; k < nchunk; ++k) {
    sg += part[((int64_t)k * 2) * dim + c];
    sb += part[((int64_t)k * 2 + 1) * dim + c];
  }
  if (dgamma) dgamma[c] = accumulate ? dgamma[c] + sg : sg;
  if (dbeta) dbeta[c] = accumulate ? dbeta[c] + sb : sb;
}

int nchunks(int64_t rows) {
  int64_t n = fk_cdiv(rows, 256);
  if (n > 512) n = 512;
  if (n < 1) n = 1;
  return (int)n;
}

}  // namespace

extern "C" {

int fk_norm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int64_t rows,
                int64_t dim, float eps, int kind, int dtype, void* stream) {
  FK_CHECK_ARG(dtype == FK_F32 || dtype == FK_BF16, "fk_norm_fwd: bad dtype %d", dtype);
  FK_CHECK_ARG(kind == FK_NORM_LAYER || kind == FK_NORM_RMS, "fk_norm_fwd: bad kind %d", kind);
  const int vec = dtype == FK_BF16 ? 8 : 4;
  FK_CHECK_ARG(rows > 0 && dim > 0 && dim % vec == 0 && dim < (1 << 24), "fk_norm_fwd: dim %lld must be a multiple of %d", (long long)dim, vec);
  FK_CHECK_ARG(x && y && gamma && rstd && (kind == FK_NORM_RMS || mean), "fk_norm_fwd: null pointer");
  FK_CHECK_ARG((((uintptr_t)x | (uintptr_t)y) & 15) == 0, "fk_norm_fwd: x/y must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const int per16 = 16 * vec, per64 = 64 * vec;
#define FK_NF(TT, LPR, MI)                                                                                                 \
  do {                                                                                                                     \
    int64_t nbf = fk_cdiv(rows, 4 * (64 / LPR));                                                                           \
    if (nbf > 2048) nbf = 2048;                                                                                            \
    hipLaunchKernelGGL((norm_fwd_fast_kernel<TT, LPR, MI>), dim3((unsigned)nbf), dim3(256), 0, s, (const TT*)x, gamma, beta, \
                       (TT*)y, mean, rstd, rows, (int)dim, eps, kind);                                                     \
  } while (0)
  bool done = (((uintptr_t)gamma | (uintptr_t)beta) & 15) == 0;       // the fast kernels read gamma / beta as 16-byte vectors
  if (!done) {
  } else if (dtype == FK_BF16) {
    if (dim <= per16) FK_NF(bf16_t, 16, 1); else if (dim <= 2 * per16) FK_NF(bf16_t, 16, 2); else if (dim <= 3 * per16) FK_NF(bf16_t, 16, 3);
    else if (dim <= 4 * per16) FK_NF(bf16_t, 16, 4); else if (dim <= 2 * per64) FK_NF(bf16_t, 64, 2); else if (dim <= 4 * per64) FK_NF(bf16_t, 64, 4);
    else done = false;
  } else {
    if (dim <= 2 * per16) FK_NF(float, 16, 2); else if (dim <= 4 * per16) FK_NF(float, 16, 4); else if (dim <= 6 * per16) FK_NF(float, 16, 6);
    else if (dim <= 2 * per64) FK_NF(float, 64, 2); else if (dim <= 4 * per64) FK_NF(float, 64, 4);
    else done = false;
  }
#undef FK_NF
  if (!done) {
    int64_t nb = fk_cdiv(rows, 4);
    if (nb > 8192) nb = 8192;
    if (dtype == FK_BF16)
      hipLaunchKernelGGL(norm_fwd_kernel<bf16_t>, dim3((unsigned)nb), dim3(256), 0, s, (const bf16_t*)x, gamma, beta, (bf16_t*)y, mean, rstd, rows, (int)dim, eps, kind);
    else
      hipLaunchKernelGGL(norm_fwd_kernel<float>, dim3((unsigned)nb), dim3(256), 0, s, (const float*)x, gamma, beta, (float*)y, mean, rstd, rows, (int)dim, eps, kind);
  }
  FK_CHECK_LAUNCH("fk_norm_fwd");
  return FK_OK;
}

size_t fk_norm_bwd_workspace_bytes(int64_t rows, int64_t dim) {
  int64_t nbf = fk_cdiv(rows, 16);
  if (nbf > 1024) nbf = 1024;
  if (nbf < 1) nbf = 1;
  const int64_t n = nbf > nchunks(rows) ? nbf : nchunks(rows);
  return (size_t)n * 2 * dim * sizeof(float);
}

int fk_norm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, const void* dres,
                void* dx, float* dgamma, float* dbeta, int64_t rows, int64_t dim, int kind, int accumulate, int dtype,
                void* workspace, size_t workspace_bytes, void* stream) {
  FK_CHECK_ARG(dtype == FK_F32 || dtype == FK_BF16, "fk_norm_bwd: bad dtype %d", dtype);
  FK_CHECK_ARG(kind == FK_NORM_LAYER || kind == FK_NORM_RMS, "fk_norm_bwd: bad kind %d", kind);
  const int vec = dtype == FK_BF16 ? 8 : 4;
  FK_CHECK_ARG(rows > 0 && dim > 0 && dim % vec == 0, "fk_norm_bwd: dim %lld must be a multiple of %d", (long long)dim, vec);
  FK_CHECK_ARG(dy && x && gamma && rstd && dx, "fk_norm_bwd: null pointer");
  hipStream_t s = (hipStream_t)stream;
  const int per16 = 16 * vec, per64 = 64 * vec;
  const bool fits = (dtype == FK_BF16 ? dim <= 4 * per64 : dim <= 4 * per64) && (size_t)8 * dim * sizeof(float) <= 65536 &&
                    ((uintptr_t)gamma & 15) == 0;     // the fused kernel reads gamma as 16-byte vectors
  if (fits) {
    // fused sweep: dx + per-block dgamma/dbeta partials
    const bool want = dgamma || dbeta;
    int64_t nbf = fk_cdiv(rows, 16);
    if (nbf > 1024) nbf = 1024;
    if (nbf < 1) nbf = 1;
    if (want) FK_CHECK_ARG(workspace && workspace_bytes >= (size_t)nbf * 2 * dim * sizeof(float), "fk_norm_bwd: workspace too small");
    float* part = want ? (float*)workspace : nullptr;
    const size_t sh = (size_t)8 * dim * sizeof(float);
#define FK_NB(TT, LPR, MI)                                                                                                  \
  hipLaunchKernelGGL((norm_bwd_fused_kernel<TT, LPR, MI>), dim3((unsigned)nbf), dim3(256), sh, s, (const TT*)dy, (const TT*)x, \
                     gamma, mean, rstd, (const TT*)dres, (TT*)dx, part, rows, (int)dim, kind)
    if (dtype == FK_BF16) {
      if (dim <= per16) FK_NB(bf16_t, 16, 1); else if (dim <= 2 * per16) FK_NB(bf16_t, 16, 2); else if (dim <= 3 * per16) FK_NB(bf16_t, 16, 3);
      else if (dim <= 4 * per16) FK_NB(bf16_t, 16, 4); else if (dim <= 2 * per64) FK_NB(bf16_t, 64, 2); else FK_NB(bf16_t, 64, 4);
    } else {
      if (dim <= 2 * per16) FK_NB(float, 16, 2); else if (dim <= 4 * per16) FK_NB(float, 16, 4); else if (dim <= 6 * per16) FK_NB(float, 16, 6);
      else if (dim <= 2 * per64) FK_NB(float, 64, 2); else FK_NB(float, 64, 4);
    }
#undef FK_NB
    FK_CHECK_LAUNCH("fk_norm_bwd(fused)");
    if (want) {
      hipLaunchKernelGGL(norm_bwd_fused_final, dim3((unsigned)fk_cdiv(dim, 16), 2), dim3(1024), 0, s, (const float*)part, dgamma, dbeta, (int)dim, (int)nbf, accumulate);
      FK_CHECK_LAUNCH("fk_norm_bwd(fused final)");
    }
    return FK_OK;
  }
  int64_t nb = fk_cdiv(rows, 4);
  if (nb > 8192) nb = 8192;
  if (dtype == FK_BF16)
    hipLaunchKernelGGL(norm_bwd_dx_kernel<bf16_t>, dim3((unsigned)nb), dim3(256), 0, s, (const bf16_t*)dy, (const bf16_t*)x, gamma, mean, rstd, (const bf16_t*)dres, (bf16_t*)dx, rows, (int)dim, kind);
  else
    hipLaunchKernelGGL(norm_bwd_dx_kernel<float>, dim3((unsigned)nb), dim3(256), 0, s, (const float*)dy, (const float*)x, gamma, mean, rstd, (const float*)dres, (float*)dx, rows, (int)dim, kind);
  FK_CHECK_LAUNCH("fk_norm_bwd(dx)");
  if (dgamma || dbeta) {
    const int nc = nchunks(rows);
    FK_CHECK_ARG(workspace && workspace_bytes >= (size_t)nc * 2 * dim * sizeof(float), "fk_norm_bwd: workspace too small");
    const int rpc = (int)fk_cdiv(rows, nc);
    dim3 grid((unsigned)fk_cdiv(dim, 128), (unsigned)nc);
    if (dtype == FK_BF16)
      hipLaunchKernelGGL(norm_bwd_param_partial<bf16_t>, grid, dim3(128), 0, s, (const bf16_t*)dy, (const bf16_t*)x, mean, rstd, (float*)workspace, rows, (int)dim, rpc, kind);
    else
      hipLaunchKernelGGL(norm_bwd_param_partial<float>, grid, dim3(128), 0, s, (const float*)dy, (const float*)x, mean, rstd, (float*)workspace, rows, (int)dim, rpc, kind);
    FK_CHECK_LAUNCH("fk_norm_bwd(partial)");
    hipLaunchKernelGGL(norm_bwd_param_final, dim3((unsigned)fk_cdiv(dim, 128)), dim3(128), 0, s, (const float*)workspace, dgamma, dbeta, (int)dim, nc, accumulate);
    FK_CHECK_LAUNCH("fk_norm_bwd(final)");
  }
  return FK_OK;
}

}  // extern "C"
