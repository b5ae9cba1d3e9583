// elementwise.hip — HBM-bound pointwise / layout kernels of the hot path (SURVEY.md §2.3 K1, K4, K7 act, K8, K11).
//   fk_rope       apply_rope                       models/brainformer.py:70-91
//   fk_patchify   Rearrange('b (t p1) c -> b (t c) p1')  models/brainformer.py:282,338
//   fk_swiglu_*   silu(w1 x) * w3 x                models/brainformer.py:124
//   fk_gelu_*     nn.GELU() (exact erf)            models/gpt2_model.py:83,89
//   fk_cast_pack / fk_cast / fk_add / fk_copy2d    weight-shadow and residual plumbing
//   fk_embedding_*, fk_gpt_embed_fwd               nn.Embedding + prefix concat + wpe, models/gpt2_model.py:183-196
// All bf16 traffic is 16-byte vectorised; math in fp32.
#include "fk_common.h"

namespace {

constexpr int TPB = 256;
inline unsigned grid_for(int64_t work, int64_t cap = 1 << 20) {
  int64_t b = fk_cdiv(work, TPB);
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (unsigned)b;
}

template <typename T> struct V16;
template <> struct V16<bf16_t> {
  static constexpr int N = 8;
  FK_DEV static void ld(const bf16_t* p, float (&v)[8]) {
    bf16x8 a = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)a[i];
  }
  FK_DEV static void st(bf16_t* p, const float (&v)[8]) {
    bf16x8 a;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (bf16_t)v[i];
    *reinterpret_cast<bf16x8*>(p) = a;
  }
};
template <> struct V16<float> {
  static constexpr int N = 4;
  FK_DEV static void ld(const float* p, float (&v)[4]) {
    f32x4 a = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = a[i];
  }
  FK_DEV static void st(float* p, const float (&v)[4]) {
    f32x4 a;
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = v[i];
    *reinterpret_cast<f32x4*>(p) = a;
  }
};

// ---------------------------------------------------------------------------------------------- RoPE
template <typename T>
__global__ void rope_kernel(T* x, int64_t B, int64_t T_, int64_t ld, int rot_cols, int D, const float* table,
                            int64_t table_bs, int64_t pos_off, float sgn) {
  constexpr int N = V16<T>::N;
  const int cpr = rot_cols / N;   // vector chunks per row
  const int64_t total = B * T_ * cpr;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ch = (int)(i % cpr);
    const int64_t row = i / cpr, t = row % T_, b = row / T_;
    const int col = ch * N, d = col % D;   // d even, chunk stays inside one head (D % N == 0)
    T* px = x + row * ld + col;
    const float* tb = table + b * table_bs + ((pos_off + t) * (D / 2) + d / 2) * 2;
    float v[N], o[N];
    V16<T>::ld(px, v);
#pragma unroll
    for (int j = 0; j < N / 2; ++j) {
      const float c = tb[2 * j], s = sgn * tb[2 * j + 1];
      o[2 * j] = v[2 * j] * c - v[2 * j + 1] * s;
      o[2 * j + 1] = v[2 * j] * s + v[2 * j + 1] * c;
    }
    V16<T>::st(px, o);
  }
}

// ---------------------------------------------------------------------------------------------- patchify
template <typename T>
__global__ void patchify_kernel(const float* x, T* tok, int T_, int C, int P, int ldp) {
  extern __shared__ float slab[];   // [P][C+1]
  const int nT = T_ / P;
  const int b = blockIdx.x / nT, t = blockIdx.x % nT;
  const float* src = x + ((int64_t)b * T_ + (int64_t)t * P) * C;
  for (int i = threadIdx.x; i < P * C; i += blockDim.x) slab[(i / C) * (C + 1) + (i % C)] = src[i];
  __syncthreads();
  T* dst = tok + ((int64_t)blockIdx.x * C) * ldp;
  for (int i = threadIdx.x; i < C * ldp; i += blockDim.x) {
    const int c = i / ldp, p = i % ldp;
    dst[i] = from_f32<T>(p < P ? slab[p * (C + 1) + c] : 0.0f);
  }
}

// ---------------------------------------------------------------------------------------------- SwiGLU / GELU
FK_DEV float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

template <typename T>
__global__ void swiglu_fwd_kernel(const T* h13, T* g, int64_t rows, int H) {
  constexpr int N = V16<T>::N;
  const int cpr = H / N;
  const int64_t total = rows * cpr;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cpr;
    const int c = (int)(i % cpr) * N;
    float a[N], b[N], o[N];
    V16<T>::ld(h13 + r * 2 * H + c, a);
    V16<T>::ld(h13 + r * 2 * H + H + c, b);
#pragma unroll
    for (int j = 0; j < N; ++j) o[j] = a[j] * sigmoidf_(a[j]) * b[j];
    V16<T>::st(g + r * H + c, o);
  }
}
template <typename T>
__global__ void swiglu_bwd_kernel(const T* h13, const T* dg, T* dh13, int64_t rows, int H) {
  constexpr int N = V16<T>::N;
  const int cpr = H / N;
  const int64_t total = rows * cpr;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cpr;
    const int c = (int)(i % cpr) * N;
    float a[N], b[N], d[N], oa[N], ob[N];
    V16<T>::ld(h13 + r * 2 * H + c, a);
    V16<T>::ld(h13 + r * 2 * H + H + c, b);
    V16<T>::ld(dg + r * H + c, d);
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const float sg = sigmoidf_(a[j]);
      oa[j] = d[j] * b[j] * sg * (1.0f + a[j] * (1.0f - sg));
      ob[j] = d[j] * a[j] * sg;
    }
    V16<T>::st(dh13 + r * 2 * H + c, oa);
    V16<T>::st(dh13 + r * 2 * H + H + c, ob);
  }
}
template <typename T>
__global__ void gelu_fwd_kernel(const T* x, T* y, int64_t nvec) {
  constexpr int N = V16<T>::N;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
    float a[N], o[N];
    V16<T>::ld(x + i * N, a);
#pragma unroll
    for (int j = 0; j < N; ++j) o[j] = 0.5f * a[j] * (1.0f + erff(a[j] * 0.70710678118654752f));
    V16<T>::st(y + i * N, o);
  }
}
// y = [res +] keep(i) ? x * 1/(1-p) : 0     (nn.Dropout in training mode; the backward is the same kernel on dy without `res`)
template <typename T>
__global__ void dropout_kernel(const T* x, const T* res, T* y, int64_t nvec, const unsigned* seed, unsigned site, unsigned thresh, float ks) {
  constexpr int N = V16<T>::N;
  const DropKey key = drop_key(seed, site, thresh);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
    float a[N], o[N];
    V16<T>::ld(x + i * N, a);
    if (res) V16<T>::ld(res + i * N, o);
    const int64_t e0 = i * N;
    const unsigned row = drop_row(key, (unsigned)(e0 >> 32));       // N divides 2^32: the vector never straddles a change of the high word
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const float v = drop_keep(key, row, (unsigned)e0 + j) ? a[j] * ks : 0.0f;
      o[j] = res ? o[j] + v : v;
    }
    V16<T>::st(y + i * N, o);
  }
}
template <typename T>
__global__ void gelu_bwd_kernel(const T* x, const T* dy, T* dx, int64_t nvec) {
  constexpr int N = V16<T>::N;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
    float a[N], d[N], o[N];
    V16<T>::ld(x + i * N, a);
    V16<T>::ld(dy + i * N, d);
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const float cdf = 0.5f * (1.0f + erff(a[j] * 0.70710678118654752f));
      const float pdf = 0.3989422804014327f * __expf(-0.5f * a[j] * a[j]);
      o[j] = d[j] * (cdf + a[j] * pdf);
    }
    V16<T>::st(dx + i * N, o);
  }
}

// ---------------------------------------------------------------------------------------------- casts / copies
// dst row of source row j:  j' = (j / rblk) * rstride + (j % rblk) + roff   (rblk = 0: identity) — used to build the
// interleaved [w1 x4 | w3 x4] SwiGLU weight shadows.
template <typename T>
__global__ void cast_pack_kernel(const float* src, int64_t lds, T* dst, int64_t ldd, int rows, int cols, int transpose,
                                 int rblk, int rstride, int roff) {
  const int64_t total = (int64_t)rows * cols;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int r, c;
    if (transpose) { r = (int)(i % rows); c = (int)(i / rows); }   // consecutive threads walk dst rows: coalesced stores
    else { r = (int)(i / cols); c = (int)(i % cols); }
    const int rd = rblk > 0 ? (r / rblk) * rstride + (r % rblk) + roff : r;
    const float v = src[(int64_t)r * lds + c];
    if (transpose) dst[(int64_t)c * ldd + rd] = from_f32<T>(v);
    else dst[(int64_t)rd * ldd + c] = from_f32<T>(v);
  }
}
// all shadow re-packs of a step in one launch: a block takes chunks, finds the owning job by a (block-uniform) binary search over
// chunk_begin and then does what cast_pack_kernel does.  Plain jobs: a chunk = 1024 consecutive elements.  Transposed jobs: a chunk =
// one 32 x 32 source tile, read row-wise and written column-wise through LDS, so both sides move 128-byte segments (the element-wise
// form fetched 16x the bytes it needed).
template <typename T>
__global__ __launch_bounds__(256) void cast_pack_multi_kernel(const fk_pack_job* jobs, int njobs, int64_t total_chunks) {
  __shared__ float tile[32][33];
  for (int64_t ch = blockIdx.x; ch < total_chunks; ch += gridDim.x) {
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (jobs[mid].chunk_begin <= ch) lo = mid; else hi = mid - 1;
    }
    const fk_pack_job jb = jobs[lo];
    T* dst = (T*)jb.dst;
    const int64_t local = ch - jb.chunk_begin;
    if (!jb.transpose) {
      const int64_t total = (int64_t)jb.rows * jb.cols, base = local * 1024;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int64_t i = base + q * 256 + threadIdx.x;
        if (i >= total) break;
        const int r = (int)(i / jb.cols), c = (int)(i % jb.cols);
        const int rd = jb.rblk > 0 ? (r / jb.rblk) * jb.rstride + (r % jb.rblk) + jb.roff : r;
        dst[(int64_t)rd * jb.ldd + c] = from_f32<T>(jb.src[(int64_t)r * jb.lds + c]);
      }
    } else {
      const int tcols = (jb.cols + 31) / 32;
      const int r0 = (int)(local / tcols) * 32, c0 = (int)(local % tcols) * 32;
      const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
      __syncthreads();                                     // previous chunk's tile fully consumed
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = r0 + ty + 8 * q, c = c0 + tx;
        tile[ty + 8 * q][tx] = (r < jb.rows && c < jb.cols) ? jb.src[(int64_t)r * jb.lds + c] : 0.0f;
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = c0 + ty + 8 * q, r = r0 + tx;
        if (r < jb.rows && c < jb.cols) {
          const int rd = jb.rblk > 0 ? (r / jb.rblk) * jb.rstride + (r % jb.rblk) + jb.roff : r;
          dst[(int64_t)c * jb.ldd + rd] = from_f32<T>(tile[tx][ty + 8 * q]);
        }
      }
    }
  }
}
template <typename TS, typename TD>
__global__ void cast_kernel(const TS* src, TD* dst, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dst[i] = from_f32<TD>(to_f32<TS>(src[i]));
}
template <typename T>
__global__ void add_kernel(const T* a, const T* b, T* y, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = from_f32<T>(to_f32<T>(a[i]) + to_f32<T>(b[i]));
}
template <typename T>
__global__ void copy2d_kernel(const T* src, int64_t lds, T* dst, int64_t ldd, int64_t rows, int64_t cols) {
  const int64_t total = rows * cols;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
    dst[(i / cols) * ldd + (i % cols)] = src[(i / cols) * lds + (i % cols)];
}

__global__ void add2d_kernel(const float* src, int64_t lds, float* dst, int64_t ldd, int64_t rows, int64_t cols) {
  const int64_t total = rows * cols;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
    dst[(i / cols) * ldd + (i % cols)] += src[(i / cols) * lds + (i % cols)];
}

// ---------------------------------------------------------------------------------------------- embeddings
// out[r, :] = (src_sel ? prefix row : table[idx]) + pos[t]   for the GPT prefix-concat embedding
template <typename T>
__global__ void gpt_embed_kernel(const int64_t* idx, const T* prefix, const float* wte, const float* wpe, T* out, int B,
                                 int t_ctx, int t_words, int dim, int vocab) {
  const int t_full = t_ctx + t_words;
  const int64_t total = (int64_t)B * t_full * dim;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % dim);
    const int64_t row = i / dim;
    const int t = (int)(row % t_full), b = (int)(row / t_full);
    float v;
    if (t < t_ctx) v = to_f32<T>(prefix[((int64_t)b * t_ctx + t) * dim + c]);
    else {
      int64_t id = idx[(int64_t)b * t_words + (t - t_ctx)];
      id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
      v = wte[id * dim + c];
    }
    out[i] = from_f32<T>(v + wpe[(int64_t)t * dim + c]);
  }
}
// dtable[idx[r]] += dout[row_map(r)]   (token rows only; fp32 atomics, rows = B*t_words is small)
template <typename T>
__global__ void embed_bwd_kernel(const int64_t* idx, const T* dout, int64_t ldo_rows_per_b, int t_ctx, int t_words,
                                 float* dtable, int B, int dim, int vocab) {
  const int64_t total = (int64_t)B * t_words * dim;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % dim);
    const int64_t r = i / dim;
    const int j = (int)(r % t_words), b = (int)(r / t_words);
    int64_t id = idx[r];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    const float g = to_f32<T>(dout[((int64_t)b * ldo_rows_per_b + t_ctx + j) * dim + c]);
    atomicAdd(dtable + id * dim + c, g);
  }
}

// ---------------------------------------------------------------------------------------------- row gather / scatter (MAE)
// dst[b, i, :] = src[b, idx[b, i], :] (gather)  or  dst[b, idx[b, i], :] = src[b, i, :] (scatter); W elements per row.
// src_bs = 0 broadcasts one table over the batch (embedding lookup); TS -> TD converts on the fly.
template <typename TS, typename TD>
__global__ void gather_rows_kernel(const TS* src, int64_t src_bs, const int64_t* idx, int64_t idx_mod, TD* dst, int B, int n,
                                   int W, int scatter, int64_t dst_bs) {
  const int64_t total = (int64_t)B * n * W;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % W);
    const int64_t r = i / W;
    const int b = (int)(r / n), k = (int)(r % n);
    int64_t j = idx[r];
    if (idx_mod > 0) j %= idx_mod;
    if (scatter) dst[(int64_t)b * dst_bs + j * W + c] = from_f32<TD>(to_f32<TS>(src[(int64_t)b * src_bs + (int64_t)k * W + c]));
    else dst[(int64_t)b * dst_bs + (int64_t)k * W + c] = from_f32<TD>(to_f32<TS>(src[(int64_t)b * src_bs + j * W + c]));
  }
}
// The same for rows of W % 8 == 0 elements with 16-byte-aligned rows: a lane moves 8 elements (one 16-byte load on the narrower side), a
// wave covers 64 / (W / 8) rows per pass (W = 384: one row on 48 lanes), the row index is read once per row and the row arithmetic is
// 32-bit.  The element-wise kernel above spends a 64-bit division and modulo per ELEMENT: SimpleMAE's seven token gathers of 38 400 x 384
// took 82 us each at B = 256 (0.7 TB/s); same values (a copy, or the same conversion).
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void gather_rows_vec_kernel(const TS* src, int64_t src_bs, const int64_t* idx, int64_t idx_mod, TD* dst, int B, int n,
                                                              int W, int scatter, int64_t dst_bs) {
  const int cpr = W >> 3;                                   // 8-element chunks per row
  const int lane = threadIdx.x & 63;
  const int rpw = cpr >= 64 ? 1 : 64 / cpr;                 // rows per wave and pass
  const int sub = cpr >= 64 ? 0 : lane / cpr;               // which of them this lane works on
  const int c0 = cpr >= 64 ? lane : lane - sub * cpr;
  if (sub >= rpw) return;
  const unsigned rows = (unsigned)B * (unsigned)n;
  const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  for (unsigned r = wave * rpw + sub; r < rows; r += nwaves * rpw) {
    const unsigned b = r / (unsigned)n, k = r - b * (unsigned)n;
    int64_t j = idx[r];
    if (idx_mod > 0) j %= idx_mod;
    const TS* sp = src + (int64_t)b * src_bs + (scatter ? (int64_t)k : j) * W;
    TD* dp = dst + (int64_t)b * dst_bs + (scatter ? j : (int64_t)k) * W;
    for (int c = c0; c < cpr; c += 64) {
      float v[8];
      if constexpr (sizeof(TS) == 2) {
        const bf16x8 x = *reinterpret_cast<const bf16x8*>(sp + 8 * c);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (float)x[e];
      } else {
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(sp + 8 * c), x1 = *reinterpret_cast<const f32x4*>(sp + 8 * c + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = x0[e]; v[4 + e] = x1[e]; }
      }
      if constexpr (sizeof(TD) == 2) {
        bf16x8 y;
#pragma unroll
        for (int e = 0; e < 8; ++e) y[e] = from_f32<TD>(v[e]);
        *reinterpret_cast<bf16x8*>(dp + 8 * c) = y;
      } else {
        f32x4 y0, y1;
#pragma unroll
        for (int e = 0; e < 4; ++e) { y0[e] = v[e]; y1[e] = v[4 + e]; }
        *reinterpret_cast<f32x4*>(dp + 8 * c) = y0;
        *reinterpret_cast<f32x4*>(dp + 8 * c + 4) = y1;
      }
    }
  }
}
// table[idx[b, i] % idx_mod, :] += src[b, i, :]   (fp32 atomics; gradient of a broadcast-table gather)
template <typename TS>
__global__ void scatter_add_rows_kernel(const TS* src, const int64_t* idx, int64_t idx_mod, float* table, int64_t rows, int W) {
  const int64_t total = rows * W;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % W);
    int64_t j = idx[i / W];
    if (idx_mod > 0) j %= idx_mod;
    atomicAdd(table + j * W + c, to_f32<TS>(src[i]));
  }
}
// prefix-mask tables from sorted per-token block ids: limits[b, i] = #keys j with kid[b, j] / C <= qid[b, i] / C,
// qfirst[b, j] = first query i with qid[b, i] / C >= kid[b, j] / C   (ids ascending within a sample)
__global__ void prefix_mask_kernel(const int64_t* qid, const int64_t* kid, int C, int* limits, int* qfirst, int B, int nq, int nk) {
  const int64_t total = (int64_t)B * (nq + nk);
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(t / (nq + nk)), r = (int)(t % (nq + nk));
    const int64_t* q = qid + (int64_t)b * nq;
    const int64_t* k = kid + (int64_t)b * nk;
    if (r < nq) {                       // upper bound over key blocks
      const int64_t blk = q[r] / C;
      int lo = 0, hi = nk;
      while (lo < hi) { const int mid = (lo + hi) >> 1; if (k[mid] / C <= blk) lo = mid + 1; else hi = mid; }
      limits[(int64_t)b * nq + r] = lo;
    } else {                            // lower bound over query blocks
      const int j = r - nq;
      const int64_t blk = k[j] / C;
      int lo = 0, hi = nq;
      while (lo < hi) { const int mid = (lo + hi) >> 1; if (q[mid] / C < blk) lo = mid + 1; else hi = mid; }
      qfirst[(int64_t)b * nk + j] = lo;
    }
  }
}

// ---------------------------------------------------------------------------------------------- split-key attention: combine
// parts [B, S, T, H, D]: S partial attention results of the same queries over S disjoint key ranges (F.scaled_dot_product_attention of
// models/brainformer.py:215 with 32 perceiver queries against 6144 context keys: one workgroup per (batch, head) left 7 of 8 waves idle).
// lse != null (forward): out = sum_s exp(lse[b, s, h, t] - L) parts[b, s], L = log sum_s exp(lse[b, s, h, t]) written to lse_out [B, H, T].
// lse == null (the query gradient of the backward, whose P was normalised with the global L already): out = sum_s parts[b, s].
template <typename T>
__global__ void attn_combine_kernel(const T* parts, const float* lse, T* out, float* lse_out, int B, int S, int Tq, int H, int D) {
  const int64_t total = (int64_t)B * Tq * H * D;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int d = (int)(i % D);
    const int h = (int)((i / D) % H);
    const int t = (int)((i / ((int64_t)D * H)) % Tq);
    const int b = (int)(i / ((int64_t)D * H * Tq));
    const int64_t pstride = (int64_t)Tq * H * D;
    const T* pp = parts + (int64_t)b * S * pstride + ((int64_t)t * H + h) * D + d;
    float acc = 0.0f;
    if (lse) {
      const float* lp = lse + ((int64_t)b * S * H + h) * Tq + t;      // [B, S, H, T]
      float M = -INFINITY;
      for (int s = 0; s < S; ++s) M = fmaxf(M, lp[(int64_t)s * H * Tq]);
      float sum = 0.0f;
      for (int s = 0; s < S; ++s) {
        const float l = lp[(int64_t)s * H * Tq];
        const float w = (l == -INFINITY || M == -INFINITY || M == INFINITY) ? 0.0f : __expf(l - M);
        sum += w;
        acc += w * to_f32<T>(pp[(int64_t)s * pstride]);
      }
      acc = sum > 0.0f ? acc / sum : 0.0f;
      if (d == 0 && lse_out) lse_out[((int64_t)b * H + h) * Tq + t] = sum > 0.0f ? M + logf(sum) : M;
    } else {
      for (int s = 0; s < S; ++s) acc += to_f32<T>(pp[(int64_t)s * pstride]);
    }
    out[i] = from_f32<T>(acc);
  }
}

}  // namespace

#define FK_DT_CHECK(name) FK_CHECK_ARG(dtype == FK_F32 || dtype == FK_BF16, name ": bad dtype %d", dtype)

extern "C" {

int fk_rope(void* x, int64_t B, int64_t T, int64_t ld, int64_t nheads, int64_t D, const float* table, int64_t table_bs,
            int64_t pos_off, int conj, int dtype, void* stream) {
  FK_DT_CHECK("fk_rope");
  const int vec = dtype == FK_BF16 ? 8 : 4;
  FK_CHECK_ARG(x && table && B > 0 && T > 0 && nheads > 0, "fk_rope: bad arguments");
  FK_CHECK_ARG(D % vec == 0 && D % 2 == 0 && ld % vec == 0 && nheads * D <= ld, "fk_rope: D=%lld ld=%lld must be multiples of %d", (long long)D, (long long)ld, vec);
  FK_CHECK_ARG(((uintptr_t)x & 15) == 0, "fk_rope: x must be 16-byte aligned");
  const int64_t work = B * T * (nheads * D / vec);
  hipStream_t s = (hipStream_t)stream;
  const float sgn = conj ? -1.0f : 1.0f;
  if (dtype == FK_BF16)
    hipLaunchKernelGGL(rope_kernel<bf16_t>, dim3(grid_for(work, 16384)), dim3(TPB), 0, s, (bf16_t*)x, B, T, ld, (int)(nheads * D), (int)D, table, table_bs, pos_off, sgn);
  else
    hipLaunchKernelGGL(rope_kernel<float>, dim3(grid_for(work, 16384)), dim3(TPB), 0, s, (float*)x, B, T, ld, (int)(nheads * D), (int)D, table, table_bs, pos_off, sgn);
  FK_CHECK_LAUNCH("fk_rope");
  return FK_OK;
}

int fk_patchify(const float* x, void* tok, int64_t B, int64_t T, int64_t C, int64_t P, int64_t ldp, int dtype, void* stream) {
  FK_DT_CHECK("fk_patchify");
  FK_CHECK_ARG(x && tok && B > 0 && T > 0 && C > 0 && P > 0 && T % P == 0 && ldp >= P, "fk_patchify: bad shape (T %% P must be 0, ldp >= P)");
  const size_t sh = (size_t)P * (C + 1) * sizeof(float);
  FK_CHECK_ARG(sh <= 65536, "fk_patchify: P*(C+1) slab of %zu bytes exceeds 64 KiB LDS", sh);
  FK_CHECK_ARG(B * (T / P) < (1LL << 31), "fk_patchify: too many patches");
  dim3 grid((unsigned)(B * (T / P)));
  hipStream_t s = (hipStream_t)stream;
  if (dtype == FK_BF16) hipLaunchKernelGGL(patchify_kernel<bf16_t>, grid, dim3(TPB), sh, s, x, (bf16_t*)tok, (int)T, (int)C, (int)P, (int)ldp);
  else hipLaunchKernelGGL(patchify_kernel<float>, grid, dim3(TPB), sh, s, x, (float*)tok, (int)T, (int)C, (int)P, (int)ldp);
  FK_CHECK_LAUNCH("fk_patchify");
  return FK_OK;
}

int fk_swiglu_fwd(const void* h13, void* g, int64_t rows, int64_t H, int dtype, void* stream) {
  FK_DT_CHECK("fk_swiglu_fwd");
  const int vec = dtype == FK_BF16 ? 8 : 4;
  FK_CHECK_ARG(h13 && g && rows > 0 && H > 0 && H % vec == 0, "fk_swiglu_fwd: H must be a multiple of %d", vec);
  hipStream_t s = (hipStream_t)stream;
  const int64_t work = rows * (H / vec);
  if (dtype == FK_BF16) hipLaunchKernelGGL(swiglu_fwd_kernel<bf16_t>, dim3(grid_for(work, 16384)), dim3(TPB), 0, s, (const bf16_t*)h13, (bf16_t*)g, rows, (int)H);
  else hipLaunchKernelGGL(swiglu_fwd_kernel<float>, dim3(grid_for(work, 16384)), dim3(TPB), 0, s, (const float*)h13, (float*)g, rows, (int)H);
  FK_CHECK_LAUNCH("fk_swiglu_fwd");
  return FK_OK;
}
int fk_swiglu_bwd(const void* h13, const void* dg, void* dh13, int64_t rows, int64_t H, int dtype, void* stream) {
  FK_DT_CHECK("fk_swiglu_bwd");
  const int vec = dtype == FK_BF16 ? 8 : 4;
  FK_CHECK_ARG(h13 && dg && dh13 && rows > 0 && H > 0 && H % vec == 0, "fk_swiglu_bwd: H must be a multiple of %d", vec);
  hipStream_t s = (hipStream_t)stream;
  const int64_t work = rows * (H / vec);
  if (dtype == FK_BF16) hipLaunchKernelGGL(swiglu_bwd_kernel<bf16_t>, dim3(grid_for(work, 16384)), dim3(TPB), 0, s, (const bf16_t*)h13, (const bf16_t*)dg, (bf16_t*)dh13, rows, (int)H);
  else hipLaunchKernelGGL(swiglu_bwd_kernel<float>, dim3(grid_for(work, 16384)), dim3(TPB), 0, s, (const float*)h13, (const float*)dg, (float*)dh13, rows, (int)H);
  FK_CHECK_LAUNCH("fk_swiglu_bwd");
  return FK_OK;
}
int fk_gelu_fwd(const void* x, void* y, int64_t n, int dtype, void* stream) {
  FK_DT_CHECK("fk_gelu_fwd");
  const int vec = dtype == FK_BF16 ? 8 : 4;
  FK_CHECK_ARG(x && y && n > 0 && n % vec == 0, "fk_gelu_fwd: n must be a multiple of %d", vec);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == FK_BF16) hipLaunchKernelGGL(gelu_fwd_kernel<bf16_t>, dim3(grid_for(n / vec, 16384)), dim3(TPB), 0, s, (const bf16_t*)x, (bf16_t*)y, n / vec);
  else hipLaunchKernelGGL(gelu_fwd_kernel<float>, dim3(grid_for(n / vec, 16384)), dim3(TPB), 0, s, (const float*)x, (float*)y, n / vec);
  FK_CHECK_LAUNCH("fk_gelu_fwd");
  return FK_OK;
}
int fk_dropout(const void* x, const void* res, void* y, int64_t n, float p, const uint32_t* seed, uint32_t site, int dtype, void* stream) {
  FK_DT_CHECK("fk_dropout");
  const int vec = dtype == FK_BF16 ? 8 : 4;
  FK_CHECK_ARG(x && y && seed && n > 0 && n % vec == 0, "fk_dropout: n must be a multiple of %d", vec);
  FK_CHECK_ARG(p > 0.0f && p < 1.0f, "fk_dropout: p = %g outside (0, 1) (p = 0 is the caller's no-op)", (double)p);
  const double t = (double)p * 4294967296.0;
  const unsigned thresh = t >= 4294967295.0 ? 4294967295u : (t < 1.0 ? 1u : (unsigned)t);
  const float ks = 1.0f / (1.0f - p);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == FK_BF16) hipLaunchKernelGGL(dropout_kernel<bf16_t>, dim3(grid_for(n / vec, 16384)), dim3(TPB), 0, s, (const bf16_t*)x, (const bf16_t*)res, (bf16_t*)y, n / vec, seed, site, thresh, ks);
  else hipLaunchKernelGGL(dropout_kernel<float>, dim3(grid_for(n / vec, 16384)), dim3(TPB), 0, s, (const float*)x, (const float*)res, (float*)y, n / vec, seed, site, thresh, ks);
  FK_CHECK_LAUNCH("fk_dropout");
  return FK_OK;
}
int fk_gelu_bwd(const void* x, const void* dy, void* dx, int64_t n, int dtype, void* stream) {
  FK_DT_CHECK("fk_gelu_bwd");
  const int vec = dtype == FK_BF16 ? 8 : 4;
  FK_CHECK_ARG(x && dy && dx && n > 0 && n % vec == 0, "fk_gelu_bwd: n must be a multiple of %d", vec);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == FK_BF16) hipLaunchKernelGGL(gelu_bwd_kernel<bf16_t>, dim3(grid_for(n / vec, 16384)), dim3(TPB), 0, s, (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)dx, n / vec);
  else hipLaunchKernelGGL(gelu_bwd_kernel<float>, dim3(grid_for(n / vec, 16384)), dim3(TPB), 0, s, (const float*)x, (const float*)dy, (float*)dx, n / vec);
  FK_CHECK_LAUNCH("fk_gelu_bwd");
  return FK_OK;
}

int fk_cast_pack_rows(const float* src, int64_t lds, void* dst, int64_t ldd, int64_t rows, int64_t cols, int transpose,
                      int64_t rblk, int64_t rstride, int64_t roff, int dtype, void* stream) {
  FK_DT_CHECK("fk_cast_pack");
  FK_CHECK_ARG(src && dst && rows > 0 && cols > 0 && rows < (1LL << 31) && cols < (1LL << 31), "fk_cast_pack: bad shape");
  FK_CHECK_ARG(rblk >= 0 && (rblk == 0 || (rstride >= rblk && roff >= 0)), "fk_cast_pack: bad row map");
  hipStream_t s = (hipStream_t)stream;
  if (dtype == FK_BF16) hipLaunchKernelGGL(cast_pack_kernel<bf16_t>, dim3(grid_for(rows * cols, 8192)), dim3(TPB), 0, s, src, lds, (bf16_t*)dst, ldd, (int)rows, (int)cols, transpose, (int)rblk, (int)rstride, (int)roff);
  else hipLaunchKernelGGL(cast_pack_kernel<float>, dim3(grid_for(rows * cols, 8192)), dim3(TPB), 0, s, src, lds, (float*)dst, ldd, (int)rows, (int)cols, transpose, (int)rblk, (int)rstride, (int)roff);
  FK_CHECK_LAUNCH("fk_cast_pack");
  return FK_OK;
}
int fk_cast_pack(const float* src, int64_t lds, void* dst, int64_t ldd, int64_t rows, int64_t cols, int transpose,
                 int dtype, void* stream) {
  return fk_cast_pack_rows(src, lds, dst, ldd, rows, cols, transpose, 0, 0, 0, dtype, stream);
}
int fk_cast_pack_multi(const fk_pack_job* jobs, int64_t njobs, int64_t total_chunks, int dtype, void* stream) {
  FK_DT_CHECK("fk_cast_pack_multi");
  FK_CHECK_ARG(jobs && njobs > 0 && njobs < (1LL << 31) && total_chunks > 0, "fk_cast_pack_multi: bad arguments");
  static_assert(TPB == 256, "a chunk is 4 x 256 elements");
  hipStream_t s = (hipStream_t)stream;
  const unsigned nb = (unsigned)(total_chunks < 8192 ? total_chunks : 8192);
  if (dtype == FK_BF16) hipLaunchKernelGGL(cast_pack_multi_kernel<bf16_t>, dim3(nb), dim3(TPB), 0, s, jobs, (int)njobs, total_chunks);
  else hipLaunchKernelGGL(cast_pack_multi_kernel<float>, dim3(nb), dim3(TPB), 0, s, jobs, (int)njobs, total_chunks);
  FK_CHECK_LAUNCH("fk_cast_pack_multi");
  return FK_OK;
}
int fk_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, void* stream) {
  FK_CHECK_ARG((src_dtype == FK_F32 || src_dtype == FK_BF16) && (dst_dtype == FK_F32 || dst_dtype == FK_BF16), "fk_cast: bad dtype");
  FK_CHECK_ARG(src && dst && n > 0, "fk_cast: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  dim3 g(grid_for(n, 16384)), b(TPB);
  if (src_dtype == FK_F32 && dst_dtype == FK_BF16) hipLaunchKernelGGL((cast_kernel<float, bf16_t>), g, b, 0, s, (const float*)src, (bf16_t*)dst, n);
  else if (src_dtype == FK_BF16 && dst_dtype == FK_F32) hipLaunchKernelGGL((cast_kernel<bf16_t, float>), g, b, 0, s, (const bf16_t*)src, (float*)dst, n);
  else if (src_dtype == FK_F32) hipLaunchKernelGGL((cast_kernel<float, float>), g, b, 0, s, (const float*)src, (float*)dst, n);
  else hipLaunchKernelGGL((cast_kernel<bf16_t, bf16_t>), g, b, 0, s, (const bf16_t*)src, (bf16_t*)dst, n);
  FK_CHECK_LAUNCH("fk_cast");
  return FK_OK;
}
int fk_add(const void* a, const void* b, void* y, int64_t n, int dtype, void* stream) {
  FK_DT_CHECK("fk_add");
  FK_CHECK_ARG(a && b && y && n > 0, "fk_add: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  if (dtype == FK_BF16) hipLaunchKernelGGL(add_kernel<bf16_t>, dim3(grid_for(n, 16384)), dim3(TPB), 0, s, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)y, n);
  else hipLaunchKernelGGL(add_kernel<float>, dim3(grid_for(n, 16384)), dim3(TPB), 0, s, (const float*)a, (const float*)b, (float*)y, n);
  FK_CHECK_LAUNCH("fk_add");
  return FK_OK;
}
int fk_copy2d(const void* src, int64_t lds, void* dst, int64_t ldd, int64_t rows, int64_t cols, int dtype, void* stream) {
  FK_DT_CHECK("fk_copy2d");
  FK_CHECK_ARG(src && dst && rows > 0 && cols > 0, "fk_copy2d: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  if (dtype == FK_BF16) hipLaunchKernelGGL(copy2d_kernel<bf16_t>, dim3(grid_for(rows * cols, 16384)), dim3(TPB), 0, s, (const bf16_t*)src, lds, (bf16_t*)dst, ldd, rows, cols);
  else hipLaunchKernelGGL(copy2d_kernel<float>, dim3(grid_for(rows * cols, 16384)), dim3(TPB), 0, s, (const float*)src, lds, (float*)dst, ldd, rows, cols);
  FK_CHECK_LAUNCH("fk_copy2d");
  return FK_OK;
}

int fk_attn_combine(const void* parts, const float* lse_parts, void* out, float* lse_out, int64_t B, int64_t S, int64_t T, int64_t H,
                    int64_t D, int dtype, void* stream) {
  FK_DT_CHECK("fk_attn_combine");
  FK_CHECK_ARG(parts && out && B > 0 && S > 0 && T > 0 && H > 0 && D > 0 && B * S * T * H * D < (1LL << 40) && S < 65536, "fk_attn_combine: bad arguments");
  FK_CHECK_ARG(lse_parts || !lse_out, "fk_attn_combine: lse_out without lse_parts");
  hipStream_t s = (hipStream_t)stream;
  const int64_t work = B * T * H * D;
  if (dtype == FK_BF16) hipLaunchKernelGGL(attn_combine_kernel<bf16_t>, dim3(grid_for(work, 16384)), dim3(TPB), 0, s, (const bf16_t*)parts, lse_parts, (bf16_t*)out, lse_out, (int)B, (int)S, (int)T, (int)H, (int)D);
  else hipLaunchKernelGGL(attn_combine_kernel<float>, dim3(grid_for(work, 16384)), dim3(TPB), 0, s, (const float*)parts, lse_parts, (float*)out, lse_out, (int)B, (int)S, (int)T, (int)H, (int)D);
  FK_CHECK_LAUNCH("fk_attn_combine");
  return FK_OK;
}

int fk_add2d(const float* src, int64_t lds, float* dst, int64_t ldd, int64_t rows, int64_t cols, void* stream) {
  FK_CHECK_ARG(src && dst && rows > 0 && cols > 0, "fk_add2d: bad arguments");
  hipLaunchKernelGGL(add2d_kernel, dim3(grid_for(rows * cols, 16384)), dim3(TPB), 0, (hipStream_t)stream, src, lds, dst, ldd, rows, cols);
  FK_CHECK_LAUNCH("fk_add2d");
  return FK_OK;
}

int fk_gpt_embed_fwd(const int64_t* idx, const void* prefix, const float* wte, const float* wpe, void* out, int64_t B,
                     int64_t t_ctx, int64_t t_words, int64_t dim, int64_t vocab, int dtype, void* stream) {
  FK_DT_CHECK("fk_gpt_embed_fwd");
  FK_CHECK_ARG(idx && wte && wpe && out && B > 0 && t_words > 0 && t_ctx >= 0 && dim > 0 && vocab > 0, "fk_gpt_embed_fwd: bad arguments");
  FK_CHECK_ARG(t_ctx == 0 || prefix, "fk_gpt_embed_fwd: prefix is NULL but t_ctx > 0");
  hipStream_t s = (hipStream_t)stream;
  const int64_t work = B * (t_ctx + t_words) * dim;
  if (dtype == FK_BF16) hipLaunchKernelGGL(gpt_embed_kernel<bf16_t>, dim3(grid_for(work, 16384)), dim3(TPB), 0, s, idx, (const bf16_t*)prefix, wte, wpe, (bf16_t*)out, (int)B, (int)t_ctx, (int)t_words, (int)dim, (int)vocab);
  else hipLaunchKernelGGL(gpt_embed_kernel<float>, dim3(grid_for(work, 16384)), dim3(TPB), 0, s, idx, (const float*)prefix, wte, wpe, (float*)out, (int)B, (int)t_ctx, (int)t_words, (int)dim, (int)vocab);
  FK_CHECK_LAUNCH("fk_gpt_embed_fwd");
  return FK_OK;
}
int fk_gpt_embed_bwd_wte(const int64_t* idx, const void* dout, float* dwte, int64_t B, int64_t t_ctx, int64_t t_words,
                         int64_t dim, int64_t vocab, int dtype, void* stream) {
  FK_DT_CHECK("fk_gpt_embed_bwd_wte");
  FK_CHECK_ARG(idx && dout && dwte && B > 0 && t_words > 0 && dim > 0, "fk_gpt_embed_bwd_wte: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const int64_t work = B * t_words * dim;
  if (dtype == FK_BF16) hipLaunchKernelGGL(embed_bwd_kernel<bf16_t>, dim3(grid_for(work, 16384)), dim3(TPB), 0, s, idx, (const bf16_t*)dout, t_ctx + t_words, (int)t_ctx, (int)t_words, dwte, (int)B, (int)dim, (int)vocab);
  else hipLaunchKernelGGL(embed_bwd_kernel<float>, dim3(grid_for(work, 16384)), dim3(TPB), 0, s, idx, (const float*)dout, t_ctx + t_words, (int)t_ctx, (int)t_words, dwte, (int)B, (int)dim, (int)vocab);
  FK_CHECK_LAUNCH("fk_gpt_embed_bwd_wte");
  return FK_OK;
}

int fk_gather_rows(const void* src, int64_t src_bs, int src_dtype, const int64_t* idx, int64_t idx_mod, void* dst, int64_t dst_bs,
                   int dst_dtype, int64_t B, int64_t n, int64_t W, int scatter, void* stream) {
  FK_CHECK_ARG((src_dtype == FK_F32 || src_dtype == FK_BF16) && (dst_dtype == FK_F32 || dst_dtype == FK_BF16), "fk_gather_rows: bad dtype");
  FK_CHECK_ARG(src && idx && dst && B > 0 && n > 0 && W > 0 && B * n < (1LL << 31) && W < (1LL << 31), "fk_gather_rows: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  dim3 g(grid_for(B * n * W, 16384)), b(TPB);
  // whole 8-element chunks, rows and batches on 16-byte boundaries on both sides: the vector kernel
  const bool vec = W % 8 == 0 && src_bs % 8 == 0 && dst_bs % 8 == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0;
  if (vec) {
    const int64_t cpr = W / 8, rpw = cpr >= 64 ? 1 : 64 / cpr, waves = fk_cdiv(B * n, rpw);
    g = dim3((unsigned)(fk_cdiv(waves, TPB / 64) < 4096 ? fk_cdiv(waves, TPB / 64) : 4096));
  }
#define FK_GR(TS, TD)                                                                                                                              \
  do {                                                                                                                                             \
    if (vec) hipLaunchKernelGGL((gather_rows_vec_kernel<TS, TD>), g, b, 0, s, (const TS*)src, src_bs, idx, idx_mod, (TD*)dst, (int)B, (int)n, (int)W, scatter, dst_bs); \
    else hipLaunchKernelGGL((gather_rows_kernel<TS, TD>), g, b, 0, s, (const TS*)src, src_bs, idx, idx_mod, (TD*)dst, (int)B, (int)n, (int)W, scatter, dst_bs); \
  } while (0)
  if (src_dtype == FK_F32 && dst_dtype == FK_F32) FK_GR(float, float);
  else if (src_dtype == FK_F32) FK_GR(float, bf16_t);
  else if (dst_dtype == FK_F32) FK_GR(bf16_t, float);
  else FK_GR(bf16_t, bf16_t);
#undef FK_GR
  FK_CHECK_LAUNCH("fk_gather_rows");
  return FK_OK;
}
int fk_scatter_add_rows(const void* src, int src_dtype, const int64_t* idx, int64_t idx_mod, float* table, int64_t rows, int64_t W,
                        void* stream) {
  FK_CHECK_ARG(src_dtype == FK_F32 || src_dtype == FK_BF16, "fk_scatter_add_rows: bad dtype");
  FK_CHECK_ARG(src && idx && table && rows > 0 && W > 0, "fk_scatter_add_rows: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  if (src_dtype == FK_BF16) hipLaunchKernelGGL(scatter_add_rows_kernel<bf16_t>, dim3(grid_for(rows * W, 16384)), dim3(TPB), 0, s, (const bf16_t*)src, idx, idx_mod, table, rows, (int)W);
  else hipLaunchKernelGGL(scatter_add_rows_kernel<float>, dim3(grid_for(rows * W, 16384)), dim3(TPB), 0, s, (const float*)src, idx, idx_mod, table, rows, (int)W);
  FK_CHECK_LAUNCH("fk_scatter_add_rows");
  return FK_OK;
}
int fk_prefix_mask(const int64_t* q_ids, const int64_t* k_ids, int64_t block, int32_t* limits, int32_t* qfirst, int64_t B,
                   int64_t nq, int64_t nk, void* stream) {
  FK_CHECK_ARG(q_ids && k_ids && limits && qfirst && block > 0 && B > 0 && nq > 0 && nk > 0, "fk_prefix_mask: bad arguments");
  hipLaunchKernelGGL(prefix_mask_kernel, dim3(grid_for(B * (nq + nk), 4096)), dim3(TPB), 0, (hipStream_t)stream, q_ids, k_ids, (int)block,
                     limits, qfirst, (int)B, (int)nq, (int)nk);
  FK_CHECK_LAUNCH("fk_prefix_mask");
  return FK_OK;
}

}  // extern "C"
