// head_ce.hip — vocabulary head + cross entropy as ONE product each way.
//
// Replaces  logits = lm_head(x); F.cross_entropy(logits.view(-1, V), targets, ignore_index=-100)  of models/gpt2_model.py:205-210 (tied
// lm_head, V = 50257) and the `to_words` head + loss of the notebook CE BrainFormer (notebooks_trainer/train_brainformer.ipynb cell 3)
// for callers that only need the loss (utils/train_utils.py:138-139 discards the logits).  The [rows, V] logits never exist:
//
//   fk_head_ce_fwd : every 128 x 128 tile of  H W^T (+ bias)  stays in the MFMA accumulators ("row on the lane": the C^T orientation of
//                    gemm.hip, so a row's maximum / sum of exponentials are per-lane scalars plus one cross-half shuffle); each wave
//                    writes one (max, sum exp) pair per row for its 64 columns and the lane that holds the target column writes the
//                    target logit.  A second small launch merges the V / 64 partial pairs of a row in a fixed order; fk_ce_chunk_finish
//                    turns them into row_lse and loss2 as before.
//   fk_head_ce_bwd : the same product again, "vocabulary entry on the lane" (the C orientation), so that the epilogue's
//                    dl = (exp(logit - lse) - onehot) * grad_out / #valid  goes out TRANSPOSED, dlT [V, rows]: the two gradient products
//                    are then contractions the existing kernels are built for — dH = fk_gemm_tn(dlT, W) (over the vocabulary, split
//                    slabs) and dW = fk_gemm_nt(dlT, H^T) (over the rows) — instead of a [rows x d] GEMM with K = 50257.
//   fk_transpose2d : H^T for that last product (rows x d elements; nothing else in the library needs a plain transpose).
#include "fk_common.h"
#include <type_traits>

namespace {

#include "gemm_tile.h"

struct HeadCeArgs {
  const void* H; const void* W; const void* bias;   // H [rows, K] (ldh); W [wrows >= V, K] (ldw); bias [>= V] or null, compute dtype
  const int64_t* tgt;
  int64_t ldh, ldw, ignore;
  int rows, V, wrows, K;
  // forward: partial statistics [rows, npart] per 64 vocabulary columns, target logit per row
  float* pmax; float* psum; float* tlogit; int npart;
  // backward
  const float* lse; const float* loss2; const float* gout;
  void* dlT; int64_t lddl; int rows_pad, vpad;
  float* dbpart;                                     // [2 * row tiles, vpad] column sums of dl per 64-row half tile (bias gradient), or null
};

// MODE 0: forward statistics (C^T: lane = row).  MODE 1: backward dl^T (C: lane = vocabulary entry).
template <typename T, int MODE>
__global__ __launch_bounds__(NTHREADS, 2) void head_ce_kernel(HeadCeArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BK = KT<T>::BK, VEC = KT<T>::VEC, STEPS = KT<T>::STEPS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
  const int ntn = (p.V + BN - 1) / BN;
  const unsigned L = xcd_remap(blockIdx.x, gridDim.x);
  const int mt = (int)(L / ntn), m0 = mt * BM, n0 = (int)(L % ntn) * BN;
  const T* A = (const T*)p.H;
  const T* B = (const T*)p.W;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  u32x4 ra[4], rb[4];
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int id = tid + NTHREADS * i, row = id >> 3, ch = id & 7, k = k0 + ch * VEC;
      u32x4 z = {0u, 0u, 0u, 0u};
      ra[i] = (m0 + row < p.rows && k < p.K) ? *reinterpret_cast<const u32x4*>(A + (int64_t)(m0 + row) * p.ldh + k) : z;
      rb[i] = (n0 + row < p.wrows && k < p.K) ? *reinterpret_cast<const u32x4*>(B + (int64_t)(n0 + row) * p.ldw + k) : z;
    }
  };
  auto lstore = [&](int buf) {
    char* as = smem + buf * 2 * TILE_BYTES;
    char* bs = as + TILE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int id = tid + NTHREADS * i, row = id >> 3, ch = id & 7;
      *reinterpret_cast<u32x4*>(as + nt_off(row, ch)) = ra[i];
      *reinterpret_cast<u32x4*>(bs + nt_off(row, ch)) = rb[i];
    }
  };
  const int nk = (p.K + BK - 1) / BK;
  gload(0);
  lstore(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) gload((kt + 1) * BK);
    const char* as = smem + (kt & 1) * 2 * TILE_BYTES;
    const char* bs = as + TILE_BYTES;
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      Frag<T> fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) nt_frag<T>(fa[i], as, wm * 64 + i * 32 + li, s, lh);
#pragma unroll
      for (int j = 0; j < 2; ++j) nt_frag<T>(fb[j], bs, wn * 64 + j * 32 + li, s, lh);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if constexpr (MODE == 0) mma32<T>(acc[i][j], fb[j], fa[i]);   // C^T: rows n (registers), columns m (lane)
          else mma32<T>(acc[i][j], fa[i], fb[j]);                        // C:   rows m (registers), columns n (lane)
        }
    }
    if (kt + 1 < nk) lstore((kt + 1) & 1);
    __syncthreads();
  }

  const T* bias = (const T*)p.bias;
  if constexpr (MODE == 0) {
    const int nbase = n0 + wn * 64, part = nbase >> 6;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = m0 + wm * 64 + i * 32 + li;
      const bool m_ok = m < p.rows;
      const int64_t tg = m_ok ? p.tgt[m] : (int64_t)-1;
      float v[2][16];
      float mx = -INFINITY, tl = 0.0f;
      bool hit = false;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = nbase + j * 32 + acc_row(r, lh);
          const bool ok = n < p.V;
          const float x = acc[i][j][r] + ((bias && ok) ? to_f32<T>(bias[n]) : 0.0f);
          v[j][r] = ok ? x : -INFINITY;
          mx = fmaxf(mx, v[j][r]);
          if (ok && (int64_t)n == tg) { tl = x; hit = true; }
        }
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float ref = mx == -INFINITY ? 0.0f : mx;        // a wave whose 64 columns all lie past V: sum 0, maximum -inf
      float sum = 0.0f;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += __expf(v[j][r] - ref);
      sum += __shfl_xor(sum, 32, 64);
      if (m_ok && lh == 0 && part < p.npart) {      // (the tile's second column half may lie wholly past the last part: V = 129 has 3 parts, 2 tiles)
        p.pmax[(int64_t)m * p.npart + part] = mx;
        p.psum[(int64_t)m * p.npart + part] = sum;
      }
      if (m_ok && hit) p.tlogit[m] = tl;
    }
  } else {
    const float gs = p.gout[0] / fmaxf(p.loss2[1], 1.0f);
    using TV = typename std::conditional<sizeof(T) == 2, bf16x4, f32x4>::type;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wn * 64 + j * 32 + li;
      const bool n_ok = n < p.V;
      const float bv = (bias && n_ok) ? to_f32<T>(bias[n]) : 0.0f;
      float colsum = 0.0f;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int mb = m0 + wm * 64 + i * 32;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int mq = mb + 8 * g + 4 * lh;                 // 4 consecutive rows: one store of dl^T[n, mq .. mq + 3]
          float o[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int m = mq + e;
            float x = 0.0f;
            if (n_ok && m < p.rows) {
              const int64_t tg = p.tgt[m];
              if (tg != p.ignore && tg >= 0 && tg < p.V) x = (__expf(acc[i][j][4 * g + e] + bv - p.lse[m]) - ((int64_t)n == tg ? 1.0f : 0.0f)) * gs;
            }
            o[e] = x;
            colsum += x;
          }
          if (n < p.vpad && mq < p.rows_pad) {
            T* dst = (T*)p.dlT + (int64_t)n * p.lddl + mq;
            TV q;
#pragma unroll
            for (int e = 0; e < 4; ++e) q[e] = from_f32<T>(o[e]);
            *reinterpret_cast<TV*>(dst) = q;
          }
        }
      }
      if (p.dbpart) {
        colsum += __shfl_xor(colsum, 32, 64);
        if (lh == 0 && n < p.vpad) p.dbpart[(int64_t)(2 * mt + wm) * p.vpad + n] = colsum;
      }
    }
  }
}

// row_m / row_s of a row = its npart (max, sum exp) pairs merged by one wave (coalesced reads, a fixed summation tree: deterministic)
__global__ __launch_bounds__(64) void head_ce_merge_kernel(const float* pmax, const float* psum, float* row_m, float* row_s, int rows, int npart) {
  const int row = blockIdx.x, lane = threadIdx.x;
  const float* pm = pmax + (int64_t)row * npart;
  const float* ps = psum + (int64_t)row * npart;
  float M = -INFINITY;
  for (int i = lane; i < npart; i += 64) M = fmaxf(M, pm[i]);
  M = wave_max(M);
  float S = 0.0f;
  for (int i = lane; i < npart; i += 64) {
    const float m = pm[i];
    if (m != -INFINITY) S += ps[i] * __expf(m - M);
  }
  S = wave_sum(S);
  if (lane == 0) {
    row_m[row] = M;
    row_s[row] = S;
  }
}

template <typename T>
__global__ void transpose2d_kernel(const T* src, int64_t lds, T* dst, int64_t ldd, int rows, int cols) {
  __shared__ T tile[32][33];
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: 32 x 8
  for (int y = ty; y < 32; y += 8)
    if (r0 + y < rows && c0 + tx < cols) tile[y][tx] = src[(int64_t)(r0 + y) * lds + c0 + tx];
  __syncthreads();
  for (int y = ty; y < 32; y += 8)
    if (c0 + y < cols && r0 + tx < rows) dst[(int64_t)(c0 + y) * ldd + r0 + tx] = tile[tx][y];
}

}  // namespace

extern "C" {

size_t fk_head_ce_workspace_bytes(int64_t rows, int64_t V) { return (size_t)rows * (size_t)fk_cdiv(V, 64) * 2 * sizeof(float); }

int fk_head_ce_fwd(const void* H, int64_t ldh, const void* W, int64_t ldw, int64_t wrows, const void* bias, const int64_t* targets,
                   float* row_m, float* row_s, float* row_t, int64_t rows, int64_t V, int64_t K, int dtype, void* workspace,
                   size_t workspace_bytes, void* stream) {
  FK_CHECK_ARG(dtype == FK_F32 || dtype == FK_BF16, "fk_head_ce_fwd: bad dtype %d", dtype);
  const int vec = dtype == FK_BF16 ? 8 : 4;
  FK_CHECK_ARG(H && W && targets && row_m && row_s && row_t && rows > 0 && V > 0 && K > 0 && wrows >= V, "fk_head_ce_fwd: bad arguments");
  FK_CHECK_ARG(rows < (1LL << 24) && V < (1LL << 30) && K < (1LL << 30) && fk_cdiv(rows, BM) * fk_cdiv(V, BN) < (1LL << 31), "fk_head_ce_fwd: problem too large");
  FK_CHECK_ARG(K % vec == 0 && ldh % vec == 0 && ldw % vec == 0 && (((uintptr_t)H | (uintptr_t)W) & 15) == 0,
               "fk_head_ce_fwd: K / ldh / ldw must be multiples of %d elements and H / W 16-byte aligned", vec);
  FK_CHECK_ARG(workspace && workspace_bytes >= fk_head_ce_workspace_bytes(rows, V), "fk_head_ce_fwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  HeadCeArgs p{};
  p.H = H; p.W = W; p.bias = bias; p.tgt = targets; p.ldh = ldh; p.ldw = ldw; p.ignore = -100;
  p.rows = (int)rows; p.V = (int)V; p.wrows = (int)wrows; p.K = (int)K;
  p.npart = (int)fk_cdiv(V, 64);
  p.pmax = (float*)workspace; p.psum = p.pmax + rows * (int64_t)p.npart; p.tlogit = row_t;
  if (hipMemsetAsync(row_t, 0, (size_t)rows * sizeof(float), s) != hipSuccess) return fk_set_error(FK_ELAUNCH, "fk_head_ce_fwd: memset failed");
  const dim3 grid((unsigned)(fk_cdiv(rows, BM) * fk_cdiv(V, BN))), block(NTHREADS);
  if (dtype == FK_BF16) hipLaunchKernelGGL((head_ce_kernel<bf16_t, 0>), grid, block, 4 * TILE_BYTES, s, p);
  else hipLaunchKernelGGL((head_ce_kernel<float, 0>), grid, block, 4 * TILE_BYTES, s, p);
  FK_CHECK_LAUNCH("fk_head_ce_fwd");
  hipLaunchKernelGGL(head_ce_merge_kernel, dim3((unsigned)rows), dim3(64), 0, s, (const float*)p.pmax, (const float*)p.psum, row_m, row_s, (int)rows, p.npart);
  FK_CHECK_LAUNCH("fk_head_ce_fwd(merge)");
  return FK_OK;
}

int fk_head_ce_bwd(const void* H, int64_t ldh, const void* W, int64_t ldw, int64_t wrows, const void* bias, const int64_t* targets,
                   const float* row_lse, const float* loss2, const float* grad_out, void* dlT, int64_t lddl, int64_t rows_pad,
                   int64_t vpad, float* dbpart, int64_t rows, int64_t V, int64_t K, int64_t ignore_index, int dtype, void* stream) {
  FK_CHECK_ARG(dtype == FK_F32 || dtype == FK_BF16, "fk_head_ce_bwd: bad dtype %d", dtype);
  const int vec = dtype == FK_BF16 ? 8 : 4;
  FK_CHECK_ARG(H && W && targets && row_lse && loss2 && grad_out && dlT && rows > 0 && V > 0 && K > 0 && wrows >= V, "fk_head_ce_bwd: bad arguments");
  FK_CHECK_ARG(rows < (1LL << 24) && V < (1LL << 30) && K < (1LL << 30) && fk_cdiv(rows, BM) * fk_cdiv(V, BN) < (1LL << 31), "fk_head_ce_bwd: problem too large");
  FK_CHECK_ARG(K % vec == 0 && ldh % vec == 0 && ldw % vec == 0 && (((uintptr_t)H | (uintptr_t)W | (uintptr_t)dlT) & 15) == 0,
               "fk_head_ce_bwd: K / ldh / ldw must be multiples of %d elements and H / W / dlT 16-byte aligned", vec);
  FK_CHECK_ARG(rows_pad >= rows && rows_pad % 4 == 0 && rows_pad <= fk_cdiv(rows, BM) * BM && lddl >= rows_pad && lddl % vec == 0 && vpad >= V &&
               vpad <= fk_cdiv(V, BN) * BN, "fk_head_ce_bwd: dlT is [vpad >= V, rows_pad >= rows] inside the tile grid, rows_pad %% 4 == 0, lddl %% %d == 0", vec);
  hipStream_t s = (hipStream_t)stream;
  HeadCeArgs p{};
  p.H = H; p.W = W; p.bias = bias; p.tgt = targets; p.ldh = ldh; p.ldw = ldw; p.ignore = ignore_index;
  p.rows = (int)rows; p.V = (int)V; p.wrows = (int)wrows; p.K = (int)K;
  p.lse = row_lse; p.loss2 = loss2; p.gout = grad_out; p.dlT = dlT; p.lddl = lddl; p.rows_pad = (int)rows_pad; p.vpad = (int)vpad; p.dbpart = dbpart;
  const dim3 grid((unsigned)(fk_cdiv(rows, BM) * fk_cdiv(V, BN))), block(NTHREADS);
  if (dtype == FK_BF16) hipLaunchKernelGGL((head_ce_kernel<bf16_t, 1>), grid, block, 4 * TILE_BYTES, s, p);
  else hipLaunchKernelGGL((head_ce_kernel<float, 1>), grid, block, 4 * TILE_BYTES, s, p);
  FK_CHECK_LAUNCH("fk_head_ce_bwd");
  return FK_OK;
}

int fk_transpose2d(const void* src, int64_t lds, void* dst, int64_t ldd, int64_t rows, int64_t cols, int dtype, void* stream) {
  FK_CHECK_ARG((dtype == FK_F32 || dtype == FK_BF16) && src && dst && rows > 0 && cols > 0 && lds >= cols && ldd >= rows && rows < (1LL << 31) &&
               cols < (1LL << 31) && fk_cdiv(rows, 32) < 65536, "fk_transpose2d: bad arguments");
  const dim3 grid((unsigned)fk_cdiv(cols, 32), (unsigned)fk_cdiv(rows, 32)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == FK_BF16) hipLaunchKernelGGL(transpose2d_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)src, lds, (bf16_t*)dst, ldd, (int)rows, (int)cols);
  else hipLaunchKernelGGL(transpose2d_kernel<float>, grid, block, 0, s, (const float*)src, lds, (float*)dst, ldd, (int)rows, (int)cols);
  FK_CHECK_LAUNCH("fk_transpose2d");
  return FK_OK;
}

}  // extern "C"
