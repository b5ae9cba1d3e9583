// loss_optim.hip — losses and the fused optimizer step (SURVEY.md §2.3 K9 softmax/CE, K10, K13+K14; HBM-bound).
//   fk_l1_loss_*  F.l1_loss / F.mse_loss (mean)         models/brainformer.py:557, :473
//   fk_ce_*       F.cross_entropy(ignore_index, mean)   models/gpt2_model.py:210; train_brainformer.ipynb cell 3
//   fk_adamw_step clip_grad_value_ + AdamW.step         utils/train_utils.py:117-119,142-143
//   fk_api        version / last-error
#include <stdarg.h>
#include <stdio.h>

#include "fk_common.h"

static thread_local char g_err[512] = "";

int fk_set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

namespace {

constexpr int TPB = 256;

FK_DEV float block_sum(float v, float* sh) {   // sh: >= 4 floats; result valid in all threads
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.0f;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sh[i];
  return t;
}
FK_DEV float block_max(float v, float* sh) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = -INFINITY;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t = fmaxf(t, sh[i]);
  return t;
}

// ---- L1 / MSE ---------------------------------------------------------------------------------
// part[blk] = sum w * f(d), part[nblk + blk] = sum w  (w = row_w[i / row_len] or 1)
template <typename T>
__global__ void l1_partial_kernel(const T* pred, const T* tgt, float* part, int64_t n, int squared, const float* row_w, int64_t row_len) {
  __shared__ float sh[4];
  float s = 0.0f, c = 0.0f;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float d = to_f32<T>(pred[i]) - to_f32<T>(tgt[i]);
    const float w = row_w ? row_w[i / row_len] : 1.0f;
    s += w * (squared ? d * d : fabsf(d));
    c += w;
  }
  s = block_sum(s, sh);
  c = block_sum(c, sh);
  if (threadIdx.x == 0) { part[blockIdx.x] = s; part[gridDim.x + blockIdx.x] = c; }
}
// out[0] = sum / count, out[1] = count
__global__ void l1_final_kernel(const float* part, int nparts, float* out) {
  __shared__ float sh[4];
  float s = 0.0f, c = 0.0f;
  for (int i = threadIdx.x; i < nparts; i += blockDim.x) { s += part[i]; c += part[nparts + i]; }
  s = block_sum(s, sh);
  c = block_sum(c, sh);
  if (threadIdx.x == 0) { out[0] = s / c; out[1] = c; }
}
template <typename T>
__global__ void l1_bwd_kernel(const T* pred, const T* tgt, const float* gout, T* dpred, int64_t n, int squared,
                              const float* row_w, int64_t row_len, const float* loss2) {
  const float g = gout[0] / (row_w ? loss2[1] : (float)n);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float d = to_f32<T>(pred[i]) - to_f32<T>(tgt[i]);
    const float v = squared ? 2.0f * d : (d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f));
    dpred[i] = from_f32<T>(v * g * (row_w ? row_w[i / row_len] : 1.0f));
  }
}

// ---- cross entropy ----------------------------------------------------------------------------------
// one block per row: lse = log sum exp(logits[row, :]); nll[row] = valid ? lse - logit[target] : 0
template <typename T>
__global__ void ce_row_kernel(const T* logits, int64_t ld, const int64_t* targets, float* row_lse, float* nll,
                              float* valid, int V, int64_t ignore_index) {
  __shared__ float sh[4];
  const int64_t row = blockIdx.x;
  const T* lr = logits + row * ld;
  float mx = -INFINITY;
  for (int c = threadIdx.x; c < V; c += blockDim.x) mx = fmaxf(mx, to_f32<T>(lr[c]));
  mx = block_max(mx, sh);
  float s = 0.0f;
  for (int c = threadIdx.x; c < V; c += blockDim.x) s += __expf(to_f32<T>(lr[c]) - mx);
  s = block_sum(s, sh);
  if (threadIdx.x == 0) {
    const float lse = mx + logf(s);
    row_lse[row] = lse;
    const int64_t t = targets[row];
    const bool ok = (t != ignore_index) && t >= 0 && t < V;
    nll[row] = ok ? lse - to_f32<T>(lr[t]) : 0.0f;
    valid[row] = ok ? 1.0f : 0.0f;
  }
}
// loss[0] = sum nll / count, loss[1] = count
__global__ void ce_final_kernel(const float* nll, const float* valid, float* loss, int rows) {
  __shared__ float sh[4];
  float s = 0.0f, c = 0.0f;
  for (int i = threadIdx.x; i < rows; i += blockDim.x) { s += nll[i]; c += valid[i]; }
  s = block_sum(s, sh);
  c = block_sum(c, sh);
  if (threadIdx.x == 0) { loss[0] = s / c; loss[1] = c; }
}
template <typename T>
__global__ void ce_bwd_kernel(const T* logits, int64_t ld, const int64_t* targets, const float* row_lse,
                              const float* loss_cnt, const float* gout, T* dlogits, int64_t ldd, int64_t rows, int V,
                              int64_t ignore_index) {
  const int64_t total = rows * V;
  const float g = gout[0] / loss_cnt[1];
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / V;
    const int c = (int)(i % V);
    const int64_t t = targets[r];
    const bool ok = (t != ignore_index) && t >= 0 && t < V;
    float d = 0.0f;
    if (ok) d = (__expf(to_f32<T>(logits[r * ld + c]) - row_lse[r]) - (c == t ? 1.0f : 0.0f)) * g;
    dlogits[r * ldd + c] = from_f32<T>(d);
  }
}


// ---- cross entropy over the vocabulary in CHUNKS (the head GEMM runs chunk by chunk; [rows, V] is never materialised) ----------------
// running (max, sum exp, target logit) per row, merged with one fp32 chunk [rows, cw] whose first column is vocabulary index col0
__global__ void ce_chunk_fwd_kernel(const float* logits, int64_t ld, const int64_t* targets, int64_t col0, float* row_m, float* row_s,
                                    float* row_t, int cw, int first) {
  __shared__ float sh[4];
  const int64_t row = blockIdx.x;
  const float* lr = logits + row * ld;
  float mx = -INFINITY;
  for (int c = threadIdx.x; c < cw; c += blockDim.x) mx = fmaxf(mx, lr[c]);
  mx = block_max(mx, sh);
  float s = 0.0f;
  for (int c = threadIdx.x; c < cw; c += blockDim.x) s += __expf(lr[c] - mx);
  s = block_sum(s, sh);
  if (threadIdx.x == 0) {
    const float m0 = first ? -INFINITY : row_m[row], s0 = first ? 0.0f : row_s[row];
    const float mn = fmaxf(m0, mx);
    row_m[row] = mn;
    row_s[row] = (m0 == -INFINITY ? 0.0f : s0 * __expf(m0 - mn)) + s * __expf(mx - mn);
    const int64_t t = targets[row] - col0;
    if (first) row_t[row] = 0.0f;
    if (t >= 0 && t < cw) row_t[row] = lr[t];
  }
}
__global__ void ce_chunk_rows_kernel(const float* row_m, const float* row_s, const float* row_t, const int64_t* targets, float* row_lse,
                                     float* nll, float* valid, int64_t rows, int64_t V, int64_t ignore_index) {
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < rows; r += (int64_t)gridDim.x * blockDim.x) {
    const float lse = row_m[r] + logf(row_s[r]);
    row_lse[r] = lse;
    const int64_t t = targets[r];
    const bool ok = (t != ignore_index) && t >= 0 && t < V;
    nll[r] = ok ? lse - row_t[r] : 0.0f;
    valid[r] = ok ? 1.0f : 0.0f;
  }
}
// dlogits chunk = (softmax - onehot) * grad_out / #valid for valid rows, 0 otherwise; written in the compute dtype
template <typename T>
__global__ void ce_chunk_bwd_kernel(const float* logits, int64_t ld, const int64_t* targets, int64_t col0, const float* row_lse,
                                    const float* loss_cnt, const float* gout, T* dl, int64_t ldd, int64_t rows, int cw, int cw_valid,
                                    int64_t V, int64_t ignore_index) {
  const int64_t total = rows * cw;
  const float g = gout[0] / loss_cnt[1];
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cw;
    const int c = (int)(i % cw);
    const int64_t t = targets[r];
    const bool ok = (t != ignore_index) && t >= 0 && t < V;
    float d = 0.0f;
    if (ok && c < cw_valid) d = (__expf(logits[r * ld + c] - row_lse[r]) - (col0 + c == t ? 1.0f : 0.0f)) * g;
    dl[r * ldd + c] = from_f32<T>(d);       // padded columns (c >= cw_valid) are written as zeros: they are GEMM operands
  }
}

// ---- AdamW --------------------------------------------------------------------------------------------
__global__ void adamw_kernel(float* p, float* g, float* m, float* v, int64_t n, float lr, float omb1, float b2, float omb2,
                             float eps, float wd, float bc1, float sqrt_bc2, float clip, float gscale, int zero_grad) {
  const int64_t nv = n >> 2;
  const float step_size = lr / bc1, decay = 1.0f - lr * wd;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
    f32x4 pv = reinterpret_cast<f32x4*>(p)[i], gv = reinterpret_cast<f32x4*>(g)[i];
    f32x4 mv = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float gg = gv[j] * gscale;
      if (clip > 0.0f) gg = fminf(fmaxf(gg, -clip), clip);
      pv[j] *= decay;
      mv[j] += omb1 * (gg - mv[j]);                    // exp_avg.lerp_(grad, 1 - beta1)
      vv[j] = b2 * vv[j] + omb2 * gg * gg;             // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
      pv[j] -= step_size * (mv[j] / (sqrtf(vv[j]) / sqrt_bc2 + eps));
    }
    reinterpret_cast<f32x4*>(p)[i] = pv;
    reinterpret_cast<f32x4*>(m)[i] = mv;
    reinterpret_cast<f32x4*>(v)[i] = vv;
    if (zero_grad) reinterpret_cast<f32x4*>(g)[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  }
  // tail
  for (int64_t i = (nv << 2) + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float gg = g[i] * gscale;
    if (clip > 0.0f) gg = fminf(fmaxf(gg, -clip), clip);
    float pv = p[i] * decay;
    const float mv = m[i] + omb1 * (gg - m[i]), vv = b2 * v[i] + omb2 * gg * gg;
    pv -= step_size * (mv / (sqrtf(vv) / sqrt_bc2 + eps));
    p[i] = pv; m[i] = mv; v[i] = vv;
    if (zero_grad) g[i] = 0.0f;
  }
}

inline unsigned grid_for(int64_t work, int64_t cap) {
  int64_t b = fk_cdiv(work, TPB);
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace

extern "C" {

int fk_version(void) { return FK_VERSION; }
const char* fk_last_error(void) { return g_err; }

size_t fk_loss_workspace_bytes(int64_t n) { return (size_t)grid_for(n, 1024) * 2 * sizeof(float); }

int fk_l1_loss_fwd(const void* pred, const void* target, float* loss2, int64_t n, int squared, const float* row_weight,
                   int64_t row_len, int dtype, void* workspace, size_t workspace_bytes, void* stream) {
  FK_CHECK_ARG(dtype == FK_F32 || dtype == FK_BF16, "fk_l1_loss_fwd: bad dtype %d", dtype);
  FK_CHECK_ARG(pred && target && loss2 && n > 0 && (!row_weight || (row_len > 0 && n % row_len == 0)), "fk_l1_loss_fwd: bad arguments");
  const unsigned nb = grid_for(n, 1024);
  FK_CHECK_ARG(workspace && workspace_bytes >= (size_t)nb * 2 * sizeof(float), "fk_l1_loss_fwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  if (dtype == FK_BF16) hipLaunchKernelGGL(l1_partial_kernel<bf16_t>, dim3(nb), dim3(TPB), 0, s, (const bf16_t*)pred, (const bf16_t*)target, (float*)workspace, n, squared, row_weight, row_len);
  else hipLaunchKernelGGL(l1_partial_kernel<float>, dim3(nb), dim3(TPB), 0, s, (const float*)pred, (const float*)target, (float*)workspace, n, squared, row_weight, row_len);
  FK_CHECK_LAUNCH("fk_l1_loss_fwd");
  hipLaunchKernelGGL(l1_final_kernel, dim3(1), dim3(TPB), 0, s, (const float*)workspace, (int)nb, loss2);
  FK_CHECK_LAUNCH("fk_l1_loss_fwd(final)");
  return FK_OK;
}
int fk_l1_loss_bwd(const void* pred, const void* target, const float* grad_out, void* dpred, int64_t n, int squared,
                   const float* row_weight, int64_t row_len, const float* loss2, int dtype, void* stream) {
  FK_CHECK_ARG(dtype == FK_F32 || dtype == FK_BF16, "fk_l1_loss_bwd: bad dtype %d", dtype);
  FK_CHECK_ARG(pred && target && grad_out && dpred && n > 0 && (!row_weight || (row_len > 0 && loss2)), "fk_l1_loss_bwd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  if (dtype == FK_BF16) hipLaunchKernelGGL(l1_bwd_kernel<bf16_t>, dim3(grid_for(n, 4096)), dim3(TPB), 0, s, (const bf16_t*)pred, (const bf16_t*)target, grad_out, (bf16_t*)dpred, n, squared, row_weight, row_len, loss2);
  else hipLaunchKernelGGL(l1_bwd_kernel<float>, dim3(grid_for(n, 4096)), dim3(TPB), 0, s, (const float*)pred, (const float*)target, grad_out, (float*)dpred, n, squared, row_weight, row_len, loss2);
  FK_CHECK_LAUNCH("fk_l1_loss_bwd");
  return FK_OK;
}

size_t fk_ce_workspace_bytes(int64_t rows) { return (size_t)rows * 2 * sizeof(float); }

int fk_ce_loss_fwd(const void* logits, int64_t ld, const int64_t* targets, float* loss2, float* row_lse, int64_t rows,
                   int64_t V, int64_t ignore_index, int dtype, void* workspace, size_t workspace_bytes, void* stream) {
  FK_CHECK_ARG(dtype == FK_F32 || dtype == FK_BF16, "fk_ce_loss_fwd: bad dtype %d", dtype);
  FK_CHECK_ARG(logits && targets && loss2 && row_lse && rows > 0 && V > 0 && rows < (1LL << 31) && V < (1LL << 31), "fk_ce_loss_fwd: bad arguments");
  FK_CHECK_ARG(workspace && workspace_bytes >= (size_t)rows * 2 * sizeof(float), "fk_ce_loss_fwd: workspace too small");
  float* nll = (float*)workspace;
  float* valid = nll + rows;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == FK_BF16) hipLaunchKernelGGL(ce_row_kernel<bf16_t>, dim3((unsigned)rows), dim3(TPB), 0, s, (const bf16_t*)logits, ld, targets, row_lse, nll, valid, (int)V, ignore_index);
  else hipLaunchKernelGGL(ce_row_kernel<float>, dim3((unsigned)rows), dim3(TPB), 0, s, (const float*)logits, ld, targets, row_lse, nll, valid, (int)V, ignore_index);
  FK_CHECK_LAUNCH("fk_ce_loss_fwd(rows)");
  hipLaunchKernelGGL(ce_final_kernel, dim3(1), dim3(TPB), 0, s, (const float*)nll, (const float*)valid, loss2, (int)rows);
  FK_CHECK_LAUNCH("fk_ce_loss_fwd(final)");
  return FK_OK;
}
int fk_ce_loss_bwd(const void* logits, int64_t ld, const int64_t* targets, const float* row_lse, const float* loss2,
                   const float* grad_out, void* dlogits, int64_t ldd, int64_t rows, int64_t V, int64_t ignore_index,
                   int dtype, void* stream) {
  FK_CHECK_ARG(dtype == FK_F32 || dtype == FK_BF16, "fk_ce_loss_bwd: bad dtype %d", dtype);
  FK_CHECK_ARG(logits && targets && row_lse && loss2 && grad_out && dlogits && rows > 0 && V > 0, "fk_ce_loss_bwd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  if (dtype == FK_BF16) hipLaunchKernelGGL(ce_bwd_kernel<bf16_t>, dim3(grid_for(rows * V, 16384)), dim3(TPB), 0, s, (const bf16_t*)logits, ld, targets, row_lse, loss2, grad_out, (bf16_t*)dlogits, ldd, rows, (int)V, ignore_index);
  else hipLaunchKernelGGL(ce_bwd_kernel<float>, dim3(grid_for(rows * V, 16384)), dim3(TPB), 0, s, (const float*)logits, ld, targets, row_lse, loss2, grad_out, (float*)dlogits, ldd, rows, (int)V, ignore_index);
  FK_CHECK_LAUNCH("fk_ce_loss_bwd");
  return FK_OK;
}

int fk_ce_chunk_fwd(const float* logits, int64_t ld, const int64_t* targets, int64_t col0, float* row_m, float* row_s, float* row_t,
                    int64_t rows, int64_t cw, int first, void* stream) {
  FK_CHECK_ARG(logits && targets && row_m && row_s && row_t && rows > 0 && rows < (1LL << 31) && cw > 0 && cw < (1LL << 31) && ld >= cw && col0 >= 0,
               "fk_ce_chunk_fwd: bad arguments");
  hipLaunchKernelGGL(ce_chunk_fwd_kernel, dim3((unsigned)rows), dim3(TPB), 0, (hipStream_t)stream, logits, ld, targets, col0, row_m, row_s, row_t, (int)cw, first);
  FK_CHECK_LAUNCH("fk_ce_chunk_fwd");
  return FK_OK;
}
int fk_ce_chunk_finish(const float* row_m, const float* row_s, const float* row_t, const int64_t* targets, float* row_lse, float* loss2,
                       int64_t rows, int64_t V, int64_t ignore_index, void* workspace, size_t workspace_bytes, void* stream) {
  FK_CHECK_ARG(row_m && row_s && row_t && targets && row_lse && loss2 && rows > 0 && rows < (1LL << 31) && V > 0, "fk_ce_chunk_finish: bad arguments");
  FK_CHECK_ARG(workspace && workspace_bytes >= (size_t)rows * 2 * sizeof(float), "fk_ce_chunk_finish: workspace too small");
  float* nll = (float*)workspace;
  float* valid = nll + rows;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(ce_chunk_rows_kernel, dim3(grid_for(rows, 1024)), dim3(TPB), 0, s, row_m, row_s, row_t, targets, row_lse, nll, valid, rows, V, ignore_index);
  FK_CHECK_LAUNCH("fk_ce_chunk_finish(rows)");
  hipLaunchKernelGGL(ce_final_kernel, dim3(1), dim3(TPB), 0, s, (const float*)nll, (const float*)valid, loss2, (int)rows);
  FK_CHECK_LAUNCH("fk_ce_chunk_finish(final)");
  return FK_OK;
}
int fk_ce_chunk_bwd(const float* logits, int64_t ld, const int64_t* targets, int64_t col0, const float* row_lse, const float* loss2,
                    const float* grad_out, void* dlogits, int64_t ldd, int64_t rows, int64_t cw, int64_t cw_valid, int64_t V,
                    int64_t ignore_index, int dtype, void* stream) {
  FK_CHECK_ARG(dtype == FK_F32 || dtype == FK_BF16, "fk_ce_chunk_bwd: bad dtype %d", dtype);
  FK_CHECK_ARG(logits && targets && row_lse && loss2 && grad_out && dlogits && rows > 0 && cw > 0 && cw_valid >= 0 && cw_valid <= cw && ld >= cw_valid && ldd >= cw,
               "fk_ce_chunk_bwd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  if (dtype == FK_BF16) hipLaunchKernelGGL(ce_chunk_bwd_kernel<bf16_t>, dim3(grid_for(rows * cw, 16384)), dim3(TPB), 0, s, logits, ld, targets, col0, row_lse, loss2, grad_out, (bf16_t*)dlogits, ldd, rows, (int)cw, (int)cw_valid, V, ignore_index);
  else hipLaunchKernelGGL(ce_chunk_bwd_kernel<float>, dim3(grid_for(rows * cw, 16384)), dim3(TPB), 0, s, logits, ld, targets, col0, row_lse, loss2, grad_out, (float*)dlogits, ldd, rows, (int)cw, (int)cw_valid, V, ignore_index);
  FK_CHECK_LAUNCH("fk_ce_chunk_bwd");
  return FK_OK;
}

int fk_adamw_step(float* p, float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2, double eps,
                  double weight_decay, int64_t step, double clip, double grad_scale, int zero_grad, void* stream) {
  FK_CHECK_ARG(p && g && m && v && n > 0 && step >= 1, "fk_adamw_step: bad arguments");
  FK_CHECK_ARG((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "fk_adamw_step: buffers must be 16-byte aligned");
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n / 4 + 1, 4096)), dim3(TPB), 0, (hipStream_t)stream, p, g, m, v, n, (float)lr, (float)(1.0 - beta1),
                     (float)beta2, (float)(1.0 - beta2), (float)eps, (float)weight_decay, (float)bc1, (float)sqrt(bc2), (float)clip, (float)grad_scale, zero_grad);
  FK_CHECK_LAUNCH("fk_adamw_step");
  return FK_OK;
}

}  // extern "C"
