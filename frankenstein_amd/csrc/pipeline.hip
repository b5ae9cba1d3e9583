// pipeline.hip — the input pipeline either side of the hot path, on device:
//   fk_block_stats        per-(recording block, channel) mean / std over all time steps of the block's trials
//                         (utils/data_utils.py:136-147: mean(axis=0), std(axis=0), std == 0 -> 1)
//   fk_zscore_smooth_pad  (x - mean) / std, Gaussian smoothing over time (scipy.ndimage.gaussian_filter1d, sigma 1, radius 4,
//                         mode 'reflect', :150-153), then zero-pad / truncate every trial to Tmax rows (:243-267)
// Trials are packed row-wise: trial i = rows off[i] .. off[i+1]-1 of x [total_rows, C] (fp32); block[i] in [0, nblocks).
// HBM-bound streaming; statistics accumulate in fp64 and are reduced in trial order (deterministic, no atomics).
#include "fk_common.h"

namespace {

// partial[i][c] = (sum, sum of squares) of trial i, channel c
__global__ void trial_sums_kernel(const float* x, const int64_t* off, int64_t C, double* partial) {
  const int64_t i = blockIdx.x;
  const int64_t r0 = off[i], r1 = off[i + 1];
  for (int64_t c = blockIdx.y * (int64_t)blockDim.x + threadIdx.x; c < C; c += (int64_t)gridDim.y * blockDim.x) {
    double s = 0.0, q = 0.0;
    for (int64_t r = r0; r < r1; ++r) {
      const double v = (double)x[r * C + c];
      s += v;
      q += v * v;
    }
    partial[(i * C + c) * 2] = s;
    partial[(i * C + c) * 2 + 1] = q;
  }
}
__global__ void block_stats_kernel(const double* partial, const int64_t* off, const int32_t* block, int64_t ntrials, int64_t C,
                                   float* mean, float* stdv) {
  const int64_t b = blockIdx.y;
  const int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s = 0.0, q = 0.0, n = 0.0;
  for (int64_t i = 0; i < ntrials; ++i) {
    if (block[i] != (int32_t)b) continue;
    s += partial[(i * C + c) * 2];
    q += partial[(i * C + c) * 2 + 1];
    n += (double)(off[i + 1] - off[i]);
  }
  double m = 0.0, sd = 1.0;
  if (n > 0.0) {
    m = s / n;
    const double var = fmax(q / n - m * m, 0.0);
    sd = sqrt(var);
    if (sd == 0.0) sd = 1.0;
  }
  mean[b * C + c] = (float)m;
  stdv[b * C + c] = (float)sd;
}

// scipy 'reflect' (half-sample symmetric): ... c b a | a b c ... | c b a ...
__device__ inline int64_t reflect_idx(int64_t t, int64_t n) {
  if (n == 1) return 0;
  const int64_t period = 2 * n;
  t %= period;
  if (t < 0) t += period;
  return t < n ? t : period - 1 - t;
}

struct Taps { float w[9]; };

__global__ void zscore_smooth_pad_kernel(const float* x, const int64_t* off, const int32_t* block, const float* mean,
                                         const float* stdv, float* out, int64_t C, int64_t Tmax, Taps taps) {
  const int64_t i = blockIdx.x;
  const int64_t r0 = off[i], len = off[i + 1] - r0;
  const int32_t b = block[i];
  for (int64_t c = threadIdx.x; c < C; c += blockDim.x) {
    const float m = mean[(int64_t)b * C + c], inv = 1.0f / stdv[(int64_t)b * C + c];
    for (int64_t t = blockIdx.y; t < Tmax; t += gridDim.y) {
      float acc = 0.0f;
      if (t < len) {
#pragma unroll
        for (int k = 0; k < 9; ++k) {
          const int64_t tt = reflect_idx(t + k - 4, len);
          acc += taps.w[k] * ((x[(r0 + tt) * C + c] - m) * inv);
        }
      }
      out[(i * Tmax + t) * C + c] = acc;
    }
  }
}

}  // namespace

extern "C" {

size_t fk_block_stats_workspace_bytes(int64_t ntrials, int64_t C) { return (size_t)ntrials * C * 2 * sizeof(double); }

int fk_block_stats(const float* x, const int64_t* off, const int32_t* block, int64_t ntrials, int64_t C, int64_t nblocks,
                   float* mean, float* stdv, void* workspace, size_t workspace_bytes, void* stream) {
  FK_CHECK_ARG(x && off && block && mean && stdv && ntrials > 0 && C > 0 && nblocks > 0 && ntrials < (1LL << 31) && nblocks < 65536,
               "fk_block_stats: bad arguments");
  FK_CHECK_ARG(workspace && workspace_bytes >= fk_block_stats_workspace_bytes(ntrials, C), "fk_block_stats: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const unsigned cy = (unsigned)fk_cdiv(C, 256);
  hipLaunchKernelGGL(trial_sums_kernel, dim3((unsigned)ntrials, cy), dim3(256), 0, s, x, off, C, (double*)workspace);
  FK_CHECK_LAUNCH("fk_block_stats(sums)");
  hipLaunchKernelGGL(block_stats_kernel, dim3(cy, (unsigned)nblocks), dim3(256), 0, s, (const double*)workspace, off, block, ntrials, C, mean, stdv);
  FK_CHECK_LAUNCH("fk_block_stats");
  return FK_OK;
}

int fk_zscore_smooth_pad(const float* x, const int64_t* off, const int32_t* block, const float* mean, const float* stdv,
                         float* out, int64_t ntrials, int64_t C, int64_t Tmax, double sigma, void* stream) {
  FK_CHECK_ARG(x && off && block && mean && stdv && out && ntrials > 0 && C > 0 && Tmax > 0 && ntrials < (1LL << 31),
               "fk_zscore_smooth_pad: bad arguments");
  FK_CHECK_ARG(sigma > 0.0 && (int)(4.0 * sigma + 0.5) == 4, "fk_zscore_smooth_pad: only a radius-4 kernel (sigma ~ 1) is built");
  Taps taps;
  double w[9], sum = 0.0;
  for (int k = 0; k < 9; ++k) { const double d = (double)(k - 4); w[k] = exp(-0.5 * d * d / (sigma * sigma)); sum += w[k]; }
  for (int k = 0; k < 9; ++k) taps.w[k] = (float)(w[k] / sum);
  hipStream_t s = (hipStream_t)stream;
  const unsigned ty = (unsigned)(Tmax < 64 ? Tmax : 64);
  hipLaunchKernelGGL(zscore_smooth_pad_kernel, dim3((unsigned)ntrials, ty), dim3(256), 0, s, x, off, block, mean, stdv, out, C, Tmax, taps);
  FK_CHECK_LAUNCH("fk_zscore_smooth_pad");
  return FK_OK;
}

}  // extern "C"
