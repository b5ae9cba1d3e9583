// decode.hip — single-token decode step with the position on the DEVICE, so that one captured hipGraph replays for every new
// token (the reference re-runs the whole sequence per token, models/gpt2_model.py:336-340).
//   fk_gpt_embed_step  x[b, :] = wte[idx[b], :] + wpe[*pos, :]                               (models/gpt2_model.py:183-196, t = 1)
//   fk_kv_append       kv[b, *pos, :] = qkv[b, d : 3d]   (key | value rows of the new token into the per-layer cache)
//   fk_attn_decode     o[b, h, :] = softmax_j( q[b,h,:] . k[b,j,h,:] * scale ) v[b,j,h,:],  j = 0 .. *pos   (causal, one query)
// All three read the position from a device int32 (bumped by the host graph between steps).  HBM / latency-bound, no MFMA.
#include "fk_common.h"

namespace {

template <typename T>
__global__ void gpt_embed_step_kernel(const int64_t* idx, const float* wte, const float* wpe, const int32_t* pos, T* out, int dim, int64_t vocab) {
  const int b = blockIdx.x;
  int64_t tok = idx[b];
  tok = tok < 0 ? 0 : (tok >= vocab ? vocab - 1 : tok);
  const float* e = wte + tok * dim;
  const float* pe = wpe + (int64_t)pos[0] * dim;
  for (int c = threadIdx.x; c < dim; c += blockDim.x) out[(int64_t)b * dim + c] = from_f32<T>(e[c] + pe[c]);
}

template <typename T>
__global__ void kv_append_kernel(const T* qkv, T* kv, const int32_t* pos, int d, int64_t tmax) {
  const int b = blockIdx.x;
  const T* src = qkv + (int64_t)b * 3 * d + d;
  T* dst = kv + ((int64_t)b * tmax + pos[0]) * 2 * d;
  for (int c = threadIdx.x; c < 2 * d; c += blockDim.x) dst[c] = src[c];
}

// block = (b, h), 256 threads; thread t owns keys t, t + 256, ...; partial (max, sum, o[D]) merged wave- then block-wide
template <typename T, int D>
__global__ __launch_bounds__(256) void attn_decode_kernel(const T* q, int64_t q_bs, const T* kv, int64_t kv_bs, int64_t kv_rs, T* out,
                                                          int64_t o_bs, const int32_t* pos, int H, float scale) {
  __shared__ float sq[D];
  __shared__ float red[4][D + 2];
  const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nk = pos[0] + 1;
  const int d_model = H * D;
  if (tid < D) sq[tid] = to_f32<T>(q[(int64_t)b * q_bs + h * D + tid]) * scale;
  __syncthreads();
  float m = -INFINITY, l = 0.0f, o[D];
#pragma unroll
  for (int i = 0; i < D; ++i) o[i] = 0.0f;
  for (int j = tid; j < nk; j += 256) {
    const T* kr = kv + (int64_t)b * kv_bs + (int64_t)j * kv_rs + h * D;
    const T* vr = kr + d_model;
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < D; ++i) s += sq[i] * to_f32<T>(kr[i]);
    const float mn = fmaxf(m, s), a = __expf(m - mn), pj = __expf(s - mn);
    l = l * a + pj;
#pragma unroll
    for (int i = 0; i < D; ++i) o[i] = o[i] * a + pj * to_f32<T>(vr[i]);
    m = mn;
  }
  // wave merge
  float wm = m;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) wm = fmaxf(wm, __shfl_xor(wm, off, 64));
  const float w = (m == -INFINITY) ? 0.0f : __expf(m - wm);
  l *= w;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) l += __shfl_xor(l, off, 64);
#pragma unroll
  for (int i = 0; i < D; ++i) {
    float v = o[i] * w;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    o[i] = v;
  }
  if (lane == 0) {
    red[wave][D] = wm;
    red[wave][D + 1] = l;
#pragma unroll
    for (int i = 0; i < D; ++i) red[wave][i] = o[i];
  }
  __syncthreads();
  if (tid < D) {
    const float M4 = fmaxf(fmaxf(red[0][D], red[1][D]), fmaxf(red[2][D], red[3][D]));
    float L = 0.0f, acc = 0.0f;
#pragma unroll
    for (int wv = 0; wv < 4; ++wv) {
      const float ww = (red[wv][D] == -INFINITY) ? 0.0f : __expf(red[wv][D] - M4);
      L += red[wv][D + 1] * ww;
      acc += red[wv][tid] * ww;
    }
    out[(int64_t)b * o_bs + h * D + tid] = from_f32<T>(acc / L);
  }
}

}  // namespace

extern "C" {

int fk_gpt_embed_step(const int64_t* idx, const float* wte, const float* wpe, const int32_t* pos, void* out, int64_t B, int64_t dim,
                      int64_t vocab, int dtype, void* stream) {
  FK_CHECK_ARG((dtype == FK_F32 || dtype == FK_BF16) && idx && wte && wpe && pos && out && B > 0 && dim > 0 && vocab > 0, "fk_gpt_embed_step: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  if (dtype == FK_BF16) hipLaunchKernelGGL(gpt_embed_step_kernel<bf16_t>, dim3((unsigned)B), dim3(128), 0, s, idx, wte, wpe, pos, (bf16_t*)out, (int)dim, vocab);
  else hipLaunchKernelGGL(gpt_embed_step_kernel<float>, dim3((unsigned)B), dim3(128), 0, s, idx, wte, wpe, pos, (float*)out, (int)dim, vocab);
  FK_CHECK_LAUNCH("fk_gpt_embed_step");
  return FK_OK;
}

int fk_kv_append(const void* qkv, void* kv, const int32_t* pos, int64_t B, int64_t d, int64_t tmax, int dtype, void* stream) {
  FK_CHECK_ARG((dtype == FK_F32 || dtype == FK_BF16) && qkv && kv && pos && B > 0 && d > 0 && tmax > 0, "fk_kv_append: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  if (dtype == FK_BF16) hipLaunchKernelGGL(kv_append_kernel<bf16_t>, dim3((unsigned)B), dim3(256), 0, s, (const bf16_t*)qkv, (bf16_t*)kv, pos, (int)d, tmax);
  else hipLaunchKernelGGL(kv_append_kernel<float>, dim3((unsigned)B), dim3(256), 0, s, (const float*)qkv, (float*)kv, pos, (int)d, tmax);
  FK_CHECK_LAUNCH("fk_kv_append");
  return FK_OK;
}

int fk_attn_decode(const void* q, int64_t q_bs, const void* kv, int64_t kv_bs, int64_t kv_rs, void* out, int64_t o_bs, const int32_t* pos,
                   int64_t B, int64_t H, int64_t D, float scale, int dtype, void* stream) {
  FK_CHECK_ARG((dtype == FK_F32 || dtype == FK_BF16) && q && kv && out && pos && B > 0 && H > 0, "fk_attn_decode: bad arguments");
  FK_CHECK_ARG(D == 16 || D == 32 || D == 64 || D == 128, "fk_attn_decode: head_dim %lld not in {16, 32, 64, 128}", (long long)D);
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)H, (unsigned)B), block(256);
#define FK_AD(TT, DD) hipLaunchKernelGGL((attn_decode_kernel<TT, DD>), grid, block, 0, s, (const TT*)q, q_bs, (const TT*)kv, kv_bs, kv_rs, (TT*)out, o_bs, pos, (int)H, scale)
  if (dtype == FK_BF16) { if (D == 16) FK_AD(bf16_t, 16); else if (D == 32) FK_AD(bf16_t, 32); else if (D == 64) FK_AD(bf16_t, 64); else FK_AD(bf16_t, 128); }
  else { if (D == 16) FK_AD(float, 16); else if (D == 32) FK_AD(float, 32); else if (D == 64) FK_AD(float, 64); else FK_AD(float, 128); }
#undef FK_AD
  FK_CHECK_LAUNCH("fk_attn_decode");
  return FK_OK;
}

}  // extern "C"
