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


// ---- sampling tail of GPT.generate on the device (models/gpt2_model.py:340-351): logits / temperature -> top-k crop (everything below
// the k-th largest value becomes -inf; ties with it stay, like `logits < v[:, [-1]]`) -> softmax -> one multinomial draw.  One block per
// row; the k-th largest value by a 4-pass radix select over the order-preserving integer image of the fp32 logits; the draw by inverse
// CDF in index order with a Philox4x32-10 uniform keyed by (seed; step counter, row), so a captured graph draws fresh numbers on
// every replay: the LAST block to finish advances the device-side step counter (and the decode position) for the next replay.
constexpr int SAMPLE_THREADS = 1024;

FK_DEV unsigned f32_sortable(float x) {          // monotone: a < b  <=>  key(a) < key(b)   (-0 < +0, NaNs above +inf)
  const unsigned u = __float_as_uint(x);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
FK_DEV void philox4x32_10(unsigned k0, unsigned k1, unsigned (&c)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1, n3 = (unsigned)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}
template <typename F> FK_DEV float block_reduce(float v, float* red, F op, float ident) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = op(v, __shfl_xor(v, o, 64));
  __syncthreads();
  if (lane == 0) red[wv] = v;
  __syncthreads();
  float r = ident;
  for (int i = 0; i < SAMPLE_THREADS / 64; ++i) r = op(r, red[i]);
  return r;
}

__global__ __launch_bounds__(SAMPLE_THREADS) void sample_topk_kernel(const float* logits, int64_t ld, int V, float inv_temp, int top_k,
                                                                     const unsigned long long* seed, int64_t* step, int32_t* pos_inc,
                                                                     int64_t* cur, int64_t* out, int64_t out_ld, int64_t out_cols, unsigned* ticket) {
  __shared__ unsigned hist[256];
  __shared__ float red[SAMPLE_THREADS / 64];
  __shared__ float scan[SAMPLE_THREADS];
  __shared__ unsigned sel_prefix, sel_remaining;
  __shared__ int winner;
  const int tid = threadIdx.x, b = blockIdx.x;
  const float* row = logits + (int64_t)b * ld;
  const int64_t my_step = step[0];                       // read before anybody can advance it (the advance happens after the last block)

  // ---- k-th largest key (top_k <= 0 or >= V: keep everything)
  unsigned kth = 0u;
  if (top_k > 0 && top_k < V) {
    if (tid == 0) { sel_prefix = 0u; sel_remaining = (unsigned)top_k; }
    for (int pass = 0; pass < 4; ++pass) {
      const int shift = 24 - 8 * pass;
      if (tid < 256) hist[tid] = 0u;
      __syncthreads();
      const unsigned prefix = sel_prefix, mask_hi = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
      for (int i = tid; i < V; i += SAMPLE_THREADS) {
        const unsigned key = f32_sortable(row[i] * inv_temp);
        if ((key & mask_hi) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
      }
      __syncthreads();
      if (tid == 0) {                                     // walk the 256 bins from the top until the k-th largest falls into one
        unsigned rem = sel_remaining;
        int bin = 255;
        for (; bin > 0; --bin) {
          if (hist[bin] >= rem) break;
          rem -= hist[bin];
        }
        sel_prefix = prefix | ((unsigned)bin << shift);
        sel_remaining = rem;
      }
      __syncthreads();
    }
    kth = sel_prefix;
  }
  // ---- softmax over the kept logits: maximum, then the sum of exp; per-thread partial sums of CONTIGUOUS index chunks for the draw
  float mx = -INFINITY;
  for (int i = tid; i < V; i += SAMPLE_THREADS) {
    const float x = row[i] * inv_temp;
    if (f32_sortable(x) >= kth) mx = fmaxf(mx, x);
  }
  mx = block_reduce(mx, red, [](float a, float c) { return fmaxf(a, c); }, -INFINITY);
  const int chunk = (V + SAMPLE_THREADS - 1) / SAMPLE_THREADS, i0 = tid * chunk, i1 = min(V, i0 + chunk);
  float part = 0.0f;
  for (int i = i0; i < i1; ++i) {
    const float x = row[i] * inv_temp;
    if (f32_sortable(x) >= kth) part += __expf(x - mx);
  }
  scan[tid] = part;
  __syncthreads();
  for (int o = 1; o < SAMPLE_THREADS; o <<= 1) {          // inclusive scan of the chunk sums (Hillis-Steele)
    const float add = tid >= o ? scan[tid - o] : 0.0f;
    __syncthreads();
    scan[tid] += add;
    __syncthreads();
  }
  const float total = scan[SAMPLE_THREADS - 1];
  // ---- one uniform in [0, 1) and the first index whose cumulative mass exceeds u * total
  unsigned c[4] = {(unsigned)my_step, (unsigned)((unsigned long long)my_step >> 32), (unsigned)b, 0x5A3Cu};
  philox4x32_10((unsigned)seed[0], (unsigned)(seed[0] >> 32), c);
  const float target = (float)(c[0] >> 8) * (1.0f / 16777216.0f) * total;
  if (tid == 0) winner = -1;
  __syncthreads();
  const float before = tid == 0 ? 0.0f : scan[tid - 1];
  if (part > 0.0f && before <= target && target < scan[tid]) {
    float acc = before;
    int pick = -1;
    for (int i = i0; i < i1; ++i) {
      const float x = row[i] * inv_temp;
      if (f32_sortable(x) >= kth) {
        acc += __expf(x - mx);
        pick = i;                                         // last kept index seen: the fallback when rounding leaves acc <= target
        if (acc > target) break;
      }
    }
    winner = pick;
  }
  __syncthreads();
  if (tid == 0) {
    int w = winner;
    if (w < 0) {                                          // target >= total by rounding: the last kept index
      for (int i = V - 1; i >= 0; --i)
        if (f32_sortable(row[i] * inv_temp) >= kth) { w = i; break; }
    }
    cur[b] = w;
    if (out && my_step < out_cols) out[(int64_t)b * out_ld + my_step] = w;       // a step counter past the buffer is not a write past it
    __threadfence();
    if (atomicAdd(ticket, 1u) == gridDim.x - 1) {         // every block has read step[0] and written its token
      ticket[0] = 0u;
      step[0] = my_step + 1;
      if (pos_inc) pos_inc[0] += 1;
    }
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

int fk_sample_topk(const float* logits, int64_t ld, int64_t B, int64_t V, float temperature, int64_t top_k, const uint64_t* seed,
                   int64_t* step, int32_t* pos_inc, int64_t* cur, int64_t* out, int64_t out_ld, int64_t out_cols, uint32_t* ticket,
                   void* stream) {
  FK_CHECK_ARG(out == nullptr || (out_cols > 0 && out_cols <= out_ld), "fk_sample_topk: out given without its width (out_cols=%lld, out_ld=%lld)",
               (long long)out_cols, (long long)out_ld);
  FK_CHECK_ARG(logits && seed && step && cur && ticket && B > 0 && B < 65536 && V > 0 && V < (1LL << 31) && ld >= V && temperature > 0.0f,
               "fk_sample_topk: bad arguments (B=%lld V=%lld temperature=%g)", (long long)B, (long long)V, (double)temperature);
  hipLaunchKernelGGL(sample_topk_kernel, dim3((unsigned)B), dim3(SAMPLE_THREADS), 0, (hipStream_t)stream, logits, ld, (int)V, 1.0f / temperature,
                     (int)(top_k > 0 && top_k < V ? top_k : 0), (const unsigned long long*)seed, step, pos_inc, cur, out, out_ld, out_cols, ticket);
  FK_CHECK_LAUNCH("fk_sample_topk");
  return FK_OK;
}

}  // extern "C"
