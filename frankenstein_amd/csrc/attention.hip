// attention.hip — flash-style fused attention forward / backward for gfx950.
//
// Replaces F.scaled_dot_product_attention at models/brainformer.py:168 (self, bool block-causal mask),
// :215 (cross, no mask) and models/gpt2_model.py:64 (is_causal) of the reference.  The [N,N] boolean
// mask of models/brainformer.py:93-111 is never materialised: it is the analytic predicate
// (k + k_off) / C <= (q + q_off) / C, evaluated per tile (and per element only on boundary tiles).
//
// Layout: Q/K/V/O are [B, N, H, D] views (row stride and batch stride in elements, head h at column
// h*D) — exactly what the QKV projection GEMM writes, so there is no transpose pass.
//
// Orientation ("query on the lane"): S^T = K Q^T, so a lane owns one query column; row max / row sum /
// rescale are per-lane scalars (one cross-half shuffle), and the S^T accumulator registers are directly
// the B operand of O^T += V^T P^T (fk_common.h: acc_row_of_slot).  K is read from LDS row-wise
// (ds_read_b128), V through the transposed read (ds_read_b64_tr_b16) for bf16.
// Backward = two kernels (no atomics, deterministic):
//   dkdv: one workgroup per 128 keys, wave = 32 keys ("key on the lane"): S = Q K^T, dP = dO V^T,
//         dV^T += dO^T P, dK^T += Q^T dS, sweeping the visible query tiles;
//   dq  : forward-shaped: S^T, dP^T = V dO^T, dQ^T += K^T dS^T.
// P is recomputed from the saved log-sum-exp; delta = rowsum(dO * O) is computed by the dQ kernel (which runs first) and read by dK/dV.
// fp32 variants (exact-fp32 MFMA) exist for the parity mode.
#include "fk_common.h"
#include <type_traits>

namespace {

constexpr int NT = 256;     // threads per workgroup (4 waves)
constexpr int BQ = 128;     // query rows per workgroup (fwd, dq): 32 per wave
constexpr int BKV = 64;     // keys per staged tile (fwd, dq)
constexpr float LOG2E = 1.4426950408889634f;

struct AttnArgs {
  const void *Q, *K, *V, *O, *dO;
  void *Out, *dQ, *dK, *dV;
  float *LSE, *delta;
  int64_t q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs;
  int B, H, Nq, Nk;
  int mask_kind, mask_c, q_off, k_off;
  float scale;
  // backward only: inverse RoPE (rotation by -angle) applied to dQ / dK as they are stored (fuses the rope backward)
  const float* rope_table; int64_t rope_bs; int rope_off;
  // FK_MASK_PREFIX: visible(q, k) = k < limits[b, q]  <=>  q >= qfirst[b, k]  (both non-decreasing; built by fk_prefix_mask
  // from sorted per-token block ids: the per-sample sub-mask of MAE, models/brainformer.py:392-413)
  const int* limits; const int* qfirst;
  int flags;          // FK_ATTN_Q_PRESCALED
  // dropout on the attention probabilities (fk_attn_*_dropout; generic kernels only): keep <=> bits >= drop_thresh, kept entries times
  // drop_scale = 1 / (1 - p); seed words in device memory (fk_common.h "dropout").  drop_thresh == 0: no dropout.
  const unsigned* drop_seed; unsigned drop_site, drop_thresh; float drop_scale;
};

template <typename T, int D> struct AT {
  static constexpr int ES = sizeof(T);
  static constexpr int RB = D * ES;                 // bytes per head row
  static constexpr int CPR = RB / 16;               // 16-byte chunks per row
  static constexpr int KSTEPS = (D + 15) / 16;      // k16 steps across the head dim (D = 8: upper half-slots are zero)
  static constexpr int DT = (D + 31) / 32;          // 32-row tiles of O^T / dK^T / dV^T
  static constexpr int DPAD = DT * 32;
  static constexpr int RSTRIDE = DPAD * ES + 16;    // row-read / dual-use LDS image stride (padded: conflict-free b128)
  static constexpr int VRB = DPAD * ES;
  static constexpr int VSTRIDE = (ES == 2) ? ((((VRB / 64) & 1) != 0) ? VRB : VRB + 64) : VRB + 16;  // tr-only image
  static constexpr int VEC = 16 / ES;
};

FK_DEV bool visible(int kind, int c, int qpos, int kpos) {
  if (kind == FK_MASK_CAUSAL) return kpos <= qpos;
  if (kind == FK_MASK_BLOCK_CAUSAL) return (kpos / c) <= (qpos / c);
  return true;
}
// exclusive upper bound of key INDICES visible to query index q (monotone predicates only)
// FK_MASK_DENSE: an arbitrary boolean mask (models/brainformer.py:160-168 passes whatever it is given): `limits` points to uint8
// [Bm, Hm, Nq, Nk] (non-zero = attend), mask_c is the batch stride and q_off the head stride in elements (0: one mask for every sample /
// every head; the position offsets q_off / k_off have no meaning for a table)
FK_DEV bool dense_vis(const AttnArgs& p, int b, int hd, int q, int k) {
  return reinterpret_cast<const unsigned char*>(p.limits)[(int64_t)b * p.mask_c + (int64_t)hd * p.q_off + (int64_t)q * p.Nk + k] != 0;
}
FK_DEV int kv_limit(const AttnArgs& p, int b, int q) {
  if (p.mask_kind == FK_MASK_PREFIX) return min(p.Nk, p.limits[(int64_t)b * p.Nq + q]);
  if (p.mask_kind == FK_MASK_KEYPAD || p.mask_kind == FK_MASK_DENSE) return 0;        // no structure: every tile takes the per-element path
  const int qpos = q + p.q_off;
  if (p.mask_kind == FK_MASK_CAUSAL) return min(p.Nk, max(0, qpos - p.k_off + 1));
  if (p.mask_kind == FK_MASK_BLOCK_CAUSAL) return min(p.Nk, max(0, (qpos / p.mask_c + 1) * p.mask_c - p.k_off));
  return p.Nk;
}
// smallest query INDEX that can see key index k
FK_DEV int q_first(const AttnArgs& p, int b, int k) {
  if (p.mask_kind == FK_MASK_PREFIX) return p.qfirst[(int64_t)b * p.Nk + k];
  if (p.mask_kind == FK_MASK_KEYPAD || p.mask_kind == FK_MASK_DENSE) return 0x3fffffff;   // 'no query sees every key for free' -> always the predicate path
  const int kpos = k + p.k_off;
  if (p.mask_kind == FK_MASK_CAUSAL) return max(0, kpos - p.q_off);
  if (p.mask_kind == FK_MASK_BLOCK_CAUSAL) return max(0, (kpos / p.mask_c) * p.mask_c - p.q_off);
  return 0;
}

// FK_MASK_KEYPAD with nothing (or only a tail) padded: the number of leading entries of a validity row that are all non-zero, so the tiles
// in front of the first padded position take the mask-free path (SimpleMAE passes a padding mask with every sample, models/mae.py:119-150,
// and most rows of it are all-valid).  Wave-uniform; every lane of the wave calls it.  The 16 loads of a round are independent (one
// exposed latency per 1024 entries), and the round loop ends at the first round that found a zero.
FK_DEV int keypad_valid_prefix(const int* valid, int n, int lane) {
  int first_bad = n;
  for (int base = 0; base < n; base += 1024) {
    int v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int k = base + 64 * i + lane;
      v[i] = k < n ? valid[k] : 1;
    }
#pragma unroll
    for (int i = 15; i >= 0; --i)
      if (v[i] == 0) first_bad = base + 64 * i + lane;
    if (__builtin_amdgcn_ballot_w64(first_bad != n) != 0) break;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) first_bad = min(first_bad, __shfl_xor(first_bad, off, 64));
  return __builtin_amdgcn_readfirstlane(first_bad);
}

// ---- LDS fragment reads -------------------------------------------------------------------------
// row-read: 8 contiguous head-dim elements (slots d = 16 s + 8 h + e) of image row `row`
template <typename T> FK_DEV void frag_row(Frag<T>& f, const char* img, int stride, int row, int s, int h) {
  frag_load_contig<T>(f, reinterpret_cast<const T*>(img + row * stride) + 16 * s + 8 * h);
}
// transposed read paired with an accumulator-as-B operand: output row = image column cb + (lane & 31),
// slot (h, e) = image row  rb + 16 s + 8 (e >> 2) + 4 h + (e & 3)
template <typename T> FK_DEV void frag_tr(Frag<T>& f, const char* img, int stride, int rb, int s, int cb, int lane);
template <> FK_DEV void frag_tr<bf16_t>(Frag<bf16_t>& f, const char* img, int stride, int rb, int s, int cb, int lane) {
  const int g = lane >> 4, i = lane & 15, h = g >> 1;
  const int col = cb + 16 * (g & 1) + 4 * (i & 3);
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int row = rb + 16 * s + 8 * t + 4 * h + (i >> 2);
    bf16x4 v = lds_read_tr4(reinterpret_cast<const bf16_t*>(img + row * stride) + col);
#pragma unroll
    for (int e = 0; e < 4; ++e) f.v[4 * t + e] = v[e];
  }
}
template <> FK_DEV void frag_tr<float>(Frag<float>& f, const char* img, int stride, int rb, int s, int cb, int lane) {
  const int h = lane >> 5, c = cb + (lane & 31);
#pragma unroll
  for (int e = 0; e < 8; ++e)
    f.v[e] = reinterpret_cast<const float*>(img + (rb + acc_row_of_slot(s, h, e)) * stride)[c];
}

// ---- dual-use LDS image (row reads AND transposed reads of the same tile) ----------------------------
// bf16, D = 64 (128-byte rows, the benchmark shape): no padding, 16-byte chunk c of row r lives at chunk
// c ^ f(r >> 1) with f(g) = g ^ ((g & 1) << 2).  ds_read_b128 row reads (16 rows distinct mod 16, one chunk) and
// ds_read_b64_tr_b16 reads (4 consecutive rows x 4 consecutive chunks per 32-lane half) are both conflict free.
// Other shapes keep the padded rows of AT::RSTRIDE (row reads conflict free, transposed reads 2-way).
template <typename T, int D> struct Img {
  static constexpr bool SWZ = (sizeof(T) == 2 && D == 64);
  static constexpr int STRIDE = SWZ ? 128 : AT<T, D>::RSTRIDE;
  FK_DEV static int off(int row, int bytecol) {
    if constexpr (SWZ) {
      const int g = (row >> 1) & 7, f = g ^ ((g & 1) << 2);
      return row * 128 + ((((bytecol >> 4) ^ f) << 4) | (bytecol & 15));
    } else {
      return row * STRIDE + bytecol;
    }
  }
};
template <typename T, int D> FK_DEV void img_row(Frag<T>& f, const char* img, int row, int s, int h) {
  if constexpr (sizeof(T) == 2) {
    f.v = *reinterpret_cast<const bf16x8*>(img + Img<T, D>::off(row, (16 * s + 8 * h) * 2));
  } else {
    frag_load_contig<T>(f, reinterpret_cast<const T*>(img + Img<T, D>::off(row, (16 * s + 8 * h) * 4)));
  }
}
template <typename T, int D> FK_DEV void img_tr(Frag<T>& f, const char* img, int rb, int s, int cb, int lane) {
  if constexpr (sizeof(T) == 2) {
    const int g = lane >> 4, i = lane & 15, h = g >> 1;
    const int col = cb + 16 * (g & 1) + 4 * (i & 3);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int row = rb + 16 * s + 8 * t + 4 * h + (i >> 2);
      bf16x4 v = lds_read_tr4(reinterpret_cast<const bf16_t*>(img + Img<T, D>::off(row, col * 2)));
#pragma unroll
      for (int e = 0; e < 4; ++e) f.v[4 * t + e] = v[e];
    }
  } else {
    frag_tr<T>(f, img, Img<T, D>::STRIDE, rb, s, cb, lane);
  }
}

// LDS-DMA (global_load_lds_dwordx4) loader for a [ROWS][64] bf16 head tile in the Img<bf16,64> layout: one wave
// instruction fills 8 image rows (1 KiB, lane-linear), so the chunk swizzle is applied to the per-lane source address.
// Rows past nrows are clamped to the last valid row (finite data; the mask / LSE=+inf zeroes their contribution).
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;
template <int ROWS, int NW = 4>
FK_DEV void dma_tile_bf16_d64(const bf16_t* base, int64_t rs, int row0, int nrows, char* img, int wave, int lane) {
  constexpr int PER_WAVE = ROWS / 8 / NW;   // wave instructions per wave
#pragma unroll
  for (int j = 0; j < PER_WAVE; ++j) {
    const int grp = wave * PER_WAVE + j;               // 8-row group
    const int row = grp * 8 + (lane >> 3);
    const int g = (row >> 1) & 7, f = g ^ ((g & 1) << 2);
    const int ch = (lane & 7) ^ f;                     // logical chunk stored at physical chunk (lane & 7)
    // 32-bit element offset from the (wave-uniform) head base: one full-rate v_mad_u32_u24 instead of a 64-bit multiply, and the
    // address goes out as SGPR base + VGPR offset.  launch_*() checks rows and row strides < 2^24.
    const unsigned off = __umul24((unsigned)min(row0 + row, nrows - 1), (unsigned)rs) + (unsigned)(ch * 8);
    const bf16_t* src = base + off;
    __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)(img + grp * 1024), 16, 0, 0);
  }
}

// one 8-row group (one wave instruction) of such a tile
FK_DEV void dma_group_bf16_d64(const bf16_t* base, int64_t rs, int row0, int nrows, char* img, int grp, int lane) {
  const int row = grp * 8 + (lane >> 3);
  const int g = (row >> 1) & 7, f = g ^ ((g & 1) << 2);
  const unsigned off = __umul24((unsigned)min(row0 + row, nrows - 1), (unsigned)rs) + (unsigned)(((lane & 7) ^ f) * 8);
  __builtin_amdgcn_global_load_lds((glb_void_t*)(base + off), (lds_void_t*)(img + grp * 1024), 16, 0, 0);
}

// The same loader as a cursor over consecutive tiles: the per-lane source pointers are computed once and advanced by ROWS rows per
// tile (one 64-bit add per instruction instead of a clamp, two 32-bit multiplies and a 64-bit multiply-add, all quarter-rate VALU work
// that sat in every tile of the hot loops); only a tile that reaches past `nrows` takes the clamping path above.  Used where registers
// allow (dQ): the forward (128) and dK/dV (256) kernels sit exactly at their occupancy limits and the 4 cursor registers would spill.
template <int ROWS, int NW = 4>
struct DmaCursor {
  static constexpr int PER_WAVE = ROWS / 8 / NW;
  const bf16_t* src[PER_WAVE];                       // the only state: the caller passes the tile's row0 again (it has it anyway)
  FK_DEV void init(const bf16_t* base, int64_t rs, int row0, int wave, int lane) {
#pragma unroll
    for (int j = 0; j < PER_WAVE; ++j) {
      const int grp = wave * PER_WAVE + j, row = grp * 8 + (lane >> 3);
      const int g = (row >> 1) & 7, f = g ^ ((g & 1) << 2);
      src[j] = base + (int64_t)(row0 + row) * rs + ((lane & 7) ^ f) * 8;
    }
  }
  // fetch the tile at the cursor (rows row0 .. row0 + ROWS) into img and move the cursor one tile on
  FK_DEV void next(const bf16_t* base, int64_t rs, int row0, int nrows, char* img, int wave, int lane) {
    if (row0 + ROWS <= nrows) {                      // wave-uniform
#pragma unroll
      for (int j = 0; j < PER_WAVE; ++j) {
        __builtin_amdgcn_global_load_lds((glb_void_t*)src[j], (lds_void_t*)(img + (wave * PER_WAVE + j) * 1024), 16, 0, 0);
        src[j] += (int64_t)ROWS * rs;
      }
    } else {
      dma_tile_bf16_d64<ROWS, NW>(base, rs, row0, nrows, img, wave, lane);
    }
  }
};

// ---- global -> register -> LDS staging of a [ROWS][D] head tile -----------------------------------
template <typename T, int D, int ROWS> struct Stager {
  using C = AT<T, D>;
  static constexpr int TOTAL = ROWS * C::CPR;
  static constexpr int NCH = (TOTAL + NT - 1) / NT;
  u32x4 r[NCH];
  FK_DEV void load(const T* base, int64_t rs, int row0, int nrows, int tid) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int id = tid + NT * i, row = id / C::CPR, ch = id % C::CPR;
      u32x4 z = {0u, 0u, 0u, 0u};
      r[i] = (id < TOTAL && row0 + row < nrows)
                 ? *reinterpret_cast<const u32x4*>(base + (int64_t)(row0 + row) * rs + ch * C::VEC) : z;
    }
  }
  FK_DEV void store(char* img, int stride, int tid) const {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int id = tid + NT * i, row = id / C::CPR, ch = id % C::CPR;
      if (id < TOTAL) *reinterpret_cast<u32x4*>(img + row * stride + ch * 16) = r[i];
    }
  }
  FK_DEV void store_img(char* img, int tid) const {   // dual-use layout (Img<T, D>)
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int id = tid + NT * i, row = id / C::CPR, ch = id % C::CPR;
      if (id < TOTAL) *reinterpret_cast<u32x4*>(img + Img<T, D>::off(row, ch * 16)) = r[i];
    }
  }
};

template <int N> FK_DEV void zero_acc(f32x16 (&a)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) a[i][r] = 0.0f;
}

// store a [d x q] accumulator set (lane = q row, registers = d) as rows of a [.., H, D] matrix.
// bf16: a lane holds 4 consecutive d (8 bytes) per register group g, its partner lane (+32) the next 4.  Stored as they stand, every row
// received 16-byte pieces, two lanes per 32-byte sector: the forward wrote 352 MB for a 156 MB output (rocprofv3 WRITE_SIZE, round 2) and
// the store tail is issue-bound (MI355X_MICROARCH.md: "row-per-lane dwordx2 store tail").  v_permlane32_swap pairs the groups instead:
// after swapping (g even, g odd) between the two half-waves, lane li owns d = 8g .. 8g+7 of g = 2 gp and lane li + 32 those of g = 2 gp + 1,
// so one global_store_dwordx4 per lane writes a whole 32-byte sector per row (half the store instructions, no partial sectors).
FK_DEV void swap_halves(unsigned& x, unsigned& y) {      // x[lanes 32-63] <-> y[lanes 0-31]
  typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
  const u32x2_t r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
  x = r[0];
  y = r[1];
}
FK_DEV unsigned pack_bf16x2(float lo, float hi) {
  bf16x2 v = {(bf16_t)lo, (bf16_t)hi};
  return __builtin_bit_cast(unsigned, v);
}

#ifndef FK_KEYPAD_NO_FREE
#define FK_KEYPAD_NO_FREE 0      // 1: a key-padding mask sends every tile down the per-element path (the form before round 4; A/B switch)
#endif
#ifndef FK_NT_STORES_ATTN
#define FK_NT_STORES_ATTN 0      // 1: O / dQ / dK / dV rows stored non-temporal (written once; the launch keeps re-reading K / V or Q / dO out of L2)
#endif
template <typename T, int D>
FK_DEV void store_rows_T(T* base, int64_t rs, int row, bool row_ok, const f32x16 (&acc)[AT<T, D>::DT], float mul, int lh) {
  if constexpr (sizeof(T) == 2 && D % 16 == 0) {
    // every lane takes part in the swaps (EXEC must be whole for them); rows past the end only skip the store
    T* rowp = base + (int64_t)row * rs;
#pragma unroll
    for (int dt = 0; dt < AT<T, D>::DT; ++dt)
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {
        if (dt * 32 + 16 * gp < D) {
          const int g = 2 * gp;
          unsigned x0 = pack_bf16x2(acc[dt][4 * g] * mul, acc[dt][4 * g + 1] * mul), x1 = pack_bf16x2(acc[dt][4 * g + 2] * mul, acc[dt][4 * g + 3] * mul);
          unsigned y0 = pack_bf16x2(acc[dt][4 * g + 4] * mul, acc[dt][4 * g + 5] * mul), y1 = pack_bf16x2(acc[dt][4 * g + 6] * mul, acc[dt][4 * g + 7] * mul);
          swap_halves(x0, y0);
          swap_halves(x1, y1);
          if (row_ok) fk_st<FK_NT_STORES_ATTN != 0>(reinterpret_cast<u32x4*>(rowp + dt * 32 + 8 * (g + lh)), u32x4{x0, x1, y0, y1});
        }
      }
    return;
  }
  if (!row_ok) return;
#pragma unroll
  for (int dt = 0; dt < AT<T, D>::DT; ++dt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int d = dt * 32 + 8 * g + 4 * lh;
      if (d < D) {
        T* dst = base + (int64_t)row * rs + d;
        if constexpr (sizeof(T) == 2) {
          bf16x4 v;
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = (bf16_t)(acc[dt][4 * g + j] * mul);
          *reinterpret_cast<bf16x4*>(dst) = v;
        } else {
          f32x4 v;
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = acc[dt][4 * g + j] * mul;
          *reinterpret_cast<f32x4*>(dst) = v;
        }
      }
    }
}

// dQ / dK store with the inverse RoPE: pairs (d, d+1) of row `row` (sequence position rope_off + row) rotated by -angle
template <typename T, int D>
FK_DEV void store_rows_T_rope(T* base, int64_t rs, int row, bool row_ok, const f32x16 (&acc)[AT<T, D>::DT], float mul, int lh,
                              const float* table) {   // table -> (cos, sin) pairs of this row: [D/2][2]
  auto rot = [&](int dt, int g, float (&o)[4]) {
    const int d = dt * 32 + 8 * g + 4 * lh;
    f32x4 cs = {1.0f, 0.0f, 1.0f, 0.0f};
    if (row_ok) cs = *reinterpret_cast<const f32x4*>(table + d);   // (c0, s0, c1, s1) for pairs d/2, d/2+1
#ifdef FK_ROPE_EPI_NOPS
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15" ::: "memory");
#endif
    const float a0 = acc[dt][4 * g] * mul, a1 = acc[dt][4 * g + 1] * mul, a2 = acc[dt][4 * g + 2] * mul, a3 = acc[dt][4 * g + 3] * mul;
    o[0] = a0 * cs[0] + a1 * cs[1];
    o[1] = -a0 * cs[1] + a1 * cs[0];
    o[2] = a2 * cs[2] + a3 * cs[3];
    o[3] = -a2 * cs[3] + a3 * cs[2];
  };
  if constexpr (sizeof(T) == 2 && D % 16 == 0) {
    T* rowp = base + (int64_t)row * rs;
#pragma unroll
    for (int dt = 0; dt < AT<T, D>::DT; ++dt)
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {
        if (dt * 32 + 16 * gp < D) {
          float ox[4], oy[4];
          rot(dt, 2 * gp, ox);
          rot(dt, 2 * gp + 1, oy);
          unsigned x0 = pack_bf16x2(ox[0], ox[1]), x1 = pack_bf16x2(ox[2], ox[3]), y0 = pack_bf16x2(oy[0], oy[1]), y1 = pack_bf16x2(oy[2], oy[3]);
          swap_halves(x0, y0);
          swap_halves(x1, y1);
          if (row_ok) fk_st<FK_NT_STORES_ATTN != 0>(reinterpret_cast<u32x4*>(rowp + dt * 32 + 8 * (2 * gp + lh)), u32x4{x0, x1, y0, y1});
        }
      }
    return;
  }
  if (!row_ok) return;
#pragma unroll
  for (int dt = 0; dt < AT<T, D>::DT; ++dt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int d = dt * 32 + 8 * g + 4 * lh;
      if (d < D) {
        float o[4];
        rot(dt, g, o);
        T* dst = base + (int64_t)row * rs + d;
        if constexpr (sizeof(T) == 2) {
          bf16x4 v = {(bf16_t)o[0], (bf16_t)o[1], (bf16_t)o[2], (bf16_t)o[3]};
          *reinterpret_cast<bf16x4*>(dst) = v;
        } else {
          *reinterpret_cast<f32x4*>(dst) = f32x4{o[0], o[1], o[2], o[3]};
        }
      }
    }
}

// ================================================================================================= forward
// NW waves per workgroup (32 query rows each).  The DMA path runs 8 waves = 256 query rows per workgroup: every K/V tile
// fetched from L2 then feeds twice the MFMA work (the 128-row version sat on the L2->LDS bandwidth ceiling).
template <typename T, int D, int NW>
__global__ __launch_bounds__(NW * 64) void attn_fwd_kernel(AttnArgs p) {
  using C = AT<T, D>;
  constexpr int BQ = NW * 32, NT = NW * 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool DMA = Img<T, D>::SWZ;   // bf16, D = 64: swizzled 128-B-row images filled by LDS-DMA
  constexpr int KIMG = DMA ? BKV * 128 : BKV * C::RSTRIDE, VIMG = DMA ? BKV * 128 : BKV * C::VSTRIDE;
  constexpr int NSLOT = DMA ? 3 : 2;      // DMA path: 3-slot ring, tiles t+1 and t+2 in flight while tile t is consumed
  auto kimg = [&](int i) -> char* { return smem + i * KIMG; };
  auto vimg = [&](int i) -> char* { return smem + NSLOT * KIMG + i * VIMG; };
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform (SGPR): tile predicates become scalar branches
  // 1-D grid, XCD-aware: each XCD walks whole (batch, head) pairs (their K/V stay in its L2), heaviest
  // (latest) query block first.
  const int nqb = (p.Nq + BQ - 1) / BQ;
  const unsigned L = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = (int)(L / nqb), b = bh / p.H, hd = bh % p.H, q0 = (nqb - 1 - (int)(L % nqb)) * BQ;
  const T* Qp = (const T*)p.Q + (int64_t)b * p.q_bs + hd * D;
  const T* Kp = (const T*)p.K + (int64_t)b * p.k_bs + hd * D;
  const T* Vp = (const T*)p.V + (int64_t)b * p.v_bs + hd * D;
  const int qrow = q0 + wave * 32 + li;
  const bool q_ok = qrow < p.Nq;
  const bool prefix = p.mask_kind == FK_MASK_PREFIX, keypad = p.mask_kind == FK_MASK_KEYPAD;
  const bool dense = p.mask_kind == FK_MASK_DENSE;
  const int my_lim = ((prefix || keypad) && q_ok) ? p.limits[(int64_t)b * p.Nq + qrow] : 0;   // prefix length / query validity

  if constexpr (C::DPAD != D) {   // zero the padded columns of the K/V images once (never restaged)
    for (int i = tid; i < (NSLOT * KIMG + NSLOT * VIMG) / 4; i += NT) reinterpret_cast<float*>(smem)[i] = 0.0f;
    __syncthreads();
  }

  Frag<T> qf[C::KSTEPS];
#pragma unroll
  for (int s = 0; s < C::KSTEPS; ++s) {
    if (q_ok && 16 * s + 8 * lh < D) frag_load_contig<T>(qf[s], Qp + (int64_t)qrow * p.q_rs + 16 * s + 8 * lh);
    else frag_zero<T>(qf[s]);
  }

  const int q_last = min(q0 + BQ, p.Nq) - 1;
  const int kv_end = (p.mask_kind == FK_MASK_KEYPAD || p.mask_kind == FK_MASK_DENSE) ? p.Nk : kv_limit(p, b, q_last);
  const int ntiles = (kv_end + BKV - 1) / BKV;
  // first key index that is NOT visible to every query row of this wave (tiles below need no mask test)
  const int wave_q_first = min(q0 + wave * 32, p.Nq - 1);
  const int full_vis_end = kv_limit(p, b, wave_q_first);

  Stager<T, D, DMA ? 4 : BKV> sk, sv;
  if constexpr (DMA) {
    // the register-resident Q fragments must be complete before the ring starts: from here on vmcnt counts only
    // LDS-DMA instructions (4 per wave and tile), which the loop retires with counted waits.
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
#pragma unroll
    for (int pt = 0; pt < 2; ++pt)
      if (pt < ntiles) {
        dma_tile_bf16_d64<BKV, NW>((const bf16_t*)Kp, p.k_rs, pt * BKV, p.Nk, kimg(pt), wave, lane);
        dma_tile_bf16_d64<BKV, NW>((const bf16_t*)Vp, p.v_rs, pt * BKV, p.Nk, vimg(pt), wave, lane);
      }
  } else {
    if (ntiles > 0) {
      sk.load(Kp, p.k_rs, 0, p.Nk, tid);
      sv.load(Vp, p.v_rs, 0, p.Nk, tid);
      sk.store(kimg(0), C::RSTRIDE, tid);
      sv.store(vimg(0), C::VSTRIDE, tid);
    }
    __syncthreads();
    // Everything loaded so far (the register-resident fragments) is complete before the loop: tells hipcc's waitcnt
    // pass that the MFMA operands are ready, so inside the loop it waits only for LDS reads and the prefetched tile's
    // global loads stay in flight behind the MFMAs (otherwise it re-waits vmcnt at the first MFMA of every iteration).
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
  }
  const float c = p.scale * LOG2E;
  float m = -INFINITY, l = 0.0f;
  f32x16 o[C::DT];
  zero_acc(o);
  DropKey dkey{};
  unsigned drow = 0;
  if (p.drop_thresh) {
    dkey = drop_key(p.drop_seed, p.drop_site, p.drop_thresh);
    drow = drop_row(dkey, (unsigned)((b * p.H + hd) * p.Nq + qrow));
  }

  int slot = 0;
  for (int t = 0; t < ntiles; ++t) {
    const int kb = t * BKV;
    if constexpr (DMA) {
      // retire tile t's DMA (oldest 4 of this wave) but leave tile t+1's in flight across the barrier; the barrier also
      // guarantees every wave is done reading slot (t+2)%3 (= tile t-1), which is refilled right after it.
      if (t + 1 < ntiles) {
        if constexpr (NW == 8) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      }
      if (t + 1 >= ntiles) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (t + 2 < ntiles) {
        const int ns = slot >= 1 ? slot - 1 : 2;     // (t + 2) % 3
        dma_tile_bf16_d64<BKV, NW>((const bf16_t*)Kp, p.k_rs, kb + 2 * BKV, p.Nk, kimg(ns), wave, lane);
        dma_tile_bf16_d64<BKV, NW>((const bf16_t*)Vp, p.v_rs, kb + 2 * BKV, p.Nk, vimg(ns), wave, lane);
      }
    } else {
      if (t + 1 < ntiles) {
        sk.load(Kp, p.k_rs, kb + BKV, p.Nk, tid);
        sv.load(Vp, p.v_rs, kb + BKV, p.Nk, tid);
      }
    }
    const char* kt = kimg(DMA ? slot : (t & 1));
    const char* vt = vimg(DMA ? slot : (t & 1));
    f32x16 sc[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[u][r] = 0.0f;
#pragma unroll
      for (int s = 0; s < C::KSTEPS; ++s) {
        Frag<T> kf;
        if constexpr (DMA) img_row<T, D>(kf, kt, 32 * u + li, s, lh);
        else frag_row<T>(kf, kt, C::RSTRIDE, 32 * u + li, s, lh);
        mma32<T>(sc[u], kf, qf[s]);
      }
    }
    if (kb + BKV > full_vis_end) {   // wave-uniform: boundary tile -> per-element predicate
      const int qpos = qrow + p.q_off;
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kb + 32 * u + acc_row(r, lh);
          if (!(key < p.Nk && (prefix ? key < my_lim : (keypad ? (my_lim != 0 && p.qfirst[(int64_t)b * p.Nk + key] != 0) : (dense ? (q_ok && dense_vis(p, b, hd, qrow, key)) : visible(p.mask_kind, p.mask_c, qpos, key + p.k_off)))))) sc[u][r] = -INFINITY;
        }
    }
    float tmax = -INFINITY;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, sc[u][r]);
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    // exact deferred rescale: when no row's running max grows in this tile (the common case after the first
    // tiles) alpha == 1 for every lane and the O / l rescale is skipped entirely (wave-uniform branch).
    if (__builtin_amdgcn_ballot_w64(tmax > m) != 0) {
      const float m_new = fmaxf(m, tmax);
      const float alpha = (m_new == -INFINITY) ? 1.0f : __builtin_amdgcn_exp2f((m - m_new) * c);
      m = m_new;
      l *= alpha;
#pragma unroll
      for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
    }
    const float mc = (m == -INFINITY) ? 0.0f : m * c;
    float rs = 0.0f;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = __builtin_amdgcn_exp2f(sc[u][r] * c - mc);
        sc[u][r] = pv;
        rs += pv;
      }
    l += rs;
    if (p.drop_thresh) {             // wave-uniform: the row sum above is the softmax's (undropped); O takes the kept entries only
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (!drop_keep(dkey, drow, (unsigned)(kb + 32 * u + acc_row(r, lh)))) sc[u][r] = 0.0f;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        Frag<T> pf;
        frag_from_acc<T>(pf, sc[u], s);
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) {
          Frag<T> vf;
          if constexpr (DMA) img_tr<T, D>(vf, vt, 32 * u, s, 32 * dt, lane);
          else frag_tr<T>(vf, vt, C::VSTRIDE, 32 * u, s, 32 * dt, lane);
          mma32<T>(o[dt], vf, pf);
        }
      }
    if constexpr (!DMA) {
      if (t + 1 < ntiles) {
        sk.store(kimg((t + 1) & 1), C::RSTRIDE, tid);
        sv.store(vimg((t + 1) & 1), C::VSTRIDE, tid);
      }
      __syncthreads();
    } else {
      slot = slot == 2 ? 0 : slot + 1;
    }
  }

  const float lt = l + __shfl_xor(l, 32, 64);
  const float inv = lt > 0.0f ? (p.drop_thresh ? p.drop_scale : 1.0f) / lt : 0.0f;   // fully masked row -> 0 (torch >= 2.1 CPU semantics)
  T* Op = (T*)p.Out + (int64_t)b * p.o_bs + hd * D;
  store_rows_T<T, D>(Op, p.o_rs, qrow, q_ok, o, inv, lh);
  if (q_ok && lh == 0 && p.LSE)
    p.LSE[((int64_t)b * p.H + hd) * p.Nq + qrow] = lt > 0.0f ? m * p.scale + logf(lt) : INFINITY;
}

// ================================================================================================= few queries, long context
// The perceiver's read-out (models/brainformer.py:204-215): 32 query tokens against 6144 keys per (batch, head).  The kernels above give
// such a launch one workgroup per (b, h) with ONE wave that has rows: 192 waves for the chip, each walking 6144 keys behind a barrier per
// tile (138 us where streaming K and V once takes 60).  Here the eight waves of the workgroup split the KEYS: wave w takes the 32-key
// tiles w, w + 8, ... through its OWN double-buffered LDS-DMA pipeline (8 KiB per tile, no barrier in the loop, counted vmcnt waits),
// keeps its own online-softmax state, and the eight partial results are merged once through LDS (the exact combination rule of
// fk_attn_combine).  bf16, D = 64, Nq <= 32, no mask, no dropout; everything else stays on attn_fwd_kernel.
constexpr int FEWQ_NW = 8, FEWQ_BK = 32, FEWQ_TILE = 2 * FEWQ_BK * 128, FEWQ_LDS = FEWQ_NW * 2 * FEWQ_TILE;
static_assert(FEWQ_LDS <= 160 * 1024 && FEWQ_NW * (32 * 64 * 4 + 2 * 64 * 4) <= FEWQ_LDS, "LDS of a CU; the merge area fits the tile buffers");

__global__ __launch_bounds__(FEWQ_NW * 64) void attn_fwd_fewq_kernel(AttnArgs p) {
  using T = bf16_t;
  constexpr int D = 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bh = blockIdx.x, b = bh / p.H, hd = bh % p.H;
  const T* Qp = (const T*)p.Q + (int64_t)b * p.q_bs + hd * D;
  const T* Kp = (const T*)p.K + (int64_t)b * p.k_bs + hd * D;
  const T* Vp = (const T*)p.V + (int64_t)b * p.v_bs + hd * D;
  const int qrow = li;
  const bool q_ok = qrow < p.Nq;
  Frag<T> qf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    if (q_ok) frag_load_contig<T>(qf[s], Qp + (int64_t)qrow * p.q_rs + 16 * s + 8 * lh);
    else frag_zero<T>(qf[s]);
  }
  char* buf = smem + wave * 2 * FEWQ_TILE;
  const int ntiles = (p.Nk + FEWQ_BK - 1) / FEWQ_BK;
  const int cnt = wave < ntiles ? (ntiles - wave + FEWQ_NW - 1) / FEWQ_NW : 0;       // this wave's tiles: wave, wave + 8, ...
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the Q fragments; from here on vmcnt counts this wave's tile requests only
  auto request = [&](int i) __attribute__((always_inline)) {
    char* img = buf + (i & 1) * FEWQ_TILE;
    const int row0 = (wave + FEWQ_NW * i) * FEWQ_BK;
    dma_tile_bf16_d64<FEWQ_BK, 1>(Kp, p.k_rs, row0, p.Nk, img, 0, lane);
    dma_tile_bf16_d64<FEWQ_BK, 1>(Vp, p.v_rs, row0, p.Nk, img + FEWQ_BK * 128, 0, lane);
  };
  if (cnt > 0) request(0);
  const float c = p.scale * LOG2E;
  float m = -INFINITY, l = 0.0f;
  f32x16 o[2];
  zero_acc(o);
  for (int i = 0; i < cnt; ++i) {
    if (i + 1 < cnt) {
      request(i + 1);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // the 8 requests of tile i have landed, those of tile i + 1 stay in flight
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const char* kt = buf + (i & 1) * FEWQ_TILE;
    const char* vt = kt + FEWQ_BK * 128;
    const int kb = (wave + FEWQ_NW * i) * FEWQ_BK;
    f32x16 sc;
#pragma unroll
    for (int r = 0; r < 16; ++r) sc[r] = 0.0f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      Frag<T> kf;
      img_row<T, D>(kf, kt, li, s, lh);
      mma32<T>(sc, kf, qf[s]);
    }
    if (kb + FEWQ_BK > p.Nk) {            // the ragged last tile (wave-uniform)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (kb + acc_row(r, lh) >= p.Nk) sc[r] = -INFINITY;
    }
    float tmax = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, sc[r]);
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    if (__builtin_amdgcn_ballot_w64(tmax > m) != 0) {
      const float m_new = fmaxf(m, tmax);
      const float alpha = (m_new == -INFINITY) ? 1.0f : __builtin_amdgcn_exp2f((m - m_new) * c);
      m = m_new;
      l *= alpha;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
    }
    const float mc = (m == -INFINITY) ? 0.0f : m * c;
    float rs = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float pv = __builtin_amdgcn_exp2f(sc[r] * c - mc);
      sc[r] = pv;
      rs += pv;
    }
    l += rs;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      Frag<T> pf;
      frag_from_acc<T>(pf, sc, s);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        Frag<T> vf;
        img_tr<T, D>(vf, vt, 0, s, 32 * dt, lane);
        mma32<T>(o[dt], vf, pf);
      }
    }
  }
  // merge: every wave leaves (m, l, O) in the (now idle) tile buffers, wave 0 combines them in wave order (a fixed order: deterministic)
  __syncthreads();
  float* area = reinterpret_cast<float*>(smem);
  constexpr int PER = 32 * 64 + 2 * 64;                    // floats per wave: 32 accumulator registers x 64 lanes, m, l
  {
    float* mine = area + wave * PER;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) mine[(dt * 16 + r) * 64 + lane] = o[dt][r];
    mine[32 * 64 + lane] = m;
    mine[32 * 64 + 64 + lane] = l;
  }
  __syncthreads();
  if (wave != 0) return;
  float M = -INFINITY;
#pragma unroll
  for (int w = 0; w < FEWQ_NW; ++w) M = fmaxf(M, area[w * PER + 32 * 64 + lane]);
  float L = 0.0f;
  zero_acc(o);
#pragma unroll 1
  for (int w = 0; w < FEWQ_NW; ++w) {
    const float* theirs = area + w * PER;
    const float mw = theirs[32 * 64 + lane];
    const float aw = (mw == -INFINITY) ? 0.0f : __builtin_amdgcn_exp2f((mw - M) * c);
    L += aw * theirs[32 * 64 + 64 + lane];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[dt][r] += aw * theirs[(dt * 16 + r) * 64 + lane];
  }
  const float lt = L + __shfl_xor(L, 32, 64);
  const float inv = lt > 0.0f ? 1.0f / lt : 0.0f;
  T* Op = (T*)p.Out + (int64_t)b * p.o_bs + hd * D;
  store_rows_T<T, D>(Op, p.o_rs, qrow, q_ok, o, inv, lh);
  if (q_ok && lh == 0 && p.LSE) p.LSE[((int64_t)b * p.H + hd) * p.Nq + qrow] = lt > 0.0f ? M * p.scale + logf(lt) : INFINITY;
}

// the query gradient of the same shape: wave-private key tiles as above; S = K Q^T and dP = V dO^T per tile, dS = P (dP - delta),
// dQ^T += K^T dS; the eight partial dQ are summed through LDS in wave order.  Publishes delta for the dK/dV kernel like attn_bwd_dq_kernel.
__global__ __launch_bounds__(FEWQ_NW * 64) void attn_bwd_dq_fewq_kernel(AttnArgs p) {
  using T = bf16_t;
  constexpr int D = 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bh = blockIdx.x, b = bh / p.H, hd = bh % p.H;
  const T* Qp = (const T*)p.Q + (int64_t)b * p.q_bs + hd * D;
  const T* Kp = (const T*)p.K + (int64_t)b * p.k_bs + hd * D;
  const T* Vp = (const T*)p.V + (int64_t)b * p.v_bs + hd * D;
  const T* Gp = (const T*)p.dO + (int64_t)b * p.o_bs + hd * D;
  const T* Op = (const T*)p.O + (int64_t)b * p.o_bs + hd * D;
  const int qrow = li;
  const bool q_ok = qrow < p.Nq;
  Frag<T> qf[4], gf[4];
  float part = 0.0f;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    if (q_ok) {
      frag_load_contig<T>(qf[s], Qp + (int64_t)qrow * p.q_rs + 16 * s + 8 * lh);
      frag_load_contig<T>(gf[s], Gp + (int64_t)qrow * p.o_rs + 16 * s + 8 * lh);
      Frag<T> of;
      frag_load_contig<T>(of, Op + (int64_t)qrow * p.o_rs + 16 * s + 8 * lh);
#pragma unroll
      for (int e = 0; e < 8; ++e) part += to_f32<T>(of.v[e]) * to_f32<T>(gf[s].v[e]);
    } else {
      frag_zero<T>(qf[s]);
      frag_zero<T>(gf[s]);
    }
  }
  const int64_t stat = ((int64_t)b * p.H + hd) * p.Nq + qrow;
  const float lse2 = q_ok ? p.LSE[stat] * LOG2E : INFINITY;   // +inf -> P = 0 for padded rows
  const float dl = part + __shfl_xor(part, 32, 64);           // delta = rowsum(dO * O)
  if (wave == 0 && q_ok && lh == 0) p.delta[stat] = dl;
  char* buf = smem + wave * 2 * FEWQ_TILE;
  const int ntiles = (p.Nk + FEWQ_BK - 1) / FEWQ_BK;
  const int cnt = wave < ntiles ? (ntiles - wave + FEWQ_NW - 1) / FEWQ_NW : 0;
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the row fragments and statistics; from here on vmcnt counts this wave's tile requests only
  auto request = [&](int i) __attribute__((always_inline)) {
    char* img = buf + (i & 1) * FEWQ_TILE;
    const int row0 = (wave + FEWQ_NW * i) * FEWQ_BK;
    dma_tile_bf16_d64<FEWQ_BK, 1>(Kp, p.k_rs, row0, p.Nk, img, 0, lane);
    dma_tile_bf16_d64<FEWQ_BK, 1>(Vp, p.v_rs, row0, p.Nk, img + FEWQ_BK * 128, 0, lane);
  };
  if (cnt > 0) request(0);
  const float c = p.scale * LOG2E;
  f32x16 dq[2];
  zero_acc(dq);
  for (int i = 0; i < cnt; ++i) {
    if (i + 1 < cnt) {
      request(i + 1);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const char* kt = buf + (i & 1) * FEWQ_TILE;
    const char* vt = kt + FEWQ_BK * 128;
    const int kb = (wave + FEWQ_NW * i) * FEWQ_BK;
    f32x16 sc, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { sc[r] = 0.0f; dp[r] = 0.0f; }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      Frag<T> kf, vf;
      img_row<T, D>(kf, kt, li, s, lh);
      img_row<T, D>(vf, vt, li, s, lh);
      mma32<T>(sc, kf, qf[s]);
      mma32<T>(dp, vf, gf[s]);
    }
    const bool ragged = kb + FEWQ_BK > p.Nk;                 // wave-uniform
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float pv = __builtin_amdgcn_exp2f(sc[r] * c - lse2);
      if (ragged && kb + acc_row(r, lh) >= p.Nk) pv = 0.0f;
      sc[r] = pv * (dp[r] - dl);                            // dS^T (the softmax scale is folded into the final store)
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      Frag<T> df;
      frag_from_acc<T>(df, sc, s);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        Frag<T> ktf;
        img_tr<T, D>(ktf, kt, 0, s, 32 * dt, lane);
        mma32<T>(dq[dt], ktf, df);
      }
    }
  }
  __syncthreads();
  float* area = reinterpret_cast<float*>(smem);
  {
    float* mine = area + wave * 32 * 64;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) mine[(dt * 16 + r) * 64 + lane] = dq[dt][r];
  }
  __syncthreads();
  if (wave != 0) return;
#pragma unroll 1
  for (int w = 1; w < FEWQ_NW; ++w) {
    const float* theirs = area + w * 32 * 64;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) dq[dt][r] += theirs[(dt * 16 + r) * 64 + lane];
  }
  T* dQp = (T*)p.dQ + (int64_t)b * p.q_bs + hd * D;
  store_rows_T<T, D>(dQp, p.q_rs, qrow, q_ok, dq, p.scale, lh);
}

// ================================================================================================= dQ
// Tile hand-over of the backward kernels: on the LDS-DMA path a wave's own requests (and its LDS stores) must have landed BEFORE it joins
// the barrier — every wave reads rows that other waves requested, and a wait placed after the barrier covers only the wave's own.
template <bool DMA> FK_DEV void tile_sync() {
  if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  else __syncthreads();
}
template <typename T, int D>
__global__ __launch_bounds__(NT) void attn_bwd_dq_kernel(AttnArgs p) {
  using C = AT<T, D>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int IMG = BKV * C::RSTRIDE;   // (Img stride <= RSTRIDE)
  auto kimg = [&](int i) -> char* { return smem + i * IMG; };
  auto vimg = [&](int i) -> char* { return smem + (2 + i) * IMG; };
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform (SGPR): tile predicates become scalar branches
  const int nqb = (p.Nq + BQ - 1) / BQ;
  const unsigned L = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = (int)(L / nqb), b = bh / p.H, hd = bh % p.H, q0 = (nqb - 1 - (int)(L % nqb)) * BQ;
  const T* Qp = (const T*)p.Q + (int64_t)b * p.q_bs + hd * D;
  const T* Kp = (const T*)p.K + (int64_t)b * p.k_bs + hd * D;
  const T* Vp = (const T*)p.V + (int64_t)b * p.v_bs + hd * D;
  const T* Gp = (const T*)p.dO + (int64_t)b * p.o_bs + hd * D;
  const int qrow = q0 + wave * 32 + li;
  const bool q_ok = qrow < p.Nq;
  const bool prefix = p.mask_kind == FK_MASK_PREFIX, keypad = p.mask_kind == FK_MASK_KEYPAD;
  const bool dense = p.mask_kind == FK_MASK_DENSE;
  const int my_lim = ((prefix || keypad) && q_ok) ? p.limits[(int64_t)b * p.Nq + qrow] : 0;   // prefix length / query validity

  if constexpr (C::DPAD != D) {   // padded columns feed the transposed K reads: keep them zero
    for (int i = tid; i < 4 * IMG / 4; i += NT) reinterpret_cast<float*>(smem)[i] = 0.0f;
    __syncthreads();
  }

  Frag<T> qf[C::KSTEPS], gf[C::KSTEPS];
#pragma unroll
  for (int s = 0; s < C::KSTEPS; ++s) {
    if (q_ok && 16 * s + 8 * lh < D) {
      frag_load_contig<T>(qf[s], Qp + (int64_t)qrow * p.q_rs + 16 * s + 8 * lh);
      frag_load_contig<T>(gf[s], Gp + (int64_t)qrow * p.o_rs + 16 * s + 8 * lh);
    } else {
      frag_zero<T>(qf[s]);
      frag_zero<T>(gf[s]);
    }
  }
  const int64_t stat = ((int64_t)b * p.H + hd) * p.Nq + qrow;
  const float lse2 = q_ok ? p.LSE[stat] * LOG2E : INFINITY;   // +inf -> P = 0 for padded rows
  // delta = rowsum(dO * O): this kernel already holds the dO row fragments, so it computes delta itself (one extra read of O) and
  // publishes it for the dK/dV kernel, which is launched after this one; no separate delta launch
  float dl = 0.0f;
  {
    const T* Op = (const T*)p.O + (int64_t)b * p.o_bs + hd * D;
    float part = 0.0f;
#pragma unroll
    for (int s = 0; s < C::KSTEPS; ++s) {
      if (q_ok && 16 * s + 8 * lh < D) {
        Frag<T> of;
        frag_load_contig<T>(of, Op + (int64_t)qrow * p.o_rs + 16 * s + 8 * lh);
#pragma unroll
        for (int e = 0; e < 8; ++e) part += to_f32<T>(of.v[e]) * to_f32<T>(gf[s].v[e]);
      }
    }
    dl = part + __shfl_xor(part, 32, 64);
    if (q_ok && lh == 0) p.delta[stat] = dl;
  }

  const int q_last = min(q0 + BQ, p.Nq) - 1;
  const int kv_end = (p.mask_kind == FK_MASK_KEYPAD || p.mask_kind == FK_MASK_DENSE) ? p.Nk : kv_limit(p, b, q_last);
  const int ntiles = (kv_end + BKV - 1) / BKV;
  const int wave_q_first = min(q0 + wave * 32, p.Nq - 1);
  const int full_vis_end = kv_limit(p, b, wave_q_first);

  constexpr bool DMA = Img<T, D>::SWZ;
  Stager<T, D, DMA ? 4 : BKV> sk, sv;
  DmaCursor<BKV> kcur, vcur;
  if (ntiles > 0) {
    if constexpr (DMA) {
      kcur.init((const bf16_t*)Kp, p.k_rs, 0, wave, lane);
      vcur.init((const bf16_t*)Vp, p.v_rs, 0, wave, lane);
      kcur.next((const bf16_t*)Kp, p.k_rs, 0, p.Nk, kimg(0), wave, lane);
      vcur.next((const bf16_t*)Vp, p.v_rs, 0, p.Nk, vimg(0), wave, lane);
    } else {
      sk.load(Kp, p.k_rs, 0, p.Nk, tid);
      sv.load(Vp, p.v_rs, 0, p.Nk, tid);
      sk.store_img(kimg(0), tid);
      sv.store(vimg(0), C::RSTRIDE, tid);
    }
  }
  tile_sync<DMA>();

  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): see attn_fwd_kernel
  const float c = p.scale * LOG2E;
  f32x16 dq[C::DT];
  zero_acc(dq);
  DropKey dkey{};
  unsigned drow = 0;
  if (p.drop_thresh) {
    dkey = drop_key(p.drop_seed, p.drop_site, p.drop_thresh);
    drow = drop_row(dkey, (unsigned)((b * p.H + hd) * p.Nq + qrow));
  }

  // the tile loop is unrolled by the ring depth so that the LDS slot is a compile-time constant: every fragment read then carries its slot
  // offset in the instruction's immediate field instead of a per-read v_add_u32 (VALU issue is what bounds these kernels)
  auto tile_step = [&](auto SL, int t) {
    constexpr int SLOT = decltype(SL)::value;
    const int kb = t * BKV;
    if (t + 1 < ntiles) {
      if constexpr (DMA) {
        kcur.next((const bf16_t*)Kp, p.k_rs, kb + BKV, p.Nk, kimg(SLOT ^ 1), wave, lane);
        vcur.next((const bf16_t*)Vp, p.v_rs, kb + BKV, p.Nk, vimg(SLOT ^ 1), wave, lane);
      } else {
        sk.load(Kp, p.k_rs, kb + BKV, p.Nk, tid);
        sv.load(Vp, p.v_rs, kb + BKV, p.Nk, tid);
      }
    }
    const char* kt = kimg(SLOT);
    const char* vt = vimg(SLOT);
    const bool boundary = kb + BKV > full_vis_end || p.drop_thresh != 0;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      f32x16 sc, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { sc[r] = 0.0f; dp[r] = 0.0f; }
#pragma unroll
      for (int s = 0; s < C::KSTEPS; ++s) {
        Frag<T> kf, vf;
        img_row<T, D>(kf, kt, 32 * u + li, s, lh);
        if constexpr (DMA) img_row<T, D>(vf, vt, 32 * u + li, s, lh);
        else frag_row<T>(vf, vt, C::RSTRIDE, 32 * u + li, s, lh);
        mma32<T>(sc, kf, qf[s]);
        mma32<T>(dp, vf, gf[s]);
      }
      if (!boundary) {
        // packed fp32 pairs (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32): ~1 % on the backward; the same change made the forward slower
        const f32x2 c2 = {c, c}, nl2 = {-lse2, -lse2}, ndl2 = {-dl, -dl};
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          f32x2 x = {sc[r], sc[r + 1]}, d = {dp[r], dp[r + 1]};
          x = __builtin_elementwise_fma(x, c2, nl2);
          x[0] = __builtin_amdgcn_exp2f(x[0]);
          x[1] = __builtin_amdgcn_exp2f(x[1]);
          x *= d + ndl2;
          sc[r] = x[0];
          sc[r + 1] = x[1];
        }
      } else {
        const int qpos = qrow + p.q_off;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float pv = __builtin_amdgcn_exp2f(sc[r] * c - lse2);
          const int key = kb + 32 * u + acc_row(r, lh);
          if (!(key < p.Nk && (prefix ? key < my_lim : (keypad ? (my_lim != 0 && p.qfirst[(int64_t)b * p.Nk + key] != 0) : (dense ? (q_ok && dense_vis(p, b, hd, qrow, key)) : visible(p.mask_kind, p.mask_c, qpos, key + p.k_off)))))) pv = 0.0f;
          float dpv = dp[r];
          if (p.drop_thresh) dpv = drop_keep(dkey, drow, (unsigned)key) ? dpv * p.drop_scale : 0.0f;     // d(P) through the dropout
          sc[r] = pv * (dpv - dl);   // dS^T (without the softmax scale; folded into the final store)
        }
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        Frag<T> df;
        frag_from_acc<T>(df, sc, s);
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) {
          Frag<T> ktf;
          img_tr<T, D>(ktf, kt, 32 * u, s, 32 * dt, lane);
          mma32<T>(dq[dt], ktf, df);
        }
      }
    }
    if constexpr (!DMA) {
      if (t + 1 < ntiles) {
        sk.store_img(kimg(SLOT ^ 1), tid);
        sv.store(vimg(SLOT ^ 1), C::RSTRIDE, tid);
      }
    }
    tile_sync<DMA>();
    };
  for (int t = 0; t < ntiles; t += 2) {
    tile_step(std::integral_constant<int, 0>{}, t);
    if (t + 1 < ntiles) tile_step(std::integral_constant<int, 1>{}, t + 1);
  }
  T* dQp = (T*)p.dQ + (int64_t)b * p.q_bs + hd * D;
  if (p.rope_table)           // wave-uniform: the stores swap registers between the half-waves, every lane takes part
    store_rows_T_rope<T, D>(dQp, p.q_rs, qrow, q_ok, dq, p.scale, lh, p.rope_table + (int64_t)b * p.rope_bs + (int64_t)(p.rope_off + qrow) * D);
  else
    store_rows_T<T, D>(dQp, p.q_rs, qrow, q_ok, dq, p.scale, lh);
}

// ================================================================================================= dK, dV
// workgroup = 128 keys (wave = 32 keys, key on the lane); sweeps query tiles of 64 rows.
template <typename T, int D>
__global__ __launch_bounds__(NT, (sizeof(T) == 2 && D <= 64) ? 2 : 1) void attn_bwd_dkdv_kernel(AttnArgs p) {
  using C = AT<T, D>;
  constexpr int TQ = 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int IMG = TQ * C::RSTRIDE;
  auto qimg = [&](int i) -> char* { return smem + i * IMG; };
  auto gimg = [&](int i) -> char* { return smem + (2 + i) * IMG; };
  float* stats = reinterpret_cast<float*>(smem + 4 * IMG);   // [2 buffers][2 (lse2, delta)][TQ]
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform (SGPR): tile predicates become scalar branches
  const int nkb = (p.Nk + 127) / 128;   // XCD-aware 1-D grid; key block 0 (seen by every query) first
  const unsigned L = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = (int)(L / nkb), b = bh / p.H, hd = bh % p.H, k0 = (int)(L % nkb) * 128;
  const T* Qp = (const T*)p.Q + (int64_t)b * p.q_bs + hd * D;
  const T* Kp = (const T*)p.K + (int64_t)b * p.k_bs + hd * D;
  const T* Vp = (const T*)p.V + (int64_t)b * p.v_bs + hd * D;
  const T* Gp = (const T*)p.dO + (int64_t)b * p.o_bs + hd * D;
  const int krow = k0 + wave * 32 + li;
  const bool k_ok = krow < p.Nk;
  const bool prefix = p.mask_kind == FK_MASK_PREFIX, keypad = p.mask_kind == FK_MASK_KEYPAD;
  const bool dense = p.mask_kind == FK_MASK_DENSE;
  const int my_qf = ((prefix || keypad) && k_ok) ? p.qfirst[(int64_t)b * p.Nk + krow] : 0;     // first query / key validity

  if constexpr (C::DPAD != D) {
    for (int i = tid; i < 4 * IMG / 4; i += NT) reinterpret_cast<float*>(smem)[i] = 0.0f;
    __syncthreads();
  }

  Frag<T> kf[C::KSTEPS], vf[C::KSTEPS];
#pragma unroll
  for (int s = 0; s < C::KSTEPS; ++s) {
    if (k_ok && 16 * s + 8 * lh < D) {
      frag_load_contig<T>(kf[s], Kp + (int64_t)krow * p.k_rs + 16 * s + 8 * lh);
      frag_load_contig<T>(vf[s], Vp + (int64_t)krow * p.v_rs + 16 * s + 8 * lh);
    } else {
      frag_zero<T>(kf[s]);
      frag_zero<T>(vf[s]);
    }
  }

  const int qs = (p.mask_kind == FK_MASK_KEYPAD || p.mask_kind == FK_MASK_DENSE) ? 0 : (q_first(p, b, k0) / TQ) * TQ;   // first query tile that can see key k0
  const int ntiles = qs < p.Nq ? (p.Nq - qs + TQ - 1) / TQ : 0;
  // queries >= this index see every key of this wave's 32 keys
  const int full_vis_q = q_first(p, b, min(k0 + wave * 32 + 31, p.Nk - 1));
  const int64_t stat0 = ((int64_t)b * p.H + hd) * p.Nq;

  constexpr bool DMA = Img<T, D>::SWZ;          // bf16, D = 64: tiles arrive by LDS-DMA, no staging registers
  Stager<T, D, DMA ? 4 : TQ> sq, sg;            // (register staging kept for the other shapes)
  float st_l = 0.0f, st_d = 0.0f;
  bool st_ok = false;
  // raw loads only: any arithmetic on the loaded value here would force an early vmcnt wait and serialise the
  // prefetch of the next tile behind this tile's MFMAs (the scaling happens in store_stats, after the compute)
  auto load_stats = [&](int qb) {
    if (tid < TQ) {
      const int q = qb + tid;
      st_ok = q < p.Nq;
      if (st_ok) {
        st_l = p.LSE[stat0 + q];
        st_d = p.delta[stat0 + q];
      }
    }
  };
  auto store_stats = [&](int buf) {
    if (tid < TQ) {
      stats[buf * 2 * TQ + tid] = st_ok ? -(st_l * LOG2E) : -INFINITY;   // staged NEGATED: phase 2 then needs no per-element sign flips
      stats[buf * 2 * TQ + TQ + tid] = st_ok ? -st_d : 0.0f;
    }
  };
  if (ntiles > 0) {
    if constexpr (DMA) {
      dma_tile_bf16_d64<TQ>((const bf16_t*)Qp, p.q_rs, qs, p.Nq, qimg(0), wave, lane);
      dma_tile_bf16_d64<TQ>((const bf16_t*)Gp, p.o_rs, qs, p.Nq, gimg(0), wave, lane);
    } else {
      sq.load(Qp, p.q_rs, qs, p.Nq, tid);
      sg.load(Gp, p.o_rs, qs, p.Nq, tid);
    }
    load_stats(qs);
    if constexpr (!DMA) {
      sq.store_img(qimg(0), tid);
      sg.store_img(gimg(0), tid);
    }
    store_stats(0);
  }
  tile_sync<DMA>();

  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): see attn_fwd_kernel
  const float c = p.scale * LOG2E;
  f32x16 dk[C::DT], dv[C::DT];
  zero_acc(dk);
  zero_acc(dv);
  DropKey dkey{};
  if (p.drop_thresh) dkey = drop_key(p.drop_seed, p.drop_site, p.drop_thresh);

  auto tile_step = [&](auto SL, int t) {             // unrolled by the ring depth: compile-time LDS slot (see attn_bwd_dq_kernel)
    constexpr int SLOT = decltype(SL)::value;
    const int qb = qs + t * TQ;
    if (t + 1 < ntiles) {
      if constexpr (DMA) {
        dma_tile_bf16_d64<TQ>((const bf16_t*)Qp, p.q_rs, qb + TQ, p.Nq, qimg(SLOT ^ 1), wave, lane);
        dma_tile_bf16_d64<TQ>((const bf16_t*)Gp, p.o_rs, qb + TQ, p.Nq, gimg(SLOT ^ 1), wave, lane);
      } else {
        sq.load(Qp, p.q_rs, qb + TQ, p.Nq, tid);
        sg.load(Gp, p.o_rs, qb + TQ, p.Nq, tid);
      }
      load_stats(qb + TQ);
    }
    const char* qt = qimg(SLOT);
    const char* gt = gimg(SLOT);
    const float* stl = stats + SLOT * 2 * TQ;
    const float* std_ = stl + TQ;
    const bool boundary = (qb < full_vis_q) || (k0 + wave * 32 + 31 >= p.Nk) || p.drop_thresh != 0;   // wave-uniform
    // phase 1: S = Q K^T and dP = dO V^T for both 32-row query sub-tiles (16 MFMAs back to back)
    f32x16 sc[2], dp[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { sc[u][r] = 0.0f; dp[u][r] = 0.0f; }
#pragma unroll
      for (int s = 0; s < C::KSTEPS; ++s) {
        Frag<T> qf, gf;
        img_row<T, D>(qf, qt, 32 * u + li, s, lh);
        img_row<T, D>(gf, gt, 32 * u + li, s, lh);
        mma32<T>(sc[u], qf, kf[s]);   // S[q][key]
        mma32<T>(dp[u], gf, vf[s]);   // dP[q][key]
      }
    }
    // phase 2: P = exp2(c S - lse2), dS = P (dP - delta)   (VALU; overlaps the partner wave's MFMA phases)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (!boundary) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 l4 = *reinterpret_cast<const f32x4*>(stl + 32 * u + 8 * g + 4 * lh);
          const f32x4 d4 = *reinterpret_cast<const f32x4*>(std_ + 32 * u + 8 * g + 4 * lh);
          const f32x2 c2 = {c, c};
#pragma unroll
          for (int j = 0; j < 4; j += 2) {
            const int r = 4 * g + j;
            f32x2 x = {sc[u][r], sc[u][r + 1]}, d = {dp[u][r], dp[u][r + 1]};
            const f32x2 nl = {l4[j], l4[j + 1]}, nd = {d4[j], d4[j + 1]};      // (-lse2, -delta)
            x = __builtin_elementwise_fma(x, c2, nl);
            x[0] = __builtin_amdgcn_exp2f(x[0]);
            x[1] = __builtin_amdgcn_exp2f(x[1]);
            d = x * (d + nd);
            sc[u][r] = x[0];                        // P
            sc[u][r + 1] = x[1];
            dp[u][r] = d[0];                        // dS (scale folded into the final store)
            dp[u][r + 1] = d[1];
          }
        }
      } else {
        const int kpos = krow + p.k_off;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 l4 = *reinterpret_cast<const f32x4*>(stl + 32 * u + 8 * g + 4 * lh);
          const f32x4 d4 = *reinterpret_cast<const f32x4*>(std_ + 32 * u + 8 * g + 4 * lh);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int r = 4 * g + j;
            float pv = __builtin_amdgcn_exp2f(sc[u][r] * c + l4[j]);
            const int q = qb + 32 * u + acc_row(r, lh);
            if (!(k_ok && (prefix ? q >= my_qf : (keypad ? (my_qf != 0 && q < p.Nq && p.limits[(int64_t)b * p.Nq + q] != 0) : (dense ? (q < p.Nq && dense_vis(p, b, hd, q, krow)) : visible(p.mask_kind, p.mask_c, q + p.q_off, kpos)))))) pv = 0.0f;
            float dpv = dp[u][r];
            if (p.drop_thresh) {     // dV takes the dropped, rescaled P; dP comes back through the same mask
              const bool keep = drop_keep(dkey, drop_row(dkey, (unsigned)((b * p.H + hd) * p.Nq + q)), (unsigned)krow);
              dpv = keep ? dpv * p.drop_scale : 0.0f;
              sc[u][r] = keep ? pv * p.drop_scale : 0.0f;
            } else {
              sc[u][r] = pv;
            }
            dp[u][r] = pv * (dpv + d4[j]);
          }
        }
      }
    }
    // phase 3: dV^T += dO^T P, dK^T += Q^T dS (16 MFMAs)
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        Frag<T> pf, df;
        frag_from_acc<T>(pf, sc[u], s);
        frag_from_acc<T>(df, dp[u], s);
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) {
          Frag<T> gtf, qtf;
          img_tr<T, D>(gtf, gt, 32 * u, s, 32 * dt, lane);
          img_tr<T, D>(qtf, qt, 32 * u, s, 32 * dt, lane);
          mma32<T>(dv[dt], gtf, pf);   // dV^T[d][key] += dO^T P
          mma32<T>(dk[dt], qtf, df);   // dK^T[d][key] += Q^T dS
        }
      }
    if (t + 1 < ntiles) {
      if constexpr (!DMA) {
        sq.store_img(qimg(SLOT ^ 1), tid);
        sg.store_img(gimg(SLOT ^ 1), tid);
      }
      store_stats(SLOT ^ 1);
    }
    tile_sync<DMA>();
    };
  for (int t = 0; t < ntiles; t += 2) {
    tile_step(std::integral_constant<int, 0>{}, t);
    if (t + 1 < ntiles) tile_step(std::integral_constant<int, 1>{}, t + 1);
  }
  T* dKp = (T*)p.dK + (int64_t)b * p.k_bs + hd * D;
  T* dVp = (T*)p.dV + (int64_t)b * p.v_bs + hd * D;
  if (p.rope_table)           // wave-uniform: the stores swap registers between the half-waves, every lane takes part
    store_rows_T_rope<T, D>(dKp, p.k_rs, krow, k_ok, dk, p.scale, lh, p.rope_table + (int64_t)b * p.rope_bs + (int64_t)(p.rope_off + krow) * D);
  else
    store_rows_T<T, D>(dKp, p.k_rs, krow, k_ok, dk, p.scale, lh);
  store_rows_T<T, D>(dVp, p.v_rs, krow, k_ok, dv, 1.0f, lh);
}


// ================================================================================================= pre-scaled Q (bf16, D = 64)
// FK_ATTN_Q_PRESCALED: the projection epilogue has already multiplied Q by scale * log2(e) (fk_gemm_nt_rope's second table), so
// S2 = Q' K^T is the score in the exp2 domain.  VALU issue, not the matrix pipe, bounds these kernels (SQ counters: 13 VALU
// instructions per MFMA in the forward, 7.5 in the backward, profiles/r01_pmc_sq_attention.txt), so everything per score that is not
// the exponential itself is moved into the matrix instruction: the row constant of each product is its INITIAL ACCUMULATOR.
//   forward : S' = Q'K^T - m_ref (accumulator preset to -m_ref),  P = exp2(S');  m_ref is a per-row reference, not the running
//             maximum: online softmax is exact for ANY reference, the maximum only keeps exp2 in range.  The hot path therefore
//             computes no maximum at all; a tile whose half-row sum of P exceeds PS_REDO (or is not finite), or a row that has no
//             reference yet, takes the exact path (true tile maximum, rescale of O and l, new reference) — wave-uniform and rare:
//             the first visible tile of a row, and afterwards only if a score outgrows the reference by more than ~40 (log2 units).
//             bf16 has fp32's exponent range, so P up to 2^40 keeps the same relative precision as P <= 1.
//   backward: S' = Q'K^T - LSE*log2(e) and dP' = dO V^T - delta start from the row constants, P = exp2(S'), dS = P * dP'.
#ifdef FK_STAMP
// diagnostic build only (tools/build_variant.sh stamp ... -DFK_STAMP): cycles per loop segment, summed over waves; never in the product
__device__ unsigned long long fk_stamp_acc[16];
#define FK_ST_DECL unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define FK_ST(i) { const unsigned long long st_t1 = __builtin_amdgcn_s_memtime(); st_acc[i] += st_t1 - st_t0; st_t0 = st_t1; }
#define FK_ST_FLUSH(base) if (lane == 0) { for (int i_ = 0; i_ < 8; ++i_) atomicAdd(&fk_stamp_acc[(base) + i_], st_acc[i_]); }
// lifetime of every wave, summed per kernel (slot 8 + k): divided by wave slots x launch duration it gives the shader clock of the launch
#define FK_LIFE_BEGIN const unsigned long long life_t0 = __builtin_amdgcn_s_memtime();
#define FK_LIFE_END(k) if ((threadIdx.x & 63) == 0) atomicAdd(&fk_stamp_acc[8 + (k)], __builtin_amdgcn_s_memtime() - life_t0);
#else
#define FK_ST_DECL
#define FK_ST(i)
#define FK_ST_FLUSH(base)
#define FK_LIFE_BEGIN
#define FK_LIFE_END(k)
#endif
constexpr float PS_REDO = 1.0e12f;
constexpr float LN2 = 0.6931471805599453f;

FK_DEV f32x16 mfma_bf16(const bf16x8& a, const bf16x8& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// every wave's LDS-DMA of the next tile has landed and every wave is done with the current one
FK_DEV void dma_wait_barrier() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int NW>
__global__ __launch_bounds__(NW * 64, 4) void attn_fwd_ps_kernel(AttnArgs p) {
  FK_LIFE_BEGIN
  using T = bf16_t;
  constexpr int D = 64, BQ = NW * 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KIMG = BKV * 128, VIMG = BKV * 128, NSLOT = 3;
  auto kimg = [&](int i) -> char* { return smem + i * KIMG; };
  auto vimg = [&](int i) -> char* { return smem + NSLOT * KIMG + i * VIMG; };
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nqb = (p.Nq + BQ - 1) / BQ;
  const unsigned L = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = (int)(L / nqb), b = bh / p.H, hd = bh % p.H, q0 = (nqb - 1 - (int)(L % nqb)) * BQ;
  const T* Qp = (const T*)p.Q + (int64_t)b * p.q_bs + hd * D;
  const T* Kp = (const T*)p.K + (int64_t)b * p.k_bs + hd * D;
  const T* Vp = (const T*)p.V + (int64_t)b * p.v_bs + hd * D;
  const int qrow = q0 + wave * 32 + li;
  const bool q_ok = qrow < p.Nq;
  const bool prefix = p.mask_kind == FK_MASK_PREFIX, keypad = p.mask_kind == FK_MASK_KEYPAD;
  const int my_lim = ((prefix || keypad) && q_ok) ? p.limits[(int64_t)b * p.Nq + qrow] : 0;
  // key padding: keys in front of the sample's first padded key need no predicate for a wave whose own 32 queries are all valid
  int keypad_free = 0;
  if (keypad && !FK_KEYPAD_NO_FREE) {
    keypad_free = keypad_valid_prefix(p.qfirst + (int64_t)b * p.Nk, p.Nk, lane);
    if (__builtin_amdgcn_ballot_w64(q_ok && my_lim == 0) != 0) keypad_free = 0;
  }

  Frag<T> qf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    if (q_ok) frag_load_contig<T>(qf[s], Qp + (int64_t)qrow * p.q_rs + 16 * s + 8 * lh);
    else frag_zero<T>(qf[s]);
  }
  const int q_last = min(q0 + BQ, p.Nq) - 1;
  const int kv_end = keypad ? p.Nk : kv_limit(p, b, q_last);
  const int ntiles = (kv_end + BKV - 1) / BKV;
  const int wave_q_first = min(q0 + wave * 32, p.Nq - 1);
  const int full_vis_end = keypad ? keypad_free : kv_limit(p, b, wave_q_first);

  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): from here on vmcnt counts only LDS-DMA instructions (see attn_fwd_kernel)
  // K and V tile t into ring slot sl: 16 one-KiB pieces per tile, NW <= 8: 16 / NW per wave; NW = 16: one piece per wave
  auto dma_kv = [&](int t, int sl) {
    if constexpr (NW == 16) {
      if (wave < 8) dma_group_bf16_d64(Kp, p.k_rs, t * BKV, p.Nk, kimg(sl), wave, lane);
      else dma_group_bf16_d64(Vp, p.v_rs, t * BKV, p.Nk, vimg(sl), wave - 8, lane);
    } else {
      dma_tile_bf16_d64<BKV, NW>(Kp, p.k_rs, t * BKV, p.Nk, kimg(sl), wave, lane);
      dma_tile_bf16_d64<BKV, NW>(Vp, p.v_rs, t * BKV, p.Nk, vimg(sl), wave, lane);
    }
  };
#pragma unroll
  for (int pt = 0; pt < 2; ++pt)
    if (pt < ntiles) dma_kv(pt, pt);
  float m = -INFINITY, l = 0.0f;        // m: this row's reference (exp2 domain); -inf = none yet
  f32x16 o[2];
  zero_acc(o);

  // Work proceeds in 32-key sub-tiles j = 2 t + u (nothing couples the two halves of a staged tile once there is no tile-wide maximum);
  // the ring bookkeeping runs when a tile is entered.
  int slot = -1, entered = -1;
  auto enter_tile = [&](int t) {
    if (t == entered) return;
    entered = t;
    slot = slot == 2 ? 0 : slot + 1;                // t % 3
    // retire tile t's DMA (oldest of this wave) but leave tile t+1's in flight across the barrier; the barrier also guarantees every
    // wave is done reading slot (t+2)%3 (= tile t-1), which is refilled right after it.
    if (t + 1 < ntiles) {
      if constexpr (NW == 16) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
      else if constexpr (NW == 8) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
    if (t + 1 >= ntiles) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (t + 2 < ntiles) {
      const int ns = slot >= 1 ? slot - 1 : 2;     // (t + 2) % 3
      dma_kv(t + 2, ns);
    }
  };
  // S2 = Q' K^T of sub-tile j (accumulator starts from literal zero), masked on boundary sub-tiles
  auto scores = [&](int j, f32x16& sc) {
    const char* kt = kimg(slot) + (j & 1) * 4096;   // 32 image rows further on (the chunk swizzle has period 16 rows)
#pragma unroll
    for (int r = 0; r < 16; ++r) sc[r] = 0.0f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      Frag<T> kf;
      img_row<T, D>(kf, kt, li, s, lh);
      mma32<T>(sc, kf, qf[s]);
    }
    if (32 * j + 32 > full_vis_end) {               // wave-uniform: boundary sub-tile -> per-element predicate
      const int qpos = qrow + p.q_off;
      // opaque to the optimiser: otherwise the 16 key indices are shared between the call sites and hoisted out of the branch into the
      // hot paths, where they cost 32 live registers (60 spills at 4 waves per SIMD)
      int kbase = 32 * j + 4 * lh;
      asm volatile("" : "+v"(kbase));
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kbase + (r & 3) + 8 * (r >> 2);
        const bool vis = key < p.Nk && (prefix ? key < my_lim : (keypad ? (my_lim != 0 && p.qfirst[(int64_t)b * p.Nk + key] != 0) : visible(p.mask_kind, p.mask_c, qpos, key + p.k_off)));
        sc[r] = vis ? sc[r] : -INFINITY;
      }
    }
  };
  // O^T += V^T P^T for sub-tile j
  auto pv = [&](int j, const f32x16& sc) {
    const char* vt = vimg(slot) + (j & 1) * 4096;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      Frag<T> pf;
      frag_from_acc<T>(pf, sc, s);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        Frag<T> vf;
        img_tr<T, D>(vf, vt, 0, s, 32 * dt, lane);
        mma32<T>(o[dt], vf, pf);
      }
    }
  };
  // classic online softmax with a running maximum (exact deferred rescale): first sub-tiles of a row, and the fallback
  auto classic = [&](int j) {
    f32x16 sc;
    scores(j, sc);
    float tmax = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, sc[r]);
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    if (__builtin_amdgcn_ballot_w64(tmax > m) != 0) {
      const float m_new = fmaxf(m, tmax);
      const float alpha = (m_new == -INFINITY) ? 1.0f : __builtin_amdgcn_exp2f(m - m_new);
      m = m_new;
      l *= alpha;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
    }
    const float mr = (m == -INFINITY) ? 0.0f : m;
    float rs = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float e = __builtin_amdgcn_exp2f(sc[r] - mr);
      sc[r] = e;
      rs += e;
    }
    l += rs;
    pv(j, sc);
  };
  // reference 0: P = exp2(S2), nothing else per score.  Returns false (sub-tile NOT consumed) when a half-row sum leaves the safe range.
  auto zref = [&](int j) -> bool {
    f32x16 sc;
    scores(j, sc);
    float rs = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float e = __builtin_amdgcn_exp2f(sc[r]);
      sc[r] = e;
      rs += e;
    }
    if (__builtin_amdgcn_ballot_w64(!(rs <= PS_REDO)) != 0) return false;
    l += rs;
    pv(j, sc);
    return true;
  };

  const int J = 2 * ntiles;
  int j = 0;
  bool win = false;
  // 1. classic until every row of the wave has a maximum inside the window in which reference 0 can neither overflow nor lose the row
  for (; j < J && !win; ++j) {
    enter_tile(j >> 1);
    classic(j);
    win = __builtin_amdgcn_ballot_w64(!(m >= -64.0f && m <= 32.0f)) == 0;
  }
  if (win) {
    // 2. re-reference to 0 and run the lean loop
    const float a = __builtin_amdgcn_exp2f(m);
    m = 0.0f;
    l *= a;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[dt][r] *= a;
    for (; j < J; ++j) {
      enter_tile(j >> 1);
      if (!zref(j)) break;
    }
  }
  // 3. whatever is left after a score outgrew the window (reference 0 stays a valid running reference for the classic loop)
  for (; j < J; ++j) {
    enter_tile(j >> 1);
    classic(j);
  }

  const float lt = l + __shfl_xor(l, 32, 64);
  const float inv = lt > 0.0f ? 1.0f / lt : 0.0f;   // fully masked row -> 0 (torch >= 2.1 CPU semantics)
  T* Op = (T*)p.Out + (int64_t)b * p.o_bs + hd * D;
  store_rows_T<T, D>(Op, p.o_rs, qrow, q_ok, o, inv, lh);
  if (q_ok && lh == 0 && p.LSE)
    p.LSE[((int64_t)b * p.H + hd) * p.Nq + qrow] = lt > 0.0f ? m * LN2 + logf(lt) : INFINITY;
  FK_LIFE_END(0)
}

// ------------------------------------------------------------------------------------------------- dQ (pre-scaled Q)
// NW waves per workgroup (32 query rows each): every staged K/V tile costs its 16 LDS-DMA pieces once per workgroup, and an LDS-DMA
// piece costs the issuing wave 60-180 cycles (MI355X_MICROARCH.md, cycle constants), so 8 waves halve that share per MFMA
template <int NW>
__global__ __launch_bounds__(NW * 64, 2) void attn_bwd_dq_ps_kernel(AttnArgs p) {
  using T = bf16_t;
  constexpr int D = 64, BQ = NW * 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // 3-slot ring: tiles t+1 and t+2 in flight while tile t is consumed (with two slots the single tile of prefetch distance left the
  // workgroup waiting at the end-of-tile barrier: 0.25 ms of a 1.8 ms launch in the dK/dV kernel, measured by leaving the DMA out)
  constexpr int IMG = BKV * 128, NS = 3, PCS = 2 * (BKV / 8) / NW;   // PCS: LDS-DMA pieces per wave and tile (K + V)
  auto kimg = [&](int i) -> char* { return smem + i * IMG; };
  auto vimg = [&](int i) -> char* { return smem + (NS + i) * IMG; };
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nqb = (p.Nq + BQ - 1) / BQ;
  const unsigned L = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = (int)(L / nqb), b = bh / p.H, hd = bh % p.H, q0 = (nqb - 1 - (int)(L % nqb)) * BQ;
  const T* Qp = (const T*)p.Q + (int64_t)b * p.q_bs + hd * D;
  const T* Kp = (const T*)p.K + (int64_t)b * p.k_bs + hd * D;
  const T* Vp = (const T*)p.V + (int64_t)b * p.v_bs + hd * D;
  const T* Gp = (const T*)p.dO + (int64_t)b * p.o_bs + hd * D;
  const int qrow = q0 + wave * 32 + li;
  const bool q_ok = qrow < p.Nq;
  const bool prefix = p.mask_kind == FK_MASK_PREFIX, keypad = p.mask_kind == FK_MASK_KEYPAD;
  const int my_lim = ((prefix || keypad) && q_ok) ? p.limits[(int64_t)b * p.Nq + qrow] : 0;

  Frag<T> qf[4], gf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    if (q_ok) {
      frag_load_contig<T>(qf[s], Qp + (int64_t)qrow * p.q_rs + 16 * s + 8 * lh);
      frag_load_contig<T>(gf[s], Gp + (int64_t)qrow * p.o_rs + 16 * s + 8 * lh);
    } else {
      frag_zero<T>(qf[s]);
      frag_zero<T>(gf[s]);
    }
  }
  const int64_t stat = ((int64_t)b * p.H + hd) * p.Nq + qrow;
  const float lse2 = q_ok ? p.LSE[stat] * LOG2E : INFINITY;   // +inf -> P = 0 for padded / fully masked rows
  float dl = 0.0f;                                             // delta = rowsum(dO * O), published for the dK/dV kernel
  {
    const T* Op = (const T*)p.O + (int64_t)b * p.o_bs + hd * D;
    float part = 0.0f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (q_ok) {
        Frag<T> of;
        frag_load_contig<T>(of, Op + (int64_t)qrow * p.o_rs + 16 * s + 8 * lh);
#pragma unroll
        for (int e = 0; e < 8; ++e) part += to_f32<T>(of.v[e]) * to_f32<T>(gf[s].v[e]);
      }
    }
    dl = part + __shfl_xor(part, 32, 64);
    // published for the dK/dV kernel in the form it consumes, as two rows of length Nq rounded up to its 64-row tile:
    //   ws[0][b, h, q] = -LSE * log2(e)   (-inf for rows past Nq and fully masked rows: P = 0),   ws[1][b, h, q] = -delta  (0 past Nq)
    const int nqp = (p.Nq + 63) / 64 * 64;
    if (lh == 0 && qrow < nqp) {
      const int64_t pst = ((int64_t)b * p.H + hd) * nqp + qrow;
      p.delta[pst] = -lse2;
      p.delta[(int64_t)p.B * p.H * nqp + pst] = q_ok ? -dl : 0.0f;
    }
  }
  f32x16 cl, cd;       // the row constants as initial accumulators: S' = Q'K^T - lse2, dP' = dO V^T - delta
#pragma unroll
  for (int r = 0; r < 16; ++r) { cl[r] = -lse2; cd[r] = -dl; }

  const int q_last = min(q0 + BQ, p.Nq) - 1;
  const int kv_end = keypad ? p.Nk : kv_limit(p, b, q_last);
  const int ntiles = (kv_end + BKV - 1) / BKV;
  const int wave_q_first = min(q0 + wave * 32, p.Nq - 1);
  // key padding: tiles in front of the sample's first padded key take the mask-free path; a padded QUERY needs no predicate here, its
  // statistic is +inf (the forward wrote LSE = +inf for a row that sees nothing), so P = exp2(S' - inf) = 0 either way
  const int full_vis_end = keypad ? (FK_KEYPAD_NO_FREE ? 0 : keypad_valid_prefix(p.qfirst + (int64_t)b * p.Nk, p.Nk, lane)) : kv_limit(p, b, wave_q_first);

  DmaCursor<BKV, NW> kcur, vcur;
  static_assert(PCS == 4 || PCS == 2, "vmcnt immediates below");
  if (ntiles > 0) {
    kcur.init(Kp, p.k_rs, 0, wave, lane);
    vcur.init(Vp, p.v_rs, 0, wave, lane);
    kcur.next(Kp, p.k_rs, 0, p.Nk, kimg(0), wave, lane);
    vcur.next(Vp, p.v_rs, 0, p.Nk, vimg(0), wave, lane);
    if (ntiles > 1) {
      kcur.next(Kp, p.k_rs, BKV, p.Nk, kimg(1), wave, lane);
      vcur.next(Vp, p.v_rs, BKV, p.Nk, vimg(1), wave, lane);
    }
  }
  // (the Q / dO / O / LSE register loads above are older than every DMA, so a counted wait covers them too)
  if (ntiles > 1) { if constexpr (PCS == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
  else dma_wait_barrier();
  f32x16 dq[2];
  zero_acc(dq);

  auto tile_step = [&](auto SL, int t) {
    constexpr int SLOT = decltype(SL)::value;
    const int kb = t * BKV;
    if (t + 2 < ntiles) {     // slot (SLOT + 2) % 3 held tile t-1: every wave left it at the barrier that ended the previous step
      kcur.next(Kp, p.k_rs, kb + 2 * BKV, p.Nk, kimg((SLOT + 2) % NS), wave, lane);
      vcur.next(Vp, p.v_rs, kb + 2 * BKV, p.Nk, vimg((SLOT + 2) % NS), wave, lane);
    }
    const char* kt = kimg(SLOT);
    const char* vt = vimg(SLOT);
    const bool boundary = kb + BKV > full_vis_end;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      f32x16 sc, dp;
      {
        Frag<T> kf, vf;
        img_row<T, D>(kf, kt, 32 * u + li, 0, lh);
        img_row<T, D>(vf, vt, 32 * u + li, 0, lh);
        sc = mfma_bf16(kf.v, qf[0].v, cl);
        dp = mfma_bf16(vf.v, gf[0].v, cd);
      }
#pragma unroll
      for (int s = 1; s < 4; ++s) {
        Frag<T> kf, vf;
        img_row<T, D>(kf, kt, 32 * u + li, s, lh);
        img_row<T, D>(vf, vt, 32 * u + li, s, lh);
        mma32<T>(sc, kf, qf[s]);
        mma32<T>(dp, vf, gf[s]);
      }
      if (!boundary) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] = __builtin_amdgcn_exp2f(sc[r]) * dp[r];    // dS^T (softmax scale folded into the final store)
      } else {
        const int qpos = qrow + p.q_off;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float pv = __builtin_amdgcn_exp2f(sc[r]);
          const int key = kb + 32 * u + acc_row(r, lh);
          if (!(key < p.Nk && (prefix ? key < my_lim : (keypad ? (my_lim != 0 && p.qfirst[(int64_t)b * p.Nk + key] != 0) : visible(p.mask_kind, p.mask_c, qpos, key + p.k_off))))) pv = 0.0f;
          sc[r] = pv * dp[r];
        }
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        Frag<T> df;
        frag_from_acc<T>(df, sc, s);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          Frag<T> ktf;
          img_tr<T, D>(ktf, kt, 32 * u, s, 32 * dt, lane);
          mma32<T>(dq[dt], ktf, df);
        }
      }
    }
    // tile t+1 has landed (everything but this step's own pieces), tile t+2 stays in flight across the barrier
    if (t + 2 < ntiles) { if constexpr (PCS == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
    else dma_wait_barrier();
  };
  for (int t = 0; t < ntiles; t += 3) {
    tile_step(std::integral_constant<int, 0>{}, t);
    if (t + 1 < ntiles) tile_step(std::integral_constant<int, 1>{}, t + 1);
    if (t + 2 < ntiles) tile_step(std::integral_constant<int, 2>{}, t + 2);
  }
  T* dQp = (T*)p.dQ + (int64_t)b * p.q_bs + hd * D;
  if (p.rope_table)           // wave-uniform: the stores swap registers between the half-waves, every lane takes part
    store_rows_T_rope<T, D>(dQp, p.q_rs, qrow, q_ok, dq, p.scale, lh, p.rope_table + (int64_t)b * p.rope_bs + (int64_t)(p.rope_off + qrow) * D);
  else
    store_rows_T<T, D>(dQp, p.q_rs, qrow, q_ok, dq, p.scale, lh);
}

#ifndef FK_NO_FWD_ASM
#include "attn_fwd_asm.inc"       // generated by tools/gen/gen_fwd_asm.py: hand-placed, software-pipelined lean forward tile steps
// Forward for shapes where every tile of every wave is fully visible and aligned (launch_fwd checks).  Same algorithm as
// attn_fwd_ps_kernel: classic online softmax until every row's maximum lies in the window, then reference 0 (P = exp2(S'), nothing else
// per score) - here as generated instruction streams, one per 64-key tile, pipelined across tiles (the P.V products of a tile's second
// half run in the next step).  The lean steps never branch: the largest half-row sum is tracked (rmax) and checked once at the end; if
// any wave of the workgroup saw one above PS_REDO (a score far outside the window: precision / overflow risk), the WHOLE workgroup
// redoes its rows with the classic loop.  Ring: 4 slots of 64 keys (K, V images), tile t + 2 requested during step t, one barrier per
// step; every step of every wave, classic or lean, follows the same request / wait / barrier protocol.
__global__ __launch_bounds__(256, 2) void attn_fwd_asm_kernel(AttnArgs p) {
  FK_LIFE_BEGIN
  using T = bf16_t;
  constexpr int D = 64, NW = 4, BQ = NW * 32, IMG = BKV * 128, NS = 4;
  static_assert(BKV == 64, "the generated streams are written for 64-key tiles");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  auto kimg = [&](int i) -> char* { return smem + i * IMG; };
  auto vimg = [&](int i) -> char* { return smem + (NS + i) * IMG; };
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nqb = p.Nq / BQ;
  const unsigned L = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = (int)(L / nqb), b = bh / p.H, hd = bh % p.H, q0 = (nqb - 1 - (int)(L % nqb)) * BQ;
  const T* Qp = (const T*)p.Q + (int64_t)b * p.q_bs + hd * D;
  const T* Kp = (const T*)p.K + (int64_t)b * p.k_bs + hd * D;
  const T* Vp = (const T*)p.V + (int64_t)b * p.v_bs + hd * D;
  const int qrow = q0 + wave * 32 + li;
  const int ntiles = kv_limit(p, b, q0 + BQ - 1) / BKV;       // >= 1; every one of them fully visible to every row of the workgroup

  unsigned vo[4];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = (wave * 2 + j) * 8 + (lane >> 3);
    const int g = (row >> 1) & 7, f = g ^ ((g & 1) << 2);
    vo[j] = (__umul24((unsigned)row, (unsigned)p.k_rs) + (unsigned)(((lane & 7) ^ f) * 8)) * 2u;
    vo[2 + j] = (__umul24((unsigned)row, (unsigned)p.v_rs) + (unsigned)(((lane & 7) ^ f) * 8)) * 2u;
  }
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_void_t*)smem;
  const unsigned ldsw = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)wave * 2048u);
  auto tile_row = [&](int tt) { return (tt < ntiles ? tt : ntiles - 1) * BKV; };             // past the end: the last tile again (never read)
  auto k_base = [&](int tt) { return (uint64_t)(uintptr_t)(Kp + (int64_t)tile_row(tt) * p.k_rs); };
  auto v_base = [&](int tt) { return (uint64_t)(uintptr_t)(Vp + (int64_t)tile_row(tt) * p.v_rs); };
  auto request = [&](int tt) __attribute__((always_inline)) {                                // tile tt -> slot tt % 4
    switch (tt & 3) {
      case 0: fwd_request_asm_slot0(vo, k_base(tt), v_base(tt), ldsw); break;
      case 1: fwd_request_asm_slot1(vo, k_base(tt), v_base(tt), ldsw); break;
      case 2: fwd_request_asm_slot2(vo, k_base(tt), v_base(tt), ldsw); break;
      default: fwd_request_asm_slot3(vo, k_base(tt), v_base(tt), ldsw); break;
    }
  };
  request(0);
  request(1);
  bf16x8 qf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(Qp + (int64_t)qrow * p.q_rs + 16 * s + 8 * lh);
  unsigned aq[4], va0, va1;
#pragma unroll
  for (int s_ = 0; s_ < 4; ++s_) aq[s_] = lds0 + (unsigned)Img<T, D>::off(li, (16 * s_ + 8 * lh) * 2);
  {
    const int g4 = lane >> 4, i16 = lane & 15, hh = g4 >> 1;
    const int rpart = (4 * hh + (i16 >> 2)) * 128 + (i16 & 1) * 8;
    const int c0 = 2 * (g4 & 1) + ((i16 & 3) >> 1), gg0 = 2 * hh + (i16 >> 3), f0 = gg0 ^ ((gg0 & 1) << 2);
    va0 = lds0 + (unsigned)(rpart + ((c0 ^ f0) << 4));
    va1 = lds0 + (unsigned)(rpart + (((c0 ^ f0) ^ 4) << 4));
  }
  // hipcc's wait for the Q fragments goes in front of this use; it is in order and so covers the two tile requests too
  asm volatile("" ::"v"(qf[0]), "v"(qf[1]), "v"(qf[2]), "v"(qf[3]));
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");

  float m = -INFINITY, l = 0.0f, rmax = 0.0f;
  f32x16 o[2], sc1;
  zero_acc(o);
  // one tile with the classic running maximum, as a step of the ring protocol
  auto classic_tile = [&](int t) __attribute__((always_inline)) {
    request(t + 2);
    const char* kt = kimg(t & 3);
    const char* vt = vimg(t & 3);
#pragma unroll 1
    for (int u = 0; u < 2; ++u) {
      f32x16 sc;
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[r] = 0.0f;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        Frag<T> kf, qq;
        img_row<T, D>(kf, kt + u * 4096, li, s, lh);
        qq.v = qf[s];
        mma32<T>(sc, kf, qq);
      }
      float tmax = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, sc[r]);
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
      if (__builtin_amdgcn_ballot_w64(tmax > m) != 0) {
        const float m_new = fmaxf(m, tmax);
        const float alpha = (m_new == -INFINITY) ? 1.0f : __builtin_amdgcn_exp2f(m - m_new);
        m = m_new;
        l *= alpha;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
      }
      const float mr = (m == -INFINITY) ? 0.0f : m;
      float rs = 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = __builtin_amdgcn_exp2f(sc[r] - mr);
        sc[r] = e;
        rs += e;
      }
      l += rs;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        Frag<T> pf;
        frag_from_acc<T>(pf, sc, s);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          Frag<T> vf;
          img_tr<T, D>(vf, vt + u * 4096, 0, s, 32 * dt, lane);
          mma32<T>(o[dt], vf, pf);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };
  // Steps outside the unrolled steady loop (the first lean step, the at most three that align the tile counter to the loop, the tail and
  // the drain) run ONE block per kind for any ring slot: the LDS addresses are moved to the slots here (eight adds per step).  With a
  // switch over four per-slot blocks hipcc moved the carried scores and an accumulator through scratch around every such step
  // (88 bytes per lane, 208 MB written and read per cfg2 call).
#define FK_FWD_STEP(KIND, tt)                                                                                                       \
  {                                                                                                                                 \
    const unsigned so_ = (unsigned)((tt) & 3) * (unsigned)IMG, sp_ = (unsigned)(((tt) + 3) & 3) * (unsigned)IMG;                    \
    const unsigned aqs_[4] = {aq[0] + so_, aq[1] + so_, aq[2] + so_, aq[3] + so_};                                                  \
    fwd_##KIND##_asm_gen(o[0], o[1], sc1, l, rmax, qf, aqs_, va0 + so_, va1 + so_, va0 + sp_, va1 + sp_, vo, k_base((tt) + 2),      \
                         v_base((tt) + 2), ldsw + (unsigned)(((tt) + 2) & 3) * (unsigned)IMG);                                      \
  }

  int* flag = reinterpret_cast<int*>(smem + 2 * NS * IMG);
  bool lean_ok = true;                                         // workgroup-uniform
  for (;;) {
    int t = 0;
    bool win = false;
    // 1. classic until every row of the wave has a maximum inside the window in which reference 0 can neither overflow nor lose the row
    for (; t < ntiles && !(win && lean_ok); ++t) {
      classic_tile(t);
      win = __builtin_amdgcn_ballot_w64(!(m >= -64.0f && m <= 32.0f)) == 0;
    }
    if (t < ntiles) {
      // 2. re-reference to 0 and run the lean steps
      const float a = __builtin_amdgcn_exp2f(m);
      m = 0.0f;
      l *= a;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] *= a;
      FK_FWD_STEP(first, t)                                    // sc1 is an output of the first step (early-clobber, pinned): nothing to initialise
      for (++t; t < ntiles && (t & 3) != 0; ++t) { FK_FWD_STEP(steady, t) }
      // steady state without a switch (hipcc then keeps every operand in one place across the four step variants).  Told here that
      // nothing of ITS memory traffic is pending (spill stores of the classic phase are vector-memory operations): otherwise its wait-count
      // pass puts a vmcnt(0) at the head of the loop below, which drains the streams' tile requests every fourth step.
      __builtin_amdgcn_s_waitcnt(0x0F70);
#define FK_FWD_ARGS(tt) o[0], o[1], sc1, l, rmax, qf, aq, va0, va1, vo, k_base((tt) + 2), v_base((tt) + 2), ldsw
      for (; t + 4 <= ntiles; t += 4) {                        // no branch inside: four steps back to back
        fwd_steady_asm_slot0(FK_FWD_ARGS(t));
        fwd_steady_asm_slot1(FK_FWD_ARGS(t + 1));
        fwd_steady_asm_slot2(FK_FWD_ARGS(t + 2));
        fwd_steady_asm_slot3(FK_FWD_ARGS(t + 3));
      }
      for (; t < ntiles; ++t) { FK_FWD_STEP(steady, t) }
#undef FK_FWD_ARGS
      FK_FWD_STEP(drain, t)                                    // slot argument: the last tile's slot + 1
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");   // stray requests landed, the last MFMAs retired
    if (!lean_ok) break;
#ifdef FK_FWD_PROBE_NOREDO
    break;                                                       // timing builds of ablated streams (tools/stream_variants.sh): their sums mean nothing
#endif
    // 3. did any wave of the workgroup see a half-row sum outside the safe range?  (rare: then everything again, classic only)
    const bool trip = __builtin_amdgcn_ballot_w64(!(rmax <= PS_REDO)) != 0;
    if (tid == 0) *flag = 0;
    __syncthreads();
    if (trip && lane == 0) *flag = 1;
    __syncthreads();
    if (*flag == 0) break;
    lean_ok = false;
    m = -INFINITY; l = 0.0f; rmax = 0.0f;
    zero_acc(o);
    __syncthreads();
    request(0);
    request(1);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  }
#undef FK_FWD_STEP
  const float lt = l + __shfl_xor(l, 32, 64);
  const float inv = lt > 0.0f ? 1.0f / lt : 0.0f;
  T* Op = (T*)p.Out + (int64_t)b * p.o_bs + hd * D;
  store_rows_T<T, D>(Op, p.o_rs, qrow, true, o, inv, lh);
  if (lh == 0 && p.LSE) p.LSE[((int64_t)b * p.H + hd) * p.Nq + qrow] = lt > 0.0f ? m * LN2 + logf(lt) : INFINITY;
  FK_LIFE_END(3)
}
constexpr size_t FWD_ASM_LDS = 2 * 4 * 64 * 128 + 16;
static_assert(FWD_ASM_LDS <= 160 * 1024, "LDS of a CU");
#endif

#ifndef FK_NO_DQ_ASM
#include "attn_dq_asm.inc"        // generated by tools/gen/gen_dq_asm.py: hand-placed instruction stream of one fully visible tile step
// dQ for shapes where every tile of every wave is fully visible and aligned (launch_bwd checks): the tile step is the generated stream,
// which also issues the K / V tile requests of the step after next and ends with the wait + barrier; the C++ around it is the prologue
// (row statistics, delta, published for the dK/dV kernel exactly as attn_bwd_dq_ps_kernel does) and the store.
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_asm_kernel(AttnArgs p) {
  FK_LIFE_BEGIN
  using T = bf16_t;
  constexpr int D = 64, NW = 4, BQ = NW * 32, IMG = BKV * 128, NS = 3;
  static_assert(BKV == 64, "the generated stream is written for 64-key tiles");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nqb = p.Nq / BQ;
  const unsigned L = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = (int)(L / nqb), b = bh / p.H, hd = bh % p.H, q0 = (nqb - 1 - (int)(L % nqb)) * BQ;
  const T* Qp = (const T*)p.Q + (int64_t)b * p.q_bs + hd * D;
  const T* Kp = (const T*)p.K + (int64_t)b * p.k_bs + hd * D;
  const T* Vp = (const T*)p.V + (int64_t)b * p.v_bs + hd * D;
  const T* Gp = (const T*)p.dO + (int64_t)b * p.o_bs + hd * D;
  const int qrow = q0 + wave * 32 + li;
  const int ntiles = kv_limit(p, b, q0 + BQ - 1) / BKV;       // >= 1; every one of them fully visible to every row of the workgroup

  // tile requests: per wave two 8-row groups of the K image and two of the V image; address = 64-bit tile base (SGPR pair) + per-lane
  // byte offset that never changes
  unsigned vo[4];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = (wave * 2 + j) * 8 + (lane >> 3);
    const int g = (row >> 1) & 7, f = g ^ ((g & 1) << 2);
    vo[j] = (__umul24((unsigned)row, (unsigned)p.k_rs) + (unsigned)(((lane & 7) ^ f) * 8)) * 2u;
    vo[2 + j] = (__umul24((unsigned)row, (unsigned)p.v_rs) + (unsigned)(((lane & 7) ^ f) * 8)) * 2u;
  }
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_void_t*)smem;
  const unsigned ldsw = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)wave * 2048u);
  auto tile_row = [&](int tt) { return (tt < ntiles ? tt : ntiles - 1) * BKV; };             // past the end: the last tile again (never read)
  auto k_base = [&](int tt) { return (uint64_t)(uintptr_t)(Kp + (int64_t)tile_row(tt) * p.k_rs); };
  auto v_base = [&](int tt) { return (uint64_t)(uintptr_t)(Vp + (int64_t)tile_row(tt) * p.v_rs); };
  dq_request_asm_slot0(vo, k_base(0), v_base(0), ldsw);
  dq_request_asm_slot1(vo, k_base(1), v_base(1), ldsw);

  bf16x8 qf[4], gf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    qf[s] = *reinterpret_cast<const bf16x8*>(Qp + (int64_t)qrow * p.q_rs + 16 * s + 8 * lh);
    gf[s] = *reinterpret_cast<const bf16x8*>(Gp + (int64_t)qrow * p.o_rs + 16 * s + 8 * lh);
  }
  const int64_t stat = ((int64_t)b * p.H + hd) * p.Nq + qrow;
  const float lse2 = p.LSE[stat] * LOG2E;
  float dl;
  {
    const T* Op = (const T*)p.O + (int64_t)b * p.o_bs + hd * D;
    float part = 0.0f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const bf16x8 of = *reinterpret_cast<const bf16x8*>(Op + (int64_t)qrow * p.o_rs + 16 * s + 8 * lh);
#pragma unroll
      for (int e = 0; e < 8; ++e) part += to_f32<T>(of[e]) * to_f32<T>(gf[s][e]);
    }
    dl = part + __shfl_xor(part, 32, 64);
    if (lh == 0) {                                              // Nq is a multiple of 128 here: the workspace rows are Nq long
      p.delta[stat] = -lse2;
      p.delta[(int64_t)p.B * p.H * p.Nq + stat] = -dl;
    }
  }
  f32x16 cl, cd;       // the row constants as initial accumulators: S' = Q'K^T - lse2, dP' = dO V^T - delta
#pragma unroll
  for (int r = 0; r < 16; ++r) { cl[r] = -lse2; cd[r] = -dl; }
  f32x16 dq[2];
  zero_acc(dq);
  unsigned aq[4], va0, va1;
#pragma unroll
  for (int s_ = 0; s_ < 4; ++s_) aq[s_] = lds0 + (unsigned)Img<T, D>::off(li, (16 * s_ + 8 * lh) * 2);
  {
    const int g4 = lane >> 4, i16 = lane & 15, hh = g4 >> 1;
    const int rpart = (4 * hh + (i16 >> 2)) * 128 + (i16 & 1) * 8;
    const int c0 = 2 * (g4 & 1) + ((i16 & 3) >> 1), gg0 = 2 * hh + (i16 >> 3), f0 = gg0 ^ ((gg0 & 1) << 2);
    va0 = lds0 + (unsigned)(rpart + ((c0 ^ f0) << 4));
    va1 = lds0 + (unsigned)(rpart + (((c0 ^ f0) ^ 4) << 4));
  }
  // hipcc's wait for the register loads above goes in front of this use; it is in order and so covers the two tile requests too
  asm volatile("" ::"v"(qf[0]), "v"(qf[1]), "v"(qf[2]), "v"(qf[3]), "v"(gf[0]), "v"(gf[1]), "v"(gf[2]), "v"(gf[3]), "v"(cl), "v"(cd));
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  for (int t = 0; t < ntiles; t += 3) {
    dq_tile_asm_slot0(dq[0], dq[1], qf, gf, cl, cd, aq, va0, va1, vo, k_base(t + 2), v_base(t + 2), ldsw);
    if (t + 1 < ntiles) dq_tile_asm_slot1(dq[0], dq[1], qf, gf, cl, cd, aq, va0, va1, vo, k_base(t + 3), v_base(t + 3), ldsw);
    if (t + 2 < ntiles) dq_tile_asm_slot2(dq[0], dq[1], qf, gf, cl, cd, aq, va0, va1, vo, k_base(t + 4), v_base(t + 4), ldsw);
  }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");   // stray requests landed, the stream's last MFMAs retired
  T* dQp = (T*)p.dQ + (int64_t)b * p.q_bs + hd * D;
  if (p.rope_table)
    store_rows_T_rope<T, D>(dQp, p.q_rs, qrow, true, dq, p.scale, lh, p.rope_table + (int64_t)b * p.rope_bs + (int64_t)(p.rope_off + qrow) * D);
  else
    store_rows_T<T, D>(dQp, p.q_rs, qrow, true, dq, p.scale, lh);
  FK_LIFE_END(1)
}
#endif

#ifdef FK_DQ16_ASM
#include "attn_dq16_asm.inc"      // generated by tools/gen/gen_dq16_asm.py: the dQ tile step on v_mfma_f32_16x16x32_bf16
// Same contract as attn_bwd_dq_asm_kernel (shapes, workspace rows for the dK/dV kernel), fragments and accumulators in the layouts of the
// 16x16x32 MFMA: lane = (c, g) = (lane % 16, lane / 16); query blocks qb of 16 rows, d in 16-byte chunks 4 ks + tau(g), tau = [0, 3, 1, 2].
__global__ __launch_bounds__(256, 2) void attn_bwd_dq16_asm_kernel(AttnArgs p) {
  using T = bf16_t;
  constexpr int D = 64, NW = 4, BQ = NW * 32, IMG = BKV * 128, NS = 3;
  static_assert(BKV == 64, "the generated stream is written for 64-key tiles");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nqb = p.Nq / BQ;
  const unsigned L = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = (int)(L / nqb), b = bh / p.H, hd = bh % p.H, q0 = (nqb - 1 - (int)(L % nqb)) * BQ;
  const T* Qp = (const T*)p.Q + (int64_t)b * p.q_bs + hd * D;
  const T* Kp = (const T*)p.K + (int64_t)b * p.k_bs + hd * D;
  const T* Vp = (const T*)p.V + (int64_t)b * p.v_bs + hd * D;
  const T* Gp = (const T*)p.dO + (int64_t)b * p.o_bs + hd * D;
  const T* Op = (const T*)p.O + (int64_t)b * p.o_bs + hd * D;
  const int ntiles = kv_limit(p, b, q0 + BQ - 1) / BKV;
  const int tau = (0x2130 >> (4 * g)) & 3;                        // d-chunk of lane group g within a k-step: 0, 3, 1, 2

  unsigned vo[4];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = (wave * 2 + j) * 8 + (lane >> 3);
    const int gg = (row >> 1) & 7, f = gg ^ ((gg & 1) << 2);
    vo[j] = (__umul24((unsigned)row, (unsigned)p.k_rs) + (unsigned)(((lane & 7) ^ f) * 8)) * 2u;
    vo[2 + j] = (__umul24((unsigned)row, (unsigned)p.v_rs) + (unsigned)(((lane & 7) ^ f) * 8)) * 2u;
  }
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_void_t*)smem;
  const unsigned ldsw = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)wave * 2048u);
  auto tile_row = [&](int tt) { return (tt < ntiles ? tt : ntiles - 1) * BKV; };
  auto k_base = [&](int tt) { return (uint64_t)(uintptr_t)(Kp + (int64_t)tile_row(tt) * p.k_rs); };
  auto v_base = [&](int tt) { return (uint64_t)(uintptr_t)(Vp + (int64_t)tile_row(tt) * p.v_rs); };
  dq_request_asm_slot0(vo, k_base(0), v_base(0), ldsw);
  dq_request_asm_slot1(vo, k_base(1), v_base(1), ldsw);

  bf16x8 qf[2][2], gf[2][2];
  f32x4 cl[2], cd[2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int qrow = q0 + wave * 32 + 16 * qb + c;
    float part = 0.0f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int d0 = 32 * ks + 8 * tau;
      qf[qb][ks] = *reinterpret_cast<const bf16x8*>(Qp + (int64_t)qrow * p.q_rs + d0);
      gf[qb][ks] = *reinterpret_cast<const bf16x8*>(Gp + (int64_t)qrow * p.o_rs + d0);
      const bf16x8 of = *reinterpret_cast<const bf16x8*>(Op + (int64_t)qrow * p.o_rs + d0);
#pragma unroll
      for (int e = 0; e < 8; ++e) part += to_f32<T>(of[e]) * to_f32<T>(gf[qb][ks][e]);
    }
    part += __shfl_xor(part, 16, 64);
    part += __shfl_xor(part, 32, 64);                           // delta = rowsum(dO * O) over the four lane groups
    const int64_t stat = ((int64_t)b * p.H + hd) * p.Nq + qrow;
    const float lse2 = p.LSE[stat] * LOG2E;
    if (g == 0) {
      p.delta[stat] = -lse2;
      p.delta[(int64_t)p.B * p.H * p.Nq + stat] = -part;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { cl[qb][r] = -lse2; cd[qb][r] = -part; }
  }
  f32x4 dq[4][2];
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) dq[db][qb] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  unsigned aq[2], va[4];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) aq[ks] = lds0 + (unsigned)Img<T, D>::off(c, (4 * ks + tau) * 16);
#pragma unroll
  for (int db = 0; db < 4; ++db) va[db] = lds0 + (unsigned)Img<T, D>::off(4 * g + (c >> 2), (16 * db + 4 * (c & 3)) * 2);
  asm volatile("" ::"v"(qf[0][0]), "v"(qf[0][1]), "v"(qf[1][0]), "v"(qf[1][1]), "v"(gf[0][0]), "v"(gf[0][1]), "v"(gf[1][0]), "v"(gf[1][1]),
               "v"(cl[0]), "v"(cl[1]), "v"(cd[0]), "v"(cd[1]));
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  for (int t = 0; t < ntiles; t += 3) {
    dq16_tile_asm_slot0(dq, qf, gf, cl, cd, aq, va, vo, k_base(t + 2), v_base(t + 2), ldsw);
    if (t + 1 < ntiles) dq16_tile_asm_slot1(dq, qf, gf, cl, cd, aq, va, vo, k_base(t + 3), v_base(t + 3), ldsw);
    if (t + 2 < ntiles) dq16_tile_asm_slot2(dq, qf, gf, cl, cd, aq, va, vo, k_base(t + 4), v_base(t + 4), ldsw);
  }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
  // store: block (db, qb) of this lane = dQ[query 16 qb + c][d = 16 db + 4 g .. + 3]
  T* dQp = (T*)p.dQ + (int64_t)b * p.q_bs + hd * D;
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int qrow = q0 + wave * 32 + 16 * qb + c;
    const float* table = p.rope_table ? p.rope_table + (int64_t)b * p.rope_bs + (int64_t)(p.rope_off + qrow) * D : nullptr;
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      const int d = 16 * db + 4 * g;
      const float a0 = dq[db][qb][0] * p.scale, a1 = dq[db][qb][1] * p.scale, a2 = dq[db][qb][2] * p.scale, a3 = dq[db][qb][3] * p.scale;
      float o0 = a0, o1 = a1, o2 = a2, o3 = a3;
      if (table) {                                              // inverse RoPE: pairs (d, d + 1) rotated by -angle, as store_rows_T_rope
        const f32x4 cs = *reinterpret_cast<const f32x4*>(table + d);
        o0 = a0 * cs[0] + a1 * cs[1]; o1 = -a0 * cs[1] + a1 * cs[0];
        o2 = a2 * cs[2] + a3 * cs[3]; o3 = -a2 * cs[3] + a3 * cs[2];
      }
      const bf16x4 v = {(bf16_t)o0, (bf16_t)o1, (bf16_t)o2, (bf16_t)o3};
      *reinterpret_cast<bf16x4*>(dQp + (int64_t)qrow * p.q_rs + d) = v;
    }
  }
}
#endif

#ifndef FK_NO_DKDV_ASM
#include "attn_dkdv_asm.inc"      // generated by tools/gen/gen_dkdv_asm.py: hand-placed instruction stream of one fully visible tile step
#endif
// ------------------------------------------------------------------------------------------------- dK, dV (pre-scaled Q)
template <int NW>
__global__ __launch_bounds__(NW * 64, 2) void attn_bwd_dkdv_ps_kernel(AttnArgs p) {
  using T = bf16_t;
  constexpr int D = 64, TQ = 64, BK = NW * 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // 3-slot ring (see attn_bwd_dq_ps_kernel): tiles t+1 and t+2 in flight while tile t is consumed
  constexpr int IMG = TQ * 128, NS = 3, PCS = 2 * (TQ / 8) / NW;   // PCS: LDS-DMA pieces per wave and tile (Q + dO)
  static_assert(PCS == 4 || PCS == 2, "vmcnt immediates below");
  auto qimg = [&](int i) -> char* { return smem + i * IMG; };
  auto gimg = [&](int i) -> char* { return smem + (NS + i) * IMG; };
  float* stats = reinterpret_cast<float*>(smem + 2 * NS * IMG);   // [NS buffers][2 (-lse2, -delta)][TQ]
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nkb = (p.Nk + BK - 1) / BK;
  const unsigned L = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = (int)(L / nkb), b = bh / p.H, hd = bh % p.H, k0 = (int)(L % nkb) * BK;
  const T* Qp = (const T*)p.Q + (int64_t)b * p.q_bs + hd * D;
  const T* Kp = (const T*)p.K + (int64_t)b * p.k_bs + hd * D;
  const T* Vp = (const T*)p.V + (int64_t)b * p.v_bs + hd * D;
  const T* Gp = (const T*)p.dO + (int64_t)b * p.o_bs + hd * D;
  const int krow = k0 + wave * 32 + li;
  const bool k_ok = krow < p.Nk;
  const bool prefix = p.mask_kind == FK_MASK_PREFIX, keypad = p.mask_kind == FK_MASK_KEYPAD;
  const int my_qf = ((prefix || keypad) && k_ok) ? p.qfirst[(int64_t)b * p.Nk + krow] : 0;

  Frag<T> kf[4], vf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    if (k_ok) {
      frag_load_contig<T>(kf[s], Kp + (int64_t)krow * p.k_rs + 16 * s + 8 * lh);
      frag_load_contig<T>(vf[s], Vp + (int64_t)krow * p.v_rs + 16 * s + 8 * lh);
    } else {
      frag_zero<T>(kf[s]);
      frag_zero<T>(vf[s]);
    }
  }
  const int qs = keypad ? 0 : (q_first(p, b, k0) / TQ) * TQ;
  const int ntiles = qs < p.Nq ? (p.Nq - qs + TQ - 1) / TQ : 0;
  // key padding: a wave whose 32 keys are all valid needs no predicate at all -- a padded query arrives with the staged statistic -inf
  // (its LSE is +inf), exactly like the rows past Nq of the last tile, so its P is 0 on the mask-free path too
  const bool keypad_keys_valid = !FK_KEYPAD_NO_FREE && __builtin_amdgcn_ballot_w64(k_ok && my_qf == 0) == 0;
  const int full_vis_q = keypad ? (keypad_keys_valid ? 0 : 0x3fffffff) : q_first(p, b, min(k0 + wave * 32 + 31, p.Nk - 1));
  // Row statistics: the dQ kernel has left -LSE*log2(e) and -delta in the workspace exactly as the accumulators want them (rows padded to
  // the tile, -inf / 0 past Nq), so a tile's 64 + 64 floats travel by LDS-DMA like the tile itself: one 256-byte piece each, issued by
  // waves 0 and 1 (no statistics registers, no arithmetic, no LDS stores in this kernel any more).
  const int nqp = (p.Nq + 63) / 64 * 64;
  const float* nl_g = p.delta + ((int64_t)b * p.H + hd) * nqp;
  const float* nd_g = nl_g + (int64_t)p.B * p.H * nqp;
  auto request_stats = [&](int qb, int slot) __attribute__((always_inline)) {
    if (wave < 2) {
      const float* src = (wave == 0 ? nl_g : nd_g) + qb + lane;
      __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)(stats + slot * 2 * TQ + wave * TQ), 4, 0, 0);
    }
  };
  DmaCursor<TQ, NW> qcur, gcur;
  auto request_tile = [&](int qb, int slot) __attribute__((always_inline)) {
    qcur.next(Qp, p.q_rs, qb, p.Nq, qimg(slot), wave, lane);
    gcur.next(Gp, p.o_rs, qb, p.Nq, gimg(slot), wave, lane);
    request_stats(qb, slot);
  };
  if (ntiles > 0) {
    qcur.init(Qp, p.q_rs, qs, wave, lane);
    gcur.init(Gp, p.o_rs, qs, wave, lane);
    request_tile(qs, 0);
    if (ntiles > 1) request_tile(qs + TQ, 1);
  }
  // tile t+1 has landed (all but this step's own pieces: PCS, plus the statistics piece of waves 0 and 1), tile t+2 stays in flight
  auto wait_next = [&]() __attribute__((always_inline)) {
    static_assert(PCS == 4 || PCS == 2, "vmcnt immediates");
    if (wave < 2) { if constexpr (PCS == 4) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)\n\ts_barrier" ::: "memory"); else asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
    else { if constexpr (PCS == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
  };
  if (ntiles > 1) wait_next(); else dma_wait_barrier();
  f32x16 dk[2], dv[2];
  zero_acc(dk);
  zero_acc(dv);
  FK_ST_DECL

  auto tile_step = [&](auto SL, int t) {
    constexpr int SLOT = decltype(SL)::value;
    const int qb = qs + t * TQ;
    // request tile t+2: slot (SLOT + 2) % 3 held tile t-1, which every wave left at the barrier that ended the previous step (an LDS-DMA
    // piece costs the issuing wave ~190 cycles, measured with -DFK_STAMP, wherever in the step it is issued)
    if (t + 2 < ntiles) request_tile(qb + 2 * TQ, (SLOT + 2) % NS);
    FK_ST(0)
    const char* qt = qimg(SLOT);
    const char* gt = gimg(SLOT);
    const float* stl = stats + SLOT * 2 * TQ;
    const float* std_ = stl + TQ;
    const bool boundary = (qb < full_vis_q) || (k0 + wave * 32 + 31 >= p.Nk);   // wave-uniform
    // the two 32-row query halves of the tile one after the other: 32 live score registers instead of 64 leave the scheduler room to
    // request the next LDS fragments ahead of the MFMAs that consume them
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      // S' = Q'K^T - lse2 and dP' = dO V^T - delta: the accumulators start from the staged row constants
      f32x16 sc, dp;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 l4 = *reinterpret_cast<const f32x4*>(stl + 32 * u + 8 * g + 4 * lh);
        const f32x4 d4 = *reinterpret_cast<const f32x4*>(std_ + 32 * u + 8 * g + 4 * lh);
#pragma unroll
        for (int j = 0; j < 4; ++j) { sc[4 * g + j] = l4[j]; dp[4 * g + j] = d4[j]; }
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        Frag<T> qf, gf;
        img_row<T, D>(qf, qt, 32 * u + li, s, lh);
        img_row<T, D>(gf, gt, 32 * u + li, s, lh);
        mma32<T>(sc, qf, kf[s]);   // S'[q][key]
        mma32<T>(dp, gf, vf[s]);   // dP'[q][key]
      }
      FK_ST(1)
      // P = exp2(S'), dS = P dP'
      if (!boundary) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pv = __builtin_amdgcn_exp2f(sc[r]);
          sc[r] = pv;
          dp[r] *= pv;
        }
      } else {
        const int kpos = krow + p.k_off;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float pv = __builtin_amdgcn_exp2f(sc[r]);
          const int q = qb + 32 * u + acc_row(r, lh);
          if (!(k_ok && (prefix ? q >= my_qf : (keypad ? (my_qf != 0 && q < p.Nq && p.limits[(int64_t)b * p.Nq + q] != 0) : visible(p.mask_kind, p.mask_c, q + p.q_off, kpos))))) pv = 0.0f;
          sc[r] = pv;
          dp[r] *= pv;
        }
      }
      FK_ST(2)
      // dV^T += dO^T P, dK^T += Q'^T dS
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        Frag<T> pf, df;
        frag_from_acc<T>(pf, sc, s);
        frag_from_acc<T>(df, dp, s);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          Frag<T> gtf, qtf;
          img_tr<T, D>(gtf, gt, 32 * u, s, 32 * dt, lane);
          img_tr<T, D>(qtf, qt, 32 * u, s, 32 * dt, lane);
          mma32<T>(dv[dt], gtf, pf);
          mma32<T>(dk[dt], qtf, df);
        }
      }
      FK_ST(3)
    }
    FK_ST(4)
    // tile t+1 (and wave 0's statistics of it) has landed, the request for tile t+2 stays in flight across the barrier
    if (t + 2 < ntiles) wait_next(); else dma_wait_barrier();
    FK_ST(5)
  };
  for (int t = 0; t < ntiles; t += 3) {
    tile_step(std::integral_constant<int, 0>{}, t);
    if (t + 1 < ntiles) tile_step(std::integral_constant<int, 1>{}, t + 1);
    if (t + 2 < ntiles) tile_step(std::integral_constant<int, 2>{}, t + 2);
  }
  FK_ST(6)
  // dK = scale * dS^T Q = ln(2) * dS^T Q'   (Q' = scale * log2(e) * Q)
  T* dKp = (T*)p.dK + (int64_t)b * p.k_bs + hd * D;
  T* dVp = (T*)p.dV + (int64_t)b * p.v_bs + hd * D;
  if (p.rope_table)           // wave-uniform: the stores swap registers between the half-waves, every lane takes part
    store_rows_T_rope<T, D>(dKp, p.k_rs, krow, k_ok, dk, LN2, lh, p.rope_table + (int64_t)b * p.rope_bs + (int64_t)(p.rope_off + krow) * D);
  else
    store_rows_T<T, D>(dKp, p.k_rs, krow, k_ok, dk, LN2, lh);
  store_rows_T<T, D>(dVp, p.v_rs, krow, k_ok, dv, 1.0f, lh);
  FK_ST(7)
  FK_ST_FLUSH(0)
}

#ifndef FK_NO_DKDV_ASM
// ------------------------------------------------------------------------------------------------- dK, dV: hand-placed stream
// The fully visible, fully aligned case (every tile of every workgroup visible to all its keys: no mask or a block-causal mask whose
// block is a multiple of 128 keys and of the 64-query tile, no offsets, Nk % 128 == 0, Nq % 64 == 0 — the benchmark's shape) runs the
// generated instruction stream of attn_dkdv_asm.inc; everything else stays on attn_bwd_dkdv_ps_kernel.  The kernel around the stream
// is kept small on purpose: the stream owns v100-v243, the compiler the remaining 112 registers (dK/dV accumulators 64, K/V fragments 32,
// seven LDS addresses).
__global__ __launch_bounds__(256, 2) void attn_bwd_dkdv_asm_kernel(AttnArgs p) {
  FK_LIFE_BEGIN
  using T = bf16_t;
  constexpr int D = 64, TQ = 64, BK = 128, NW = 4, IMG = TQ * 128, NS = 3;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  auto qimg = [&](int i) -> char* { return smem + i * IMG; };
  auto gimg = [&](int i) -> char* { return smem + (NS + i) * IMG; };
  float* stats = reinterpret_cast<float*>(smem + 2 * NS * IMG);
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nkb = p.Nk / BK;
  const unsigned L = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = (int)(L / nkb), b = bh / p.H, hd = bh % p.H, k0 = (int)(L % nkb) * BK;
  const T* Qp = (const T*)p.Q + (int64_t)b * p.q_bs + hd * D;
  const T* Kp = (const T*)p.K + (int64_t)b * p.k_bs + hd * D;
  const T* Vp = (const T*)p.V + (int64_t)b * p.v_bs + hd * D;
  const T* Gp = (const T*)p.dO + (int64_t)b * p.o_bs + hd * D;
  const int krow = k0 + wave * 32 + li;
  const int qs = q_first(p, b, k0);                      // a multiple of the mask block, hence of TQ
  const int ntiles = (p.Nq - qs) / TQ;
  const int nqp = p.Nq;
  const float* nl_g = p.delta + ((int64_t)b * p.H + hd) * nqp;
  const float* nd_g = nl_g + (int64_t)p.B * p.H * nqp;
  // The tile requests (LDS-DMA): per wave two 8-row groups of the Q image, two of the dO image and one row of statistics (waves 2, 3
  // repeat the rows of waves 0, 1: same bytes to the same place, and every wave then has the same five requests per tile in flight).
  // Address = wave-uniform 64-bit tile base (SGPR pair) + per-lane byte offset that never changes; the stream sets M0 itself.
  unsigned vo[5];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = (wave * 2 + j) * 8 + (lane >> 3);
    const int g = (row >> 1) & 7, f = g ^ ((g & 1) << 2);
    vo[j] = (__umul24((unsigned)row, (unsigned)p.q_rs) + (unsigned)(((lane & 7) ^ f) * 8)) * 2u;
    vo[2 + j] = (__umul24((unsigned)row, (unsigned)p.o_rs) + (unsigned)(((lane & 7) ^ f) * 8)) * 2u;
  }
  vo[4] = (unsigned)lane * 4u;
  const float* st_g = (wave & 1) ? nd_g : nl_g;
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_void_t*)smem;
  const unsigned ldsw = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)wave * 2048u);
  const unsigned ldss = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(2 * NS * IMG) + (unsigned)(wave & 1) * 256u);
  auto tile_row = [&](int tt) { return qs + (tt < ntiles ? tt : ntiles - 1) * TQ; };      // past the end: the last tile again (never read)
  auto q_base = [&](int tt) { return (uint64_t)(uintptr_t)(Qp + (int64_t)tile_row(tt) * p.q_rs); };
  auto g_base = [&](int tt) { return (uint64_t)(uintptr_t)(Gp + (int64_t)tile_row(tt) * p.o_rs); };
  auto s_base = [&](int tt) { return (uint64_t)(uintptr_t)(st_g + tile_row(tt)); };
  // prologue: tiles 0 and 1 requested first, then this wave's K / V fragments (registers for the whole kernel); the wait for the
  // fragments, placed by hipcc in front of the dummy use below, is in order and so covers the tiles too
  if (ntiles > 0) {
    dkdv_request_asm_slot0(vo, q_base(0), g_base(0), s_base(0), ldsw, ldss);
    dkdv_request_asm_slot1(vo, q_base(1), g_base(1), s_base(1), ldsw, ldss);
  }
  bf16x8 kfv[4], vfv[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    kfv[s] = *reinterpret_cast<const bf16x8*>(Kp + (int64_t)krow * p.k_rs + 16 * s + 8 * lh);
    vfv[s] = *reinterpret_cast<const bf16x8*>(Vp + (int64_t)krow * p.v_rs + 16 * s + 8 * lh);
  }
  f32x16 dk[2], dv[2];
  zero_acc(dk);
  zero_acc(dv);
  unsigned aq[4], va0, va1;
#pragma unroll
  for (int s_ = 0; s_ < 4; ++s_) aq[s_] = lds0 + (unsigned)Img<T, D>::off(li, (16 * s_ + 8 * lh) * 2);
  {
    const int g4 = lane >> 4, i16 = lane & 15, hh = g4 >> 1;
    const int rpart = (4 * hh + (i16 >> 2)) * 128 + (i16 & 1) * 8;
    const int c0 = 2 * (g4 & 1) + ((i16 & 3) >> 1), gg0 = 2 * hh + (i16 >> 3), f0 = gg0 ^ ((gg0 & 1) << 2);
    va0 = lds0 + (unsigned)(rpart + ((c0 ^ f0) << 4));
    va1 = lds0 + (unsigned)(rpart + (((c0 ^ f0) ^ 4) << 4));
  }
  const unsigned ast = lds0 + (unsigned)(2 * NS * IMG) + (unsigned)lh * 16u;
  if (ntiles > 0) {
    asm volatile("" ::"v"(kfv[0]), "v"(kfv[1]), "v"(kfv[2]), "v"(kfv[3]), "v"(vfv[0]), "v"(vfv[1]), "v"(vfv[2]), "v"(vfv[3]));
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    // Each step: tile t from ring slot t % 3, with the requests of tile t + 2 issued from inside the stream and, at its end, the wait
    // for tile t + 1 and the barrier.
    for (int t = 0; t < ntiles; t += 3) {
      dkdv_tile_asm_slot0(dk[0], dk[1], dv[0], dv[1], kfv, vfv, aq, va0, va1, ast, vo, q_base(t + 2), g_base(t + 2), s_base(t + 2), ldsw, ldss);
      if (t + 1 < ntiles) {
        dkdv_tile_asm_slot1(dk[0], dk[1], dv[0], dv[1], kfv, vfv, aq, va0, va1, ast, vo, q_base(t + 3), g_base(t + 3), s_base(t + 3), ldsw, ldss);
      }
      if (t + 2 < ntiles) {
        dkdv_tile_asm_slot2(dk[0], dk[1], dv[0], dv[1], kfv, vfv, aq, va0, va1, ast, vo, q_base(t + 4), g_base(t + 4), s_base(t + 4), ldsw, ldss);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the (unused) requests past the last tile have landed before the workgroup ends
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");    // the stream's last MFMAs have retired before the compiler reads the accumulators
  T* dKp = (T*)p.dK + (int64_t)b * p.k_bs + hd * D;
  T* dVp = (T*)p.dV + (int64_t)b * p.v_bs + hd * D;
  if (p.rope_table)
    store_rows_T_rope<T, D>(dKp, p.k_rs, krow, true, dk, LN2, lh, p.rope_table + (int64_t)b * p.rope_bs + (int64_t)(p.rope_off + krow) * D);
  else
    store_rows_T<T, D>(dKp, p.k_rs, krow, true, dk, LN2, lh);
  store_rows_T<T, D>(dVp, p.v_rs, krow, true, dv, 1.0f, lh);
  FK_LIFE_END(2)
}
#endif

#ifndef FK_NO_DKDVW_ASM
#include "attn_dkdvw_asm.inc"     // generated by tools/gen/gen_dkdvw_asm.py: the tile step of the wide dK/dV kernel
// ------------------------------------------------------------------------------------------------- dK, dV: 64 keys per wave, one wave per SIMD
// Same algorithm, ring and requests as attn_bwd_dkdv_asm_kernel; a wave owns 64 keys (two 32-key blocks), a workgroup 256, so every LDS
// fragment of the Q / dO tile feeds two MFMAs (1.25 LDS reads per MFMA instead of 2) and a tile fetch serves twice the keys.  The 128
// accumulator registers (dK, dV of both key blocks) and the 64 K / V fragment registers are "a" operands of the stream: they live in the
// accumulator half of the 512-register file, the stream's temporaries (v40-v231) and the compiler's values in the other half.
// Launched when the fully-visible / aligned conditions of the narrow stream hold for 256-key workgroups (Nk % 256 == 0, mask block % 256 == 0).
__global__ __launch_bounds__(256, 1) void attn_bwd_dkdvw_asm_kernel(AttnArgs p) {
  using T = bf16_t;
  constexpr int D = 64, TQ = 64, BK = 256, IMG = TQ * 128, NS = 3;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nkb = p.Nk / BK;
  const unsigned L = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = (int)(L / nkb), b = bh / p.H, hd = bh % p.H, k0 = (int)(L % nkb) * BK;
  const T* Qp = (const T*)p.Q + (int64_t)b * p.q_bs + hd * D;
  const T* Kp = (const T*)p.K + (int64_t)b * p.k_bs + hd * D;
  const T* Vp = (const T*)p.V + (int64_t)b * p.v_bs + hd * D;
  const T* Gp = (const T*)p.dO + (int64_t)b * p.o_bs + hd * D;
  const int krow0 = k0 + wave * 64 + li;                 // key block kb: krow0 + 32 kb
  const int qs = q_first(p, b, k0);                      // a multiple of the mask block, hence of TQ
  const int ntiles = (p.Nq - qs) / TQ;
  const int nqp = p.Nq;
  const float* nl_g = p.delta + ((int64_t)b * p.H + hd) * nqp;
  const float* nd_g = nl_g + (int64_t)p.B * p.H * nqp;
  unsigned vo[5];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = (wave * 2 + j) * 8 + (lane >> 3);
    const int g = (row >> 1) & 7, f = g ^ ((g & 1) << 2);
    vo[j] = (__umul24((unsigned)row, (unsigned)p.q_rs) + (unsigned)(((lane & 7) ^ f) * 8)) * 2u;
    vo[2 + j] = (__umul24((unsigned)row, (unsigned)p.o_rs) + (unsigned)(((lane & 7) ^ f) * 8)) * 2u;
  }
  vo[4] = (unsigned)lane * 4u;
  const float* st_g = (wave & 1) ? nd_g : nl_g;
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_void_t*)smem;
  const unsigned ldsw = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)wave * 2048u);
  const unsigned ldss = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(2 * NS * IMG) + (unsigned)(wave & 1) * 256u);
  auto tile_row = [&](int tt) { return qs + (tt < ntiles ? tt : ntiles - 1) * TQ; };
  auto q_base = [&](int tt) { return (uint64_t)(uintptr_t)(Qp + (int64_t)tile_row(tt) * p.q_rs); };
  auto g_base = [&](int tt) { return (uint64_t)(uintptr_t)(Gp + (int64_t)tile_row(tt) * p.o_rs); };
  auto s_base = [&](int tt) { return (uint64_t)(uintptr_t)(st_g + tile_row(tt)); };
  if (ntiles > 0) {
    dkdv_request_asm_slot0(vo, q_base(0), g_base(0), s_base(0), ldsw, ldss);
    dkdv_request_asm_slot1(vo, q_base(1), g_base(1), s_base(1), ldsw, ldss);
  }
  bf16x8 kfv[2][4], vfv[2][4];
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      kfv[kb][s] = *reinterpret_cast<const bf16x8*>(Kp + (int64_t)(krow0 + 32 * kb) * p.k_rs + 16 * s + 8 * lh);
      vfv[kb][s] = *reinterpret_cast<const bf16x8*>(Vp + (int64_t)(krow0 + 32 * kb) * p.v_rs + 16 * s + 8 * lh);
    }
  f32x16 dk[2][2], dv[2][2];
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) { zero_acc(dk[kb]); zero_acc(dv[kb]); }
  unsigned aq[4], va0, va1;
#pragma unroll
  for (int s_ = 0; s_ < 4; ++s_) aq[s_] = lds0 + (unsigned)Img<T, D>::off(li, (16 * s_ + 8 * lh) * 2);
  {
    const int g4 = lane >> 4, i16 = lane & 15, hh = g4 >> 1;
    const int rpart = (4 * hh + (i16 >> 2)) * 128 + (i16 & 1) * 8;
    const int c0 = 2 * (g4 & 1) + ((i16 & 3) >> 1), gg0 = 2 * hh + (i16 >> 3), f0 = gg0 ^ ((gg0 & 1) << 2);
    va0 = lds0 + (unsigned)(rpart + ((c0 ^ f0) << 4));
    va1 = lds0 + (unsigned)(rpart + (((c0 ^ f0) ^ 4) << 4));
  }
  const unsigned ast = lds0 + (unsigned)(2 * NS * IMG) + (unsigned)lh * 16u;
  if (ntiles > 0) {
    asm volatile("" ::"a"(kfv[0][0]), "a"(kfv[0][1]), "a"(kfv[0][2]), "a"(kfv[0][3]), "a"(vfv[0][0]), "a"(vfv[0][1]), "a"(vfv[0][2]), "a"(vfv[0][3]),
                 "a"(kfv[1][0]), "a"(kfv[1][1]), "a"(kfv[1][2]), "a"(kfv[1][3]), "a"(vfv[1][0]), "a"(vfv[1][1]), "a"(vfv[1][2]), "a"(vfv[1][3]));
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    // The steps are software pipelined: a step finds half 0's statistics and row fragments of ITS tile in the carried registers (read by the
    // previous step behind its barrier; by dkdvw_prefetch_asm for tile 0) and starts with its MFMAs; the wait for tile t + 1 and the
    // barrier sit inside the step, the requests of tile t + 2 behind them.
    f32x16 c0, c1, c2, c3, c4, c5;
    dkdvw_prefetch_asm(c0, c1, c2, c3, c4, c5, aq, ast);
    for (int t = 0; t < ntiles; t += 3) {
      dkdvw_tile_asm_slot0(dk, dv, kfv, vfv, c0, c1, c2, c3, c4, c5, aq, va0, va1, ast, vo, q_base(t + 2), g_base(t + 2), s_base(t + 2), ldsw, ldss);
      if (t + 1 < ntiles) dkdvw_tile_asm_slot1(dk, dv, kfv, vfv, c0, c1, c2, c3, c4, c5, aq, va0, va1, ast, vo, q_base(t + 3), g_base(t + 3), s_base(t + 3), ldsw, ldss);
      if (t + 2 < ntiles) dkdvw_tile_asm_slot2(dk, dv, kfv, vfv, c0, c1, c2, c3, c4, c5, aq, va0, va1, ast, vo, q_base(t + 4), g_base(t + 4), s_base(t + 4), ldsw, ldss);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the (unused) requests past the last tile have landed before the workgroup ends
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");    // the stream's last MFMAs have retired before the compiler reads the accumulators
  T* dKp = (T*)p.dK + (int64_t)b * p.k_bs + hd * D;
  T* dVp = (T*)p.dV + (int64_t)b * p.v_bs + hd * D;
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
    const int krow = krow0 + 32 * kb;
    if (p.rope_table)
      store_rows_T_rope<T, D>(dKp, p.k_rs, krow, true, dk[kb], LN2, lh, p.rope_table + (int64_t)b * p.rope_bs + (int64_t)(p.rope_off + krow) * D);
    else
      store_rows_T<T, D>(dKp, p.k_rs, krow, true, dk[kb], LN2, lh);
    store_rows_T<T, D>(dVp, p.v_rs, krow, true, dv[kb], 1.0f, lh);
  }
}
#endif

// ------------------------------------------------------------------------------------------------- host
template <typename T, int D> size_t fwd_lds() {
  return Img<T, D>::SWZ ? (size_t)3 * 2 * BKV * 128 : (size_t)2 * BKV * (AT<T, D>::RSTRIDE + AT<T, D>::VSTRIDE);
}
template <typename T, int D> size_t dq_lds() { return 4 * BKV * AT<T, D>::RSTRIDE; }
constexpr size_t DQ_PS_LDS = 6 * BKV * 128, DKDV_PS_LDS = 6 * 64 * 128 + 3 * 2 * 64 * sizeof(float);
static_assert(DQ_PS_LDS <= 160 * 1024, "LDS of a CU");
static_assert(DKDV_PS_LDS <= 160 * 1024, "LDS of a CU");
template <typename T, int D> size_t dkdv_lds() { return 4 * 64 * AT<T, D>::RSTRIDE + 4 * 64 * sizeof(float); }

template <typename K> void allow_lds(K kernel, size_t bytes) {
  if (bytes > 65536) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

#ifndef FK_FWD_NW
#define FK_FWD_NW 8
#endif
#ifndef FK_DQ_NW
#define FK_DQ_NW 4
#endif
#ifndef FK_DKDV_NW
#define FK_DKDV_NW 4
#endif
template <typename T, int D> int launch_fwd(const AttnArgs& a, hipStream_t s) {
  constexpr int NW = Img<T, D>::SWZ ? 8 : 4;
  if constexpr (Img<T, D>::SWZ) {
    if (a.flags & FK_ATTN_Q_PRESCALED) {
#ifndef FK_NO_FWD_ASM
      // every tile of every workgroup fully visible and aligned -> the generated instruction streams
      if (a.q_off == 0 && a.k_off == 0 && (a.mask_kind == FK_MASK_NONE || (a.mask_kind == FK_MASK_BLOCK_CAUSAL && a.mask_c % 128 == 0)) &&
          a.Nq % 128 == 0 && a.Nk % 64 == 0) {
        allow_lds(attn_fwd_asm_kernel, FWD_ASM_LDS);
        hipLaunchKernelGGL(attn_fwd_asm_kernel, dim3((unsigned)(a.Nq / 128 * a.H * a.B)), dim3(256), FWD_ASM_LDS, s, a);
        return 0;
      }
#endif
      constexpr int NWF = FK_FWD_NW;
      dim3 grid((unsigned)(((a.Nq + NWF * 32 - 1) / (NWF * 32)) * a.H * a.B));
      const size_t lds = fwd_lds<T, D>();
      allow_lds(attn_fwd_ps_kernel<NWF>, lds);
      hipLaunchKernelGGL((attn_fwd_ps_kernel<NWF>), grid, dim3(NWF * 64), lds, s, a);
      return 0;
    }
  }
  if constexpr (Img<T, D>::SWZ) {
    static const bool no_fewq = getenv("FK_ATTN_NO_FEWQ") != nullptr;      // tuning knob
    if (!no_fewq && a.Nq <= 32 && a.Nk >= 1024 && a.mask_kind == FK_MASK_NONE && !a.drop_thresh) {   // few queries, long context: the waves split the keys
      allow_lds(attn_fwd_fewq_kernel, FEWQ_LDS);
      hipLaunchKernelGGL(attn_fwd_fewq_kernel, dim3((unsigned)(a.H * a.B)), dim3(FEWQ_NW * 64), FEWQ_LDS, s, a);
      return 0;
    }
  }
  dim3 grid((unsigned)(((a.Nq + NW * 32 - 1) / (NW * 32)) * a.H * a.B));
  const size_t lds = fwd_lds<T, D>();
  allow_lds(attn_fwd_kernel<T, D, NW>, lds);
  hipLaunchKernelGGL((attn_fwd_kernel<T, D, NW>), grid, dim3(NW * 64), lds, s, a);
  return 0;
}
template <typename T, int D> int launch_bwd(const AttnArgs& a, hipStream_t s) {
  // dQ first: it also writes delta = rowsum(dO * O) [B, H, Nq], which the dK/dV kernel reads
  dim3 gk((unsigned)(((a.Nk + 127) / 128) * a.H * a.B));
  const size_t lds_kv = dkdv_lds<T, D>(), lds_q = dq_lds<T, D>();
  allow_lds(attn_bwd_dkdv_kernel<T, D>, lds_kv);
  allow_lds(attn_bwd_dq_kernel<T, D>, lds_q);
  dim3 gq((unsigned)(((a.Nq + BQ - 1) / BQ) * a.H * a.B));
  if constexpr (Img<T, D>::SWZ) {
    if (a.flags & FK_ATTN_Q_PRESCALED) {
      constexpr int NWQ = FK_DQ_NW, NWK = FK_DKDV_NW;
      dim3 gq2((unsigned)(((a.Nq + NWQ * 32 - 1) / (NWQ * 32)) * a.H * a.B)), gk2((unsigned)(((a.Nk + NWK * 32 - 1) / (NWK * 32)) * a.H * a.B));
      // every tile of every workgroup fully visible and aligned -> the generated instruction streams
      const bool vis_all = a.q_off == 0 && a.k_off == 0 && (a.mask_kind == FK_MASK_NONE || (a.mask_kind == FK_MASK_BLOCK_CAUSAL && a.mask_c % 128 == 0));
      bool dq_done = false;
#ifndef FK_NO_DQ_ASM
      if (vis_all && a.Nq % 128 == 0 && a.Nk % 64 == 0) {
#ifdef FK_DQ16_ASM
        allow_lds(attn_bwd_dq16_asm_kernel, DQ_PS_LDS);
        hipLaunchKernelGGL(attn_bwd_dq16_asm_kernel, dim3((unsigned)(a.Nq / 128 * a.H * a.B)), dim3(256), DQ_PS_LDS, s, a);
#else
        allow_lds(attn_bwd_dq_asm_kernel, DQ_PS_LDS);
        hipLaunchKernelGGL(attn_bwd_dq_asm_kernel, dim3((unsigned)(a.Nq / 128 * a.H * a.B)), dim3(256), DQ_PS_LDS, s, a);
#endif
        dq_done = true;
      }
#endif
      if (!dq_done) {
        allow_lds(attn_bwd_dq_ps_kernel<NWQ>, DQ_PS_LDS);
        hipLaunchKernelGGL(attn_bwd_dq_ps_kernel<NWQ>, gq2, dim3(NWQ * 64), DQ_PS_LDS, s, a);
      }
#ifndef FK_NO_DKDVW_ASM
      static const bool no_wide = getenv("FK_ATTN_NO_DKDVW") != nullptr;    // tuning knob: the 32-keys-per-wave stream instead
      const bool vis_256 = a.q_off == 0 && a.k_off == 0 && (a.mask_kind == FK_MASK_NONE || (a.mask_kind == FK_MASK_BLOCK_CAUSAL && a.mask_c % 256 == 0));
      if (!no_wide && vis_256 && a.Nk % 256 == 0 && a.Nq % 64 == 0) {
        allow_lds(attn_bwd_dkdvw_asm_kernel, DKDV_PS_LDS);
        hipLaunchKernelGGL(attn_bwd_dkdvw_asm_kernel, dim3((unsigned)(a.Nk / 256 * a.H * a.B)), dim3(256), DKDV_PS_LDS, s, a);
        return 0;
      }
#endif
#ifndef FK_NO_DKDV_ASM
      if (vis_all && a.Nk % 128 == 0 && a.Nq % 64 == 0) {
        allow_lds(attn_bwd_dkdv_asm_kernel, DKDV_PS_LDS);
        hipLaunchKernelGGL(attn_bwd_dkdv_asm_kernel, gk2, dim3(256), DKDV_PS_LDS, s, a);
        return 0;
      }
#endif
      allow_lds(attn_bwd_dkdv_ps_kernel<NWK>, DKDV_PS_LDS);
      hipLaunchKernelGGL(attn_bwd_dkdv_ps_kernel<NWK>, gk2, dim3(NWK * 64), DKDV_PS_LDS, s, a);
      return 0;
    }
  }
  bool dq_done = false;
  if constexpr (Img<T, D>::SWZ) {
    static const bool no_fewq = getenv("FK_ATTN_NO_FEWQ") != nullptr;      // tuning knob
    if (!no_fewq && a.Nq <= 32 && a.Nk >= 1024 && a.mask_kind == FK_MASK_NONE && !a.drop_thresh && !a.rope_table) {
      allow_lds(attn_bwd_dq_fewq_kernel, FEWQ_LDS);                        // few queries, long context: the waves split the keys
      hipLaunchKernelGGL(attn_bwd_dq_fewq_kernel, dim3((unsigned)(a.H * a.B)), dim3(FEWQ_NW * 64), FEWQ_LDS, s, a);
      dq_done = true;
    }
  }
  if (!dq_done) hipLaunchKernelGGL((attn_bwd_dq_kernel<T, D>), gq, dim3(NT), lds_q, s, a);
  hipLaunchKernelGGL((attn_bwd_dkdv_kernel<T, D>), gk, dim3(NT), lds_kv, s, a);
  return 0;
}

int check_common(const char* name, int64_t B, int64_t H, int64_t Nq, int64_t Nk, int64_t D, int dtype, int mask_kind,
                 int64_t mask_c, const int64_t* strides, int nstr) {
  FK_CHECK_ARG(dtype == FK_F32 || dtype == FK_BF16, "%s: bad dtype %d", name, dtype);
  FK_CHECK_ARG(D == 8 || D == 16 || D == 32 || D == 64 || (D == 128 && dtype == FK_BF16),
               "%s: head_dim %lld unsupported (8/16/32/64, 128 for bf16)", name, (long long)D);
  FK_CHECK_ARG(B > 0 && H > 0 && Nq > 0 && Nk > 0 && B < 65536 && H < 65536 && Nq < (1LL << 30) && Nk < (1LL << 30) && B * H * ((Nq + 127) / 128) < (1LL << 31) && B * H * ((Nk + 127) / 128) < (1LL << 31),
               "%s: bad shape B=%lld H=%lld Nq=%lld Nk=%lld", name, (long long)B, (long long)H, (long long)Nq, (long long)Nk);
  FK_CHECK_ARG(mask_kind == FK_MASK_NONE || mask_kind == FK_MASK_CAUSAL || mask_kind == FK_MASK_BLOCK_CAUSAL || mask_kind == FK_MASK_PREFIX || mask_kind == FK_MASK_KEYPAD ||
               mask_kind == FK_MASK_DENSE, "%s: mask kind %d not supported", name, mask_kind);
  FK_CHECK_ARG(mask_kind != FK_MASK_DENSE || (mask_c >= 0 && mask_c < (1LL << 31) && Nq * Nk < (1LL << 31)), "%s: dense mask too large", name);
  FK_CHECK_ARG(mask_kind != FK_MASK_BLOCK_CAUSAL || mask_c > 0, "%s: block-causal mask needs block size > 0", name);
  const int vec = dtype == FK_BF16 ? 8 : 4;
  for (int i = 0; i < nstr; ++i)
    FK_CHECK_ARG(strides[i] % vec == 0, "%s: strides must be multiples of %d elements (16 bytes)", name, vec);
  if (dtype == FK_BF16 && D == 64) {          // the LDS-DMA tile loader addresses rows with 24-bit multiplies and 32-bit offsets
    FK_CHECK_ARG(Nq < (1LL << 24) && Nk < (1LL << 24), "%s: bf16 D=64 path needs fewer than 2^24 rows", name);
    for (int i = 1; i < nstr; i += 2)          // {batch stride, row stride} pairs: the row strides
      FK_CHECK_ARG(strides[i] >= 0 && strides[i] < (1LL << 24) && strides[i] * (Nq > Nk ? Nq : Nk) < (1LL << 31),
                   "%s: bf16 D=64 path needs row strides < 2^24 elements and a head slab < 2^31 elements", name);
  }
  return FK_OK;
}

int set_dropout(const char* name, AttnArgs& a, float drop_p, const uint32_t* drop_seed, uint32_t drop_site, int flags) {
  a.drop_seed = nullptr; a.drop_site = 0; a.drop_thresh = 0; a.drop_scale = 1.0f;
  if (drop_p == 0.0f) return FK_OK;
  FK_CHECK_ARG(drop_p > 0.0f && drop_p < 1.0f && drop_seed, "%s: dropout needs 0 <= p < 1 and the device seed words", name);
  FK_CHECK_ARG(!(flags & FK_ATTN_Q_PRESCALED), "%s: dropout runs on the generic kernels (no FK_ATTN_Q_PRESCALED)", name);
  FK_CHECK_ARG((int64_t)a.B * a.H * a.Nq < (1LL << 32), "%s: dropout rows are numbered with 32 bits", name);
  const double t = (double)drop_p * 4294967296.0;
  a.drop_seed = drop_seed; a.drop_site = drop_site; a.drop_scale = 1.0f / (1.0f - drop_p);
  a.drop_thresh = t >= 4294967295.0 ? 4294967295u : (t < 1.0 ? 1u : (unsigned)t);
  return FK_OK;
}

#define FK_ATTN_DISPATCH(FN, args, stream)                                                   \
  do {                                                                                        \
    if (dtype == FK_BF16) {                                                                   \
      switch (D) {                                                                            \
        case 8: FN<bf16_t, 8>(args, stream); break;                                           \
        case 16: FN<bf16_t, 16>(args, stream); break;                                         \
        case 32: FN<bf16_t, 32>(args, stream); break;                                         \
        case 64: FN<bf16_t, 64>(args, stream); break;                                         \
        default: FN<bf16_t, 128>(args, stream); break;                                        \
      }                                                                                       \
    } else {                                                                                  \
      switch (D) {                                                                            \
        case 8: FN<float, 8>(args, stream); break;                                            \
        case 16: FN<float, 16>(args, stream); break;                                          \
        case 32: FN<float, 32>(args, stream); break;                                          \
        default: FN<float, 64>(args, stream); break;                                          \
      }                                                                                       \
    }                                                                                         \
  } while (0)

}  // namespace

extern "C" {

#ifdef FK_STAMP
int fk_debug_stamps(unsigned long long* out, int reset) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(out, HIP_SYMBOL(fk_stamp_acc), sizeof(unsigned long long) * 16);
  if (reset) { unsigned long long z[16] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(fk_stamp_acc), z, sizeof(z)); }
  return 0;
}
#endif

int fk_attn_fwd_dropout(const void* Q, const void* K, const void* V, void* O, float* LSE, int64_t B, int64_t H, int64_t Nq,
                        int64_t Nk, int64_t D, int64_t q_bs, int64_t q_rs, int64_t k_bs, int64_t k_rs, int64_t v_bs,
                        int64_t v_rs, int64_t o_bs, int64_t o_rs, int mask_kind, int64_t mask_c, int64_t q_off, int64_t k_off,
                        const int32_t* limits, const int32_t* qfirst, float scale, int flags, float drop_p, const uint32_t* drop_seed,
                        uint32_t drop_site, int dtype, void* stream) {
  const int64_t st[8] = {q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs};
  int rc = check_common("fk_attn_fwd", B, H, Nq, Nk, D, dtype, mask_kind, mask_c, st, 8);
  if (rc) return rc;
  FK_CHECK_ARG((flags & ~FK_ATTN_Q_PRESCALED) == 0 && (!(flags & FK_ATTN_Q_PRESCALED) || (dtype == FK_BF16 && D == 64)),
               "fk_attn_fwd: bad flags %d (FK_ATTN_Q_PRESCALED needs bf16, D = 64)", flags);
  FK_CHECK_ARG(Q && K && V && O, "fk_attn_fwd: null pointer");
  FK_CHECK_ARG((((uintptr_t)Q | (uintptr_t)K | (uintptr_t)V | (uintptr_t)O) & 15) == 0, "fk_attn_fwd: pointers must be 16-byte aligned");
  AttnArgs a{};
  a.Q = Q; a.K = K; a.V = V; a.Out = O; a.LSE = LSE;
  a.q_bs = q_bs; a.q_rs = q_rs; a.k_bs = k_bs; a.k_rs = k_rs; a.v_bs = v_bs; a.v_rs = v_rs; a.o_bs = o_bs; a.o_rs = o_rs;
  a.B = (int)B; a.H = (int)H; a.Nq = (int)Nq; a.Nk = (int)Nk;
  a.mask_kind = mask_kind; a.mask_c = (int)mask_c; a.q_off = (int)q_off; a.k_off = (int)k_off; a.scale = scale;
  FK_CHECK_ARG((mask_kind != FK_MASK_PREFIX && mask_kind != FK_MASK_KEYPAD) || (limits && qfirst), "fk_attn_fwd: prefix / key-padding masks need both tables");
  FK_CHECK_ARG(mask_kind != FK_MASK_DENSE || (limits && !(flags & FK_ATTN_Q_PRESCALED) && q_off >= 0 && q_off < (1LL << 31) && k_off == 0),
               "fk_attn_fwd: a dense mask needs its uint8 table in `limits`, its head stride (>= 0) in q_off, k_off = 0 and the generic kernels (no FK_ATTN_Q_PRESCALED)");
  a.limits = limits; a.qfirst = qfirst; a.flags = flags;
  rc = set_dropout("fk_attn_fwd", a, drop_p, drop_seed, drop_site, flags);
  if (rc) return rc;
  FK_ATTN_DISPATCH(launch_fwd, a, (hipStream_t)stream);
  FK_CHECK_LAUNCH("fk_attn_fwd");
  return FK_OK;
}

int fk_attn_fwd(const void* Q, const void* K, const void* V, void* O, float* LSE, int64_t B, int64_t H, int64_t Nq,
                int64_t Nk, int64_t D, int64_t q_bs, int64_t q_rs, int64_t k_bs, int64_t k_rs, int64_t v_bs,
                int64_t v_rs, int64_t o_bs, int64_t o_rs, int mask_kind, int64_t mask_c, int64_t q_off, int64_t k_off,
                const int32_t* limits, const int32_t* qfirst, float scale, int flags, int dtype, void* stream) {
  return fk_attn_fwd_dropout(Q, K, V, O, LSE, B, H, Nq, Nk, D, q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs, mask_kind, mask_c, q_off, k_off,
                             limits, qfirst, scale, flags, 0.0f, nullptr, 0, dtype, stream);
}

int fk_attn_bwd_dropout(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* LSE,
                        void* dQ, void* dK, void* dV, float* delta_ws, int64_t B, int64_t H, int64_t Nq, int64_t Nk, int64_t D,
                        int64_t q_bs, int64_t q_rs, int64_t k_bs, int64_t k_rs, int64_t v_bs, int64_t v_rs, int64_t o_bs,
                        int64_t o_rs, int mask_kind, int64_t mask_c, int64_t q_off, int64_t k_off, const int32_t* limits,
                        const int32_t* qfirst, float scale, const float* rope_table, int64_t rope_bs, int64_t rope_off, int flags,
                        float drop_p, const uint32_t* drop_seed, uint32_t drop_site, int dtype, void* stream) {
  const int64_t st[8] = {q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs};
  int rc = check_common("fk_attn_bwd", B, H, Nq, Nk, D, dtype, mask_kind, mask_c, st, 8);
  if (rc) return rc;
  FK_CHECK_ARG((flags & ~FK_ATTN_Q_PRESCALED) == 0 && (!(flags & FK_ATTN_Q_PRESCALED) || (dtype == FK_BF16 && D == 64)),
               "fk_attn_bwd: bad flags %d (FK_ATTN_Q_PRESCALED needs bf16, D = 64)", flags);
  FK_CHECK_ARG(Q && K && V && O && dO && LSE && dQ && dK && dV && delta_ws, "fk_attn_bwd: null pointer");
  FK_CHECK_ARG((((uintptr_t)Q | (uintptr_t)K | (uintptr_t)V | (uintptr_t)O | (uintptr_t)dO | (uintptr_t)dQ | (uintptr_t)dK | (uintptr_t)dV) & 15) == 0,
               "fk_attn_bwd: pointers must be 16-byte aligned");
  AttnArgs a{};
  a.Q = Q; a.K = K; a.V = V; a.O = O; a.dO = dO; a.LSE = const_cast<float*>(LSE); a.delta = delta_ws;
  a.dQ = dQ; a.dK = dK; a.dV = dV;
  a.q_bs = q_bs; a.q_rs = q_rs; a.k_bs = k_bs; a.k_rs = k_rs; a.v_bs = v_bs; a.v_rs = v_rs; a.o_bs = o_bs; a.o_rs = o_rs;
  a.B = (int)B; a.H = (int)H; a.Nq = (int)Nq; a.Nk = (int)Nk;
  a.mask_kind = mask_kind; a.mask_c = (int)mask_c; a.q_off = (int)q_off; a.k_off = (int)k_off; a.scale = scale;
  FK_CHECK_ARG(!rope_table || (Nq == Nk && D % 4 == 0 && ((uintptr_t)rope_table & 15) == 0), "fk_attn_bwd: fused inverse RoPE needs self-attention (Nq == Nk)");
  a.rope_table = rope_table; a.rope_bs = rope_bs; a.rope_off = (int)rope_off;
  FK_CHECK_ARG((mask_kind != FK_MASK_PREFIX && mask_kind != FK_MASK_KEYPAD) || (limits && qfirst), "fk_attn_bwd: prefix / key-padding masks need both tables");
  FK_CHECK_ARG(mask_kind != FK_MASK_DENSE || (limits && !(flags & FK_ATTN_Q_PRESCALED) && q_off >= 0 && q_off < (1LL << 31) && k_off == 0),
               "fk_attn_bwd: a dense mask needs its uint8 table in `limits`, its head stride (>= 0) in q_off, k_off = 0 and the generic kernels (no FK_ATTN_Q_PRESCALED)");
  a.limits = limits; a.qfirst = qfirst; a.flags = flags;
  rc = set_dropout("fk_attn_bwd", a, drop_p, drop_seed, drop_site, flags);
  if (rc) return rc;
  FK_ATTN_DISPATCH(launch_bwd, a, (hipStream_t)stream);
  FK_CHECK_LAUNCH("fk_attn_bwd");
  return FK_OK;
}

int fk_attn_bwd(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* LSE,
                void* dQ, void* dK, void* dV, float* delta_ws, int64_t B, int64_t H, int64_t Nq, int64_t Nk, int64_t D,
                int64_t q_bs, int64_t q_rs, int64_t k_bs, int64_t k_rs, int64_t v_bs, int64_t v_rs, int64_t o_bs,
                int64_t o_rs, int mask_kind, int64_t mask_c, int64_t q_off, int64_t k_off, const int32_t* limits,
                const int32_t* qfirst, float scale, const float* rope_table, int64_t rope_bs, int64_t rope_off, int flags,
                int dtype, void* stream) {
  return fk_attn_bwd_dropout(Q, K, V, O, dO, LSE, dQ, dK, dV, delta_ws, B, H, Nq, Nk, D, q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs,
                             mask_kind, mask_c, q_off, k_off, limits, qfirst, scale, rope_table, rope_bs, rope_off, flags, 0.0f, nullptr, 0,
                             dtype, stream);
}

}  // extern "C"
