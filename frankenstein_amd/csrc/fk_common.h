// fk_common.h — shared device/host helpers for libfranken_hip.so (gfx950 / CDNA4 only).
//
// MFMA fragment abstraction used by the GEMM and attention kernels:
//   * one "k16 step" of a 32x32 output tile is   acc += A(32 x 16) * B(16 x 32)
//   * every lane (i = lane & 31, h = lane >> 5) holds an 8-element fragment of A row i and of
//     B column i; element e of half h sits in k-slot (h, e).  For bf16 that is the native operand
//     layout of v_mfma_f32_32x32x16_bf16 (k = 8h + e); for fp32 the step is issued as eight
//     v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain), instruction e consuming element e of both
//     halves.  Any assignment of real k indices to slots is valid as long as A and B agree.
//   * accumulator layout (dtype independent): acc[r] of lane l is
//         row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5),   col = l & 31.
//   * an accumulator tile X used directly as the B operand of the next product (sum over X's
//     ROW index): step s takes acc[8s .. 8s+7]; slot (h, e) is then X row
//         16 s + 8 (e >> 2) + 4 h + (e & 3)                       (see acc_row_of_slot()).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/franken_hip.h"

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));   // v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 operands
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define FK_DEV __device__ __forceinline__

// ------------------------------------------------------------------------------------------------
// host-side error plumbing (thread-local string, negative return codes; never throws)
// ------------------------------------------------------------------------------------------------
int fk_set_error(int code, const char* fmt, ...);
#define FK_CHECK_ARG(cond, ...)                                   \
  do {                                                            \
    if (!(cond)) return fk_set_error(FK_EINVAL, __VA_ARGS__);     \
  } while (0)
#define FK_CHECK_LAUNCH(name)                                                         \
  do {                                                                                \
    hipError_t e__ = hipGetLastError();                                               \
    if (e__ != hipSuccess) return fk_set_error(FK_ELAUNCH, "%s: %s", name, hipGetErrorString(e__)); \
  } while (0)

static inline int64_t fk_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ------------------------------------------------------------------------------------------------
// scalar conversions
// ------------------------------------------------------------------------------------------------
template <typename T> FK_DEV float to_f32(T v);
template <> FK_DEV float to_f32<float>(float v) { return v; }
template <> FK_DEV float to_f32<bf16_t>(bf16_t v) { return (float)v; }
template <typename T> FK_DEV T from_f32(float v);
template <> FK_DEV float from_f32<float>(float v) { return v; }
template <> FK_DEV bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }  // RNE, NaN preserving (v_cvt_pk_bf16_f32)

// 16-byte vector of T (8 bf16 or 4 fp32)
template <typename T> struct Vec16;
template <> struct Vec16<bf16_t> { static constexpr int N = 8; };
template <> struct Vec16<float> { static constexpr int N = 4; };

// ------------------------------------------------------------------------------------------------
// MFMA fragment (8 k-slots per lane)
// ------------------------------------------------------------------------------------------------
template <typename T> struct Frag;
template <> struct Frag<bf16_t> { bf16x8 v; };
template <> struct Frag<float> { float v[8]; };

template <typename T> FK_DEV void frag_zero(Frag<T>& f);
template <> FK_DEV void frag_zero<bf16_t>(Frag<bf16_t>& f) {
#pragma unroll
  for (int e = 0; e < 8; ++e) f.v[e] = (bf16_t)0.0f;
}
template <> FK_DEV void frag_zero<float>(Frag<float>& f) {
#pragma unroll
  for (int e = 0; e < 8; ++e) f.v[e] = 0.0f;
}

template <typename T> FK_DEV void frag_set(Frag<T>& f, int e, float x);
template <> FK_DEV void frag_set<bf16_t>(Frag<bf16_t>& f, int e, float x) { f.v[e] = (bf16_t)x; }
template <> FK_DEV void frag_set<float>(Frag<float>& f, int e, float x) { f.v[e] = x; }

// acc(32x32) += A(32x16) * B(16x32)
template <typename T> FK_DEV void mma32(f32x16& acc, const Frag<T>& a, const Frag<T>& b);
template <> FK_DEV void mma32<bf16_t>(f32x16& acc, const Frag<bf16_t>& a, const Frag<bf16_t>& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, acc, 0, 0, 0);
}
template <> FK_DEV void mma32<float>(f32x16& acc, const Frag<float>& a, const Frag<float>& b) {
#pragma unroll
  for (int e = 0; e < 8; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[e], b.v[e], acc, 0, 0, 0);
}

// accumulator element r of lane half h  ->  tile row
FK_DEV int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
// k-slot (h, e) of step s when an accumulator tile is reused as the B operand -> tile row
FK_DEV int acc_row_of_slot(int s, int h, int e) { return 16 * s + 8 * (e >> 2) + 4 * h + (e & 3); }

// accumulator registers 8s..8s+7 (scaled) -> B-operand fragment of step s
template <typename T> FK_DEV void frag_from_acc(Frag<T>& f, const f32x16& x, int s) {
#pragma unroll
  for (int e = 0; e < 8; ++e) frag_set<T>(f, e, x[8 * s + e]);
}

// ------------------------------------------------------------------------------------------------
// fragment loads
// ------------------------------------------------------------------------------------------------
// 8 contiguous elements at p (16-byte aligned for bf16, 32-byte for fp32)
template <typename T> FK_DEV void frag_load_contig(Frag<T>& f, const T* p);
template <> FK_DEV void frag_load_contig<bf16_t>(Frag<bf16_t>& f, const bf16_t* p) {
  f.v = *reinterpret_cast<const bf16x8*>(p);
}
template <> FK_DEV void frag_load_contig<float>(Frag<float>& f, const float* p) {
  f32x4 a = *reinterpret_cast<const f32x4*>(p);
  f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
  for (int e = 0; e < 4; ++e) { f.v[e] = a[e]; f.v[4 + e] = b[e]; }
}

// LDS transposed 4x16 block read (bf16): the 16 lanes of a group supply row addresses
// (lane 4q+p -> row q, columns 4p..4p+3) and lane i of the group receives column i of the 4 rows.
FK_DEV bf16x4 lds_read_tr4(const bf16_t* p) {
  s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4*)(__attribute__((address_space(3))) void*)p);
  return __builtin_bit_cast(bf16x4, r);
}

// ------------------------------------------------------------------------------------------------
// wave helpers
// ------------------------------------------------------------------------------------------------
FK_DEV float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
FK_DEV float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ------------------------------------------------------------------------------------------------
// dropout: a counter-based keep / drop decision per element (no mask is ever stored: the backward regenerates it).
//   bits(hi, lo) = mix32(mix32(mix32(hi ^ seed) + step * 0x85EBCA6B + site) ^ (lo * 0x9E3779B9)),   keep <=> bits >= p * 2^32
// (step and site pass through a mixer round of their own: XORed in behind the first round they made the masks of neighbouring sites
// and steps index-shifted copies of one sequence, lo -> lo +- const).
// mix32 = the "lowbias32" integer finaliser.  seed / step live in DEVICE memory (seed_ptr[0..1]; the step word is advanced by the host
// framework once per forward, inside a captured graph too), `site` numbers the dropout applications of one forward.  Elementwise: hi =
// index >> 32, lo = index; attention probabilities: hi = (b * H + h) * Nq + q, lo = key.  This is the library's own stream, restated in
// tests/dropout_ref.py; it does not reproduce torch's Philox stream (the reference's CPU and CUDA dropout streams differ from each other too).
FK_DEV unsigned fk_mix32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
struct DropKey { unsigned seed, salt, thresh; };
FK_DEV DropKey drop_key(const unsigned* seed_ptr, unsigned site, unsigned thresh) {
  return DropKey{seed_ptr[0], seed_ptr[1] * 0x85EBCA6Bu + site, thresh};
}
FK_DEV unsigned drop_row(const DropKey& k, unsigned hi) { return fk_mix32(fk_mix32(hi ^ k.seed) + k.salt); }
FK_DEV bool drop_keep(const DropKey& k, unsigned row, unsigned lo) { return fk_mix32(row ^ (lo * 0x9E3779B9u)) >= k.thresh; }

// Streaming output stores.  A line written once and not read again by the same launch can be stored non-temporal (global_store ... nt:
// the L2 treats it as a streaming request), so that it does not push the operand panels the launch keeps re-reading out of the XCD's
// 4-MiB L2 (with plain stores the up-projection fetched its 151-MB activation operand 6 times).
template <bool NT, typename V> FK_DEV void fk_st(V* q, const V& v) {
  if constexpr (NT) __builtin_nontemporal_store(v, q);
  else *q = v;
}

// XCD-aware block remap (bijective for any grid size): blocks b and b+8 share an XCD (round-robin
// dispatch), so give each XCD a contiguous chunk of logical tile ids -> neighbouring tiles share L2.
FK_DEV unsigned xcd_remap(unsigned bid, unsigned nwg) {
  const unsigned q = nwg >> 3, r = nwg & 7u, x = bid & 7u, i = bid >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}
