// mlp_fused.hip — the SwiGLU MLP's data-gradient chain as ONE attention-shaped kernel (gfx950 only).
//
//   reference call sites: models/brainformer.py:119-124 (MLP.forward: w2(silu(w1 x) * w3 x)), autograd of it.
//   replaces, in the backward of the fused SwiGLU path (engine.MlpBranch.backward):
//       dh13 = fk_gemm_nt_dswiglu(dy, W2T, h13)        (dg = dy W2T^T never stored; dh13 written: 1.2 GB per cfg2 layer)
//       dx   = fk_gemm_nt(dh13, W13T)                  (dh13 read back: 1.2 GB)
//   by one launch that keeps dh13 in registers between the two products (it is still written once: the weight gradients need it).
//
// The token is on the LANE, as the query is in the attention kernels: with C^T = A B (A: weight rows from LDS, B: activations from
// registers) an accumulator tile holds 32 output features x 32 tokens, lane = token.  Per workgroup of 4 waves = 128 tokens
// (one wave per SIMD; dx^T accumulators 12 tiles = 192 registers in the accumulator half, dy^T fragments 96 registers stationary) and
// per CHUNK of 32 hidden units (64 interleaved h1 | h3 columns):
//     dg^T [32 x 32]  = W2T_c [32 x 384] dy^T                      24 MFMAs        (W2T_c: six 32-row k-tile images in LDS)
//     dh1, dh3        = SwiGLU derivative against the saved h13 rows (one 16-byte load per lane and k16-step), rounded to bf16:
//                       the 8 values of a lane and step ARE the B fragment of the next product and the 16-byte dh13 store
//     dx^T [384 x 32] += W13T_c [384 x 64] dh13^T                  48 MFMAs        (W13T_c: one 384-row image in LDS)
// (Whole 128-token tiles take mlp_bwd_fused_asm_kernel further down: the same chain with its loop rotated by one product and the second
// product's step as a generated instruction stream.)
// The weight chunks (72 KB) arrive by LDS-DMA into a two-slot ring, chunk c + 1 requested when chunk c starts.  Products, operand
// slots and summation order are those of the two GEMM kernels this replaces (k in ascending 16-blocks), so dh13 and dx come out
// bit-identical to them (tests/test_kernels_gpu.py).
//
// Built WITHOUT -amdgpu-mfma-vgpr-form (frankenstein_amd/build.py): the 208 accumulator registers live in the AGPR half, which is what
// lets a wave keep 96 + 192 + 16 stationary registers at one wave per SIMD.
#include "fk_common.h"
#include <stdlib.h>

namespace {

#include "gemm_tile.h"

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

struct MlpBwdArgs {
  const bf16_t* dy; const bf16_t* w2t; const bf16_t* h13; const bf16_t* w13t;
  bf16_t* dh13; bf16_t* dx;
  int64_t lddy, ldw2t, ldh, ldw13t, lddh, lddx;
  int M, H;
};

#ifndef MF_SPREAD
#define MF_SPREAD 1        // 1: LDS-DMA requests placed one behind each MFMA group (-7 % per call, -0.44 ms per step); 0: in bursts
#endif
#ifndef MF_SPREAD_ST
#define MF_SPREAD_ST 0
#endif
#ifndef MF_NT_DH13
#define MF_NT_DH13 1
#endif
constexpr int MF_D = 384, MF_NW = 4, MF_TOK = MF_NW * 32;
constexpr int MF_KT = MF_D / 64;                          // k-tile images of a W2T chunk (6)
constexpr int MF_W2 = MF_KT * 32 * ROW_BYTES;            // 24 KiB
constexpr int MF_W13 = MF_D * ROW_BYTES;                 // 48 KiB
constexpr int MF_SLOT = MF_W2 + MF_W13;                  // 72 KiB
constexpr int MF_HREG = 32 * ROW_BYTES;                  // per wave: the 32 x 128-byte h13 / dh13 tile of a chunk (4 KiB)
constexpr int MF_LDS = 2 * MF_SLOT + MF_NW * MF_HREG;    // 160 KiB: the whole LDS of a CU
constexpr int MF_PIECES = MF_SLOT / 1024 / MF_NW;        // LDS-DMA wave instructions per wave and chunk (18)
constexpr int MF_ROWP = MF_D * 2 + 16;                   // padded bf16 row of the dx staging (784 B: 16-byte aligned, bank step 4)
static_assert(MF_LDS <= 160 * 1024 && MF_TOK * MF_ROWP <= MF_LDS, "LDS of a CU");
static_assert(MF_SLOT % (1024 * MF_NW) == 0, "whole pieces per wave");

template <bool FAST_SIGMOID> FK_DEV float mf_sigmoid(float x) {
  if constexpr (FAST_SIGMOID) return __builtin_amdgcn_rcpf(1.0f + __expf(-x));      // the bf16 mode's form in gemm.hip (sigmoid_f<bf16_t>)
  else return 1.0f / (1.0f + __expf(-x));
}

// One LDS-DMA request of 1 KiB (64 lanes x 16 B): wave-uniform 64-bit base + per-lane 32-bit byte offset -> LDS address `dst` (wave-uniform,
// through M0).  Issued as asm: the scalar-base form costs no address arithmetic per request (hipcc's builtin takes a per-lane 64-bit pointer:
// two or three VALU instructions per request on a wave that has no partner to hide them), and hipcc's wait-count pass does not see it
// (the waits are counted by hand below).  M0 is written one instruction ahead of its use.
FK_DEV void mf_dma(const void* base, unsigned voff, unsigned dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(dst) : "memory");
}

#ifdef MF_STAMP
// diagnostic build only (tools/build_variant.py mf_stamp mlp_fused.hip -DMF_STAMP; tools/stamp_mlp.py): cycles per segment of a chunk, summed over
// waves.  s_memtime returns through lgkmcnt, so every stamp also drains the wave's LDS reads: coarse segments only.  Never in the product.
__device__ unsigned long long mf_stamp_acc[16];
#define MF_ST_DECL unsigned long long st_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long st_t = __builtin_amdgcn_s_memtime(); const unsigned long long st_t0 = st_t;
#define MF_ST(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[i] += t_ - st_t; st_t = t_; }
#define MF_ST_FLUSH if (lane == 0) { for (int i_ = 0; i_ < 10; ++i_) atomicAdd(&mf_stamp_acc[i_], st_acc[i_]); atomicAdd(&mf_stamp_acc[15], __builtin_amdgcn_s_memtime() - st_t0); }
#else
#define MF_ST_DECL
#define MF_ST(i)
#define MF_ST_FLUSH
#endif

__global__ __launch_bounds__(MF_NW * 64, 1) void mlp_bwd_fused_kernel(MlpBwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using T = bf16_t;
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = blockIdx.x * MF_TOK + wave * 32 + li;
  const bool m_ok = m < p.M;
  const int mc = m_ok ? m : p.M - 1;                      // rows past the end: clamped loads, no stores
  const int nchunks = p.H / 32;

  // ---- LDS-DMA of one weight chunk: pieces of 8 image rows x 128 B (1 KiB per wave instruction); every wave moves six pieces of the W2T
  //      part (k-tile images s6 = 0..5 of 32 rows each) and twelve of the 384-row W13T image; XOR swizzle on the SOURCE column.  The
  //      per-lane byte offsets inside a chunk are constants: chunk c starts at w2t + c * 32 rows / w13t + c * 64 columns (wave-uniform)
  const int row8 = lane >> 3, ch = lane & 7;
  unsigned off2[MF_W2 / 1024 / MF_NW], off13[MF_W13 / 1024 / MF_NW];
#pragma unroll
  for (int j = 0; j < MF_W2 / 1024 / MF_NW; ++j) {
    const int q = wave * (MF_W2 / 1024 / MF_NW) + j, s6 = q >> 2, r = (q & 3) * 8 + row8;
    off2[j] = (unsigned)((r * (int)p.ldw2t + s6 * 64 + ((ch ^ ((r >> 1) & 7)) << 3)) * 2);
  }
#pragma unroll
  for (int j = 0; j < MF_W13 / 1024 / MF_NW; ++j) {
    const int r = (wave * (MF_W13 / 1024 / MF_NW) + j) * 8 + row8;
    off13[j] = (unsigned)((r * (int)p.ldw13t + ((ch ^ ((r >> 1) & 7)) << 3)) * 2);
  }
  // LDS: W2T ring (two 24-KiB slots), W13T ring (two 48-KiB slots), the waves' h13 tiles
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_void_t*)smem;
  auto w2slot = [&](int i) -> char* { return smem + i * MF_W2; };
  auto w13slot = [&](int i) -> char* { return smem + 2 * MF_W2 + i * MF_W13; };
  // one request (piece j of this wave's share) at a time: inside the chunk loop they are placed one behind each MFMA group (MF_SPREAD), so
  // that a wave without a partner on its SIMD issues them in the shadow of its own matrix work instead of in bursts
  auto dma_w2 = [&](int c, int slot, int j) __attribute__((always_inline)) {
    const void* g2 = p.w2t + (int64_t)c * 32 * p.ldw2t;
    const unsigned d0 = __builtin_amdgcn_readfirstlane(lds0 + slot * MF_W2 + wave * (MF_W2 / MF_NW));
    mf_dma(g2, off2[j], d0 + j * 1024);
  };
  auto dma_w13 = [&](int c, int slot, int j) __attribute__((always_inline)) {
    const void* g13 = p.w13t + (int64_t)c * 64;
    const unsigned d0 = __builtin_amdgcn_readfirstlane(lds0 + 2 * MF_W2 + slot * MF_W13 + wave * (MF_W13 / MF_NW));
    mf_dma(g13, off13[j], d0 + j * 1024);
  };
  auto issue_w2 = [&](int c, int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < MF_W2 / 1024 / MF_NW; ++j) dma_w2(c, slot, j);
  };
  auto issue_w13 = [&](int c, int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < MF_W13 / 1024 / MF_NW; ++j) dma_w13(c, slot, j);
  };
  // ---- the wave's h13 rows of a chunk (32 tokens x 64 interleaved columns = 4 KiB) also arrive by LDS-DMA, into a wave-private tile
  //      with the image's swizzle; the lane reads its four 16-byte pieces from there, the dh13 pieces go back into the SAME places and
  //      leave as whole 128-byte row segments (eight lanes per row): per-lane row accesses (16-byte pieces of 64 different rows per
  //      instruction) cost this kernel 190 us in loads and 530 us in stores per cfg2 call
  char* hreg = smem + 2 * MF_SLOT + wave * MF_HREG;       // behind both weight rings
  const int m0w = blockIdx.x * MF_TOK + wave * 32;
  const bool wave_full = m0w + 32 <= p.M;
  unsigned hoff[4];
  T* hdst[4];
  bool hrow_ok[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = j * 8 + row8, mr = m0w + r;
    hrow_ok[j] = mr < p.M;
    hoff[j] = (unsigned)(((int64_t)(hrow_ok[j] ? mr : p.M - 1) * p.ldh + ((ch ^ ((r >> 1) & 7)) << 3)) * 2);      // DMA source: swizzled column piece (the launcher checks 32 bits)
    hdst[j] = p.dh13 + (int64_t)(hrow_ok[j] ? mr : p.M - 1) * p.lddh + (ch << 3);                                 // store: logical piece ch of row r
  }
  const unsigned hreg_lds = __builtin_amdgcn_readfirstlane(lds0 + 2 * MF_SLOT + wave * MF_HREG);
  auto dma_h = [&](int c, int j) __attribute__((always_inline)) { mf_dma(p.h13 + (int64_t)c * 64, hoff[j], hreg_lds + j * 1024); };
  auto issue_h = [&](int c) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 4; ++j) dma_h(c, j);
  };

  // ---- Request stream and waits.  Loads retire in order, so "s_waitcnt vmcnt(N)" with N = the number of YOUNGER loads of this wave says
  //      that a given request has landed (outstanding stores only make it wait longer, never less).  Per chunk c, per wave:
  //        B1(c): wait W2T(c), barrier   | request W13T(c + 1) [12]  | dg^T = W2T_c dy^T        | wait h13 tile(c) | SwiGLU', dh13 out [4 stores]
  //        B2(c): wait W13T(c), barrier  | request W2T(c + 2) [6], h13 tile(c + 1) [4]          | dx^T += W13T_c dh13^T
  //      so W2T is requested 1.6 chunks ahead of its use, W13T 1.4, the h13 tile one; the younger-load counts are
  //        at B1(c): tile(c-1) 4 + W13T(c) 12 + W2T(c+1) 6 + tile(c) 4 = 26;   before the SwiGLU': W13T(c+1) = 12;
  //        at B2(c): W2T(c+1) 6 + tile(c) 4 + W13T(c+1) 12 = 22.
  //      Past the last chunk the requests are repeated with the last chunk's addresses (into slots nobody reads any more), so that the
  //      counts hold to the end.
  const int last = nchunks - 1;
  issue_w2(0, 0);
  issue_w13(0, 0);
  issue_w2(last < 1 ? last : 1, 1);
  issue_h(0);
  // ---- stationary B operand of the first product: dy^T fragments of this lane's token (k = 16 t + 8 lh .. + 8)
  Frag<T> dyf[MF_D / 16];
  const T* dyrow = p.dy + (int64_t)mc * p.lddy + 8 * lh;
#pragma unroll
  for (int t = 0; t < MF_D / 16; ++t) frag_load_contig<T>(dyf[t], dyrow + 16 * t);
  f32x16 dx[MF_D / 32];
#pragma unroll
  for (int t = 0; t < MF_D / 32; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) dx[t][r] = 0.0f;
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");     // B1(0): everything requested so far (and the dy fragments) has landed

  for (int c = 0; c < nchunks; ++c) {
    const char* w2img = w2slot(c & 1);
    const char* w13img = w13slot(c & 1);
    if (c > 0) asm volatile("s_waitcnt vmcnt(26)\n\ts_barrier" ::: "memory");              // B1(c)
    const int cn1 = c + 1 < nchunks ? c + 1 : last, cn2 = c + 2 < nchunks ? c + 2 : last;
#if !defined(MF_ABL_NODMA) && !MF_SPREAD                // MF_ABL_*: timing builds only (wrong results)
    issue_w13(cn1, (c + 1) & 1);                          // that slot was last read by the second product of chunk c - 1
#endif
    // ---- dg^T = W2T_c dy^T: 24 MFMAs, the A fragments read four ahead (one wave per SIMD: nobody else hides the LDS latency)
    f32x16 dg;
#pragma unroll
    for (int r = 0; r < 16; ++r) dg[r] = 0.0f;
    Frag<T> fa[2][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) nt_frag<T>(fa[0][s], w2img, li, s, lh);
#pragma unroll
    for (int s6 = 0; s6 < MF_KT; ++s6) {
      if (s6 + 1 < MF_KT) {
#pragma unroll
        for (int s = 0; s < 4; ++s) nt_frag<T>(fa[(s6 + 1) & 1][s], w2img + (s6 + 1) * 32 * ROW_BYTES, li, s, lh);
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) mma32<T>(dg, fa[s6 & 1][s], dyf[s6 * 4 + s]);
#if !defined(MF_ABL_NODMA) && MF_SPREAD
      dma_w13(cn1, (c + 1) & 1, 2 * s6);                  // W13T(c + 1): two of this wave's twelve requests behind each of the six groups
      dma_w13(cn1, (c + 1) & 1, 2 * s6 + 1);
#endif
      // the next group's four LDS reads FIRST, then this group's four MFMAs (left alone hipcc puts the reads behind three of the MFMAs
      // and waits for them 32 cycles later)
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
#ifndef MF_ABL_NOLD
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");       // this wave's h13 tile of chunk c has landed (wave-private: no barrier)
#endif
    // ---- SwiGLU derivative: lane (token, lh) holds dg of the hidden units 4 (2 s + lh) + e, e = 0..3, in dg[4 s + e]
    Frag<T> bf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const bf16x8 hv = *reinterpret_cast<const bf16x8*>(hreg + nt_off(li, 2 * s + lh));
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float a1 = (float)hv[e], a3 = (float)hv[4 + e], g = dg[4 * s + e];
#ifdef MF_ABL_NOMATH
        bf[s].v[e] = (T)(g + a1);
        bf[s].v[4 + e] = (T)(g + a3);
#else
        const float sg = mf_sigmoid<true>(a1), ds = g * sg;
        bf[s].v[e] = (T)(ds * a3 * (1.0f + a1 * (1.0f - sg)));
        bf[s].v[4 + e] = (T)(ds * a1);
#endif
      }
      *reinterpret_cast<bf16x8*>(hreg + nt_off(li, 2 * s + lh)) = bf[s].v;       // the place this lane just read
    }
    // dh13 leaves as whole row segments (eight lanes per row): read back here (LDS instructions of a wave execute in order: these reads see
    // the writes above), stored behind the barrier so that the LDS latency hides in the wait
    bf16x8 rb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) rb[j] = *reinterpret_cast<const bf16x8*>(hreg + nt_off(j * 8 + row8, ch));
    asm volatile("s_waitcnt vmcnt(22)\n\ts_barrier" ::: "memory");               // B2(c): W13T(c) has landed; everyone is done with W2T(c)
#if !defined(MF_ABL_NODMA) && !MF_SPREAD
    issue_w2(cn2, c & 1);
#endif
#ifndef MF_ABL_NOST
    if (wave_full) {                                      // wave-uniform: no per-lane predicate (and no branch per store) for whole tiles
#if !MF_SPREAD_ST
#pragma unroll
      for (int j = 0; j < 4; ++j) fk_st<MF_NT_DH13 != 0>(reinterpret_cast<bf16x8*>(hdst[j] + c * 64), rb[j]);      // written once, read by the weight-gradient GEMMs later
#else
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rb[0]), "+v"(rb[1]), "+v"(rb[2]), "+v"(rb[3])::"memory");        // the read-back is in registers: the tile is free
#endif
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (hrow_ok[j]) fk_st<MF_NT_DH13 != 0>(reinterpret_cast<bf16x8*>(hdst[j] + c * 64), rb[j]);
    }
#else
    asm volatile("" ::"v"(rb[0]), "v"(rb[1]), "v"(rb[2]), "v"(rb[3]));
#endif
#if !defined(MF_ABL_NOLD) && !MF_SPREAD
    issue_h(cn1);                                         // the tile is free: its read-back sits in registers (the stores above waited for it)
#endif
    // ---- dx^T += W13T_c dh13^T: 48 MFMAs, fragments of feature tile t + 1 read while tile t multiplies
    constexpr int NT12 = MF_D / 32;
#pragma unroll
    for (int s = 0; s < 4; ++s) nt_frag<T>(fa[0][s], w13img, li, s, lh);
#pragma unroll
    for (int t = 0; t < NT12; ++t) {
      if (t + 1 < NT12) {
#pragma unroll
        for (int s = 0; s < 4; ++s) nt_frag<T>(fa[(t + 1) & 1][s], w13img, 32 * (t + 1) + li, s, lh);
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) mma32<T>(dx[t], fa[t & 1][s], bf[s]);
#if MF_SPREAD                                             // W2T(c + 2) behind the first six groups, then the h13 tile of chunk c + 1 (same order as the bursts: the counts hold)
#ifndef MF_ABL_NODMA
      if (t < 6) dma_w2(cn2, c & 1, t);
#endif
#ifndef MF_ABL_NOLD
      if (t >= 6 && t < 10) dma_h(cn1, t - 6);
#endif
#if MF_SPREAD_ST && !defined(MF_ABL_NOST)
      if (wave_full && t >= 6 && t < 10) fk_st<MF_NT_DH13 != 0>(reinterpret_cast<bf16x8*>(hdst[t - 6] + c * 64), rb[t - 6]);
#endif
#endif
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");     // the repeated requests of the tail have landed: the LDS is free for the dx staging

  // ---- dx: accumulators (lane = token, 4 consecutive features per register group) -> bf16 rows staged in LDS -> 16-byte row stores
  char* stg = smem + wave * 32 * MF_ROWP;
#pragma unroll
  for (int t = 0; t < MF_D / 32; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bf16x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (T)dx[t][4 * g + e];
      *reinterpret_cast<bf16x4*>(stg + li * MF_ROWP + (32 * t + 8 * g + 4 * lh) * 2) = v;
    }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // wave-private staging: no barrier needed
  if (lane < MF_D / 8) {
    const int m0 = blockIdx.x * MF_TOK + wave * 32;
    for (int r = 0; r < 32; ++r) {
      if (m0 + r >= p.M) break;
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(stg + r * MF_ROWP + lane * 16);
      fk_st<true>(reinterpret_cast<bf16x8*>(p.dx + (int64_t)(m0 + r) * p.lddx + lane * 8), v);
    }
  }
}


// ================================================================================================================================
// The same chain with its loop rotated by one product and the second product's step as a GENERATED instruction stream (round 4).
//   A wave of mlp_bwd_fused_kernel is alone on its SIMD, so everything it issues shares one in-order stream, and the stamps
//   (tools/stamp_mlp.py, profiles/r04_mf_stamps*.txt) fit one model: a gap behind an MFMA costs max(32, 8 + the issue time of what is
//   placed in it) cycles.  The kernel above spends ~1900 cycles per chunk in the SwiGLU derivative with the matrix pipe idle (32 % of a
//   wave's life); hipcc asked to interleave puts 8-9 VALU with two transcendentals into 32 of the 48 gaps of the second product (64
//   cycles each) and leaves the rest at 32: no gain.  Here iteration c runs
//        A(c): dg^T(c + 1) = W2T_{c+1} dy^T                      24 MFMAs (hipcc), the W13T(c + 1) requests behind its groups
//        B(c): dx^T += W13T_c dh13^T(c)  BESIDE  the SwiGLU derivative of chunk c + 1, its dh13 tile round trip and stores, the W2T(c + 3)
//              and h13 tile(c + 2) requests, the barrier         48 MFMAs: mlpb_step_asm (tools/gen/gen_mlpb_asm.py -> mlp_bwd_asm.inc),
//              every gap filled to the same issue budget
//   Same products and summation order, same arithmetic instruction for instruction: dh13 and dx are the bits of the kernel above.
//   Whole 128-token tiles only (M % 128 == 0); the launcher keeps the kernel above for the rest (and FK_MLP_BWD_ASM=0 for everything).
//
//   Requests and waits (loads retire in order; N = the younger loads of this wave).  Issue order per wave:
//        A(c): W13T(c + 1) [12]      B(c): W2T(c + 3) [6], dh13(c + 1) stores, h13 tile(c + 2) [4]      A(c + 1): W13T(c + 2) [12] ...
//   * barrier B(c), in front of the last MFMA group of A(c): needs W13T(c) and this wave's tile(c + 1); the youngest loads are the 10
//     W13T(c + 1) requests issued so far -> vmcnt(10); everything older has landed then, W2T(c + 2) included.  Behind it the W2T slot of
//     chunk c + 1 is free (-> W2T(c + 3)), and the caller reads the stream's first fragments and its h13 pieces under the last MFMAs.
//   * barrier A(c + 1), inside the stream in front of its last four MFMAs: nothing to wait for but this wave's own fragment reads; behind
//     it the W13T slot of chunk c is free for the W13T(c + 2) requests of A(c + 1), whose first fragments the stream reads on its way out.
typedef unsigned u32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
#ifndef FK_MLPB_INC
#define FK_MLPB_INC "mlp_bwd_asm.inc"
#endif
#include FK_MLPB_INC

__global__ __launch_bounds__(MF_NW * 64, 1) void mlp_bwd_fused_asm_kernel(MlpBwdArgs p) {
  MF_ST_DECL
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using T = bf16_t;
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = blockIdx.x * MF_TOK + wave * 32 + li;       // < M: whole tiles only
  const int nchunks = p.H / 32, last = nchunks - 1;
  auto cl = [&](int k) { return k < last ? k : last; };
  const int row8 = lane >> 3, ch = lane & 7;
  unsigned off2[MF_W2 / 1024 / MF_NW], off13[MF_W13 / 1024 / MF_NW];
#pragma unroll
  for (int j = 0; j < MF_W2 / 1024 / MF_NW; ++j) {
    const int q = wave * (MF_W2 / 1024 / MF_NW) + j, s6 = q >> 2, r = (q & 3) * 8 + row8;
    off2[j] = (unsigned)((r * (int)p.ldw2t + s6 * 64 + ((ch ^ ((r >> 1) & 7)) << 3)) * 2);
  }
#pragma unroll
  for (int j = 0; j < MF_W13 / 1024 / MF_NW; ++j) {
    const int r = (wave * (MF_W13 / 1024 / MF_NW) + j) * 8 + row8;
    off13[j] = (unsigned)((r * (int)p.ldw13t + ((ch ^ ((r >> 1) & 7)) << 3)) * 2);
  }
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_void_t*)smem;
  auto w2slot = [&](int i) -> char* { return smem + i * MF_W2; };
  auto w13slot = [&](int i) -> char* { return smem + 2 * MF_W2 + i * MF_W13; };
  auto dma_w2 = [&](int c, int slot, int j) __attribute__((always_inline)) {
    mf_dma(p.w2t + (int64_t)c * 32 * p.ldw2t, off2[j], __builtin_amdgcn_readfirstlane(lds0 + slot * MF_W2 + wave * (MF_W2 / MF_NW)) + j * 1024);
  };
  auto dma_w13 = [&](int c, int slot, int j) __attribute__((always_inline)) {
    mf_dma(p.w13t + (int64_t)c * 64, off13[j], __builtin_amdgcn_readfirstlane(lds0 + 2 * MF_W2 + slot * MF_W13 + wave * (MF_W13 / MF_NW)) + j * 1024);
  };
  char* hreg = smem + 2 * MF_SLOT + wave * MF_HREG;
  const int m0w = blockIdx.x * MF_TOK + wave * 32;
  const unsigned hreg_lds = __builtin_amdgcn_readfirstlane(lds0 + 2 * MF_SLOT + wave * MF_HREG);
  // the stream's address operands: LDS byte addresses (adr) and global byte offsets (ofs); adr[0..3] and adr[12..15] follow the ring slots
  u32x16 adr, ofs;
  unsigned ax[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    ax[s] = (unsigned)nt_off(li, 2 * s + lh);
    adr[4 + s] = hreg_lds + ax[s];                                          // AH: the lane's pieces of its tile
    adr[8 + s] = hreg_lds + (unsigned)nt_off(s * 8 + row8, ch);             // AR: tile rows, eight lanes per row
  }
#pragma unroll
  for (int j = 0; j < 6; ++j) ofs[j] = off2[j];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = j * 8 + row8;
    ofs[6 + j] = (unsigned)(((int64_t)(m0w + r) * p.ldh + ((ch ^ ((r >> 1) & 7)) << 3)) * 2);     // HOFF: tile request pieces (swizzled source column)
    ofs[10 + j] = (unsigned)(((int64_t)(m0w + r) * p.lddh + (ch << 3)) * 2);                       // VST: row stores (logical piece ch of row r)
  }
  ofs[14] = 0;
  ofs[15] = 0;

  // ---- prologue: W2T(0), W13T(0), W2T(1), tile(0); dy^T fragments; chunk 0's first product and SwiGLU derivative alone
#pragma unroll
  for (int j = 0; j < MF_W2 / 1024 / MF_NW; ++j) dma_w2(0, 0, j);
#pragma unroll
  for (int j = 0; j < MF_W13 / 1024 / MF_NW; ++j) dma_w13(0, 0, j);
#pragma unroll
  for (int j = 0; j < MF_W2 / 1024 / MF_NW; ++j) dma_w2(cl(1), 1, j);
#pragma unroll
  for (int j = 0; j < 4; ++j) mf_dma(p.h13, ofs[6 + j], hreg_lds + j * 1024);
  Frag<T> dyf[MF_D / 16];
  const T* dyrow = p.dy + (int64_t)m * p.lddy + 8 * lh;
#pragma unroll
  for (int t = 0; t < MF_D / 16; ++t) frag_load_contig<T>(dyf[t], dyrow + 16 * t);
  f32x16 dx[MF_D / 32];
#pragma unroll
  for (int t = 0; t < MF_D / 32; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) dx[t][r] = 0.0f;
  MF_ST(4)                                                  // (stamp build: requests issued, dy fragments requested)
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  MF_ST(5)                                                  // (stamp build: the first wait)
  Frag<T> fa[2][4];
  f32x16 dg;
  u32x16 fa0, bfc, bfn, hvv;
#pragma unroll
  for (int r = 0; r < 16; ++r) dg[r] = 0.0f;
#pragma unroll
  for (int s = 0; s < 4; ++s) nt_frag<T>(fa[0][s], w2slot(0), li, s, lh);
#pragma unroll
  for (int s6 = 0; s6 < MF_KT; ++s6) {
    if (s6 + 1 < MF_KT) {
#pragma unroll
      for (int s = 0; s < 4; ++s) nt_frag<T>(fa[(s6 + 1) & 1][s], w2slot(0) + (s6 + 1) * 32 * ROW_BYTES, li, s, lh);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) mma32<T>(dg, fa[s6 & 1][s], dyf[s6 * 4 + s]);
  }
  {
    bf16x8 rb[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const bf16x8 hv = *reinterpret_cast<const bf16x8*>(hreg + nt_off(li, 2 * s + lh));
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float a1 = (float)hv[e], a3 = (float)hv[4 + e], g = dg[4 * s + e];
        const float sg = mf_sigmoid<true>(a1), ds = g * sg;
        o[e] = (T)(ds * a3 * (1.0f + a1 * (1.0f - sg)));
        o[4 + e] = (T)(ds * a1);
      }
      *reinterpret_cast<bf16x8*>(hreg + nt_off(li, 2 * s + lh)) = o;
      const u32x4v w = __builtin_bit_cast(u32x4v, o);
#pragma unroll
      for (int k = 0; k < 4; ++k) bfc[4 * s + k] = w[k];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) rb[j] = *reinterpret_cast<const bf16x8*>(hreg + nt_off(j * 8 + row8, ch));
#pragma unroll
    for (int j = 0; j < 4; ++j)
      fk_st<true>(reinterpret_cast<bf16x8*>(p.dh13 + (int64_t)(m0w + j * 8 + row8) * p.lddh + (ch << 3)), rb[j]);
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");    // everyone is done with W2T(0); this wave's tile rows are in registers
#pragma unroll
  for (int j = 0; j < MF_W2 / 1024 / MF_NW; ++j) dma_w2(cl(2), 0, j);
#pragma unroll
  for (int j = 0; j < 4; ++j) mf_dma(p.h13 + (int64_t)cl(1) * 64, ofs[6 + j], hreg_lds + j * 1024);
#pragma unroll
  for (int s = 0; s < 4; ++s) {                                       // A(0)'s first fragments: W2T(1) landed in front of the first barrier
    const u32x4v w = *reinterpret_cast<const u32x4v*>(w2slot(1) + ax[s]);
#pragma unroll
    for (int k = 0; k < 4; ++k) fa0[4 * s + k] = w[k];
  }

  MF_ST(0)                                                  // prologue
  for (int c = 0; c < last; ++c) {
    const char* w2img = w2slot((c + 1) & 1);
    const char* w13img = w13slot(c & 1);
    // ---- A(c): dg^T(c + 1), 24 MFMAs, fragments one group ahead; the last group behind barrier B(c)
#pragma unroll
    for (int r = 0; r < 16; ++r) dg[r] = 0.0f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const u32x4v w = {fa0[4 * s], fa0[4 * s + 1], fa0[4 * s + 2], fa0[4 * s + 3]};
      fa[0][s].v = __builtin_bit_cast(bf16x8, w);
    }
#pragma unroll
    for (int s6 = 0; s6 < MF_KT; ++s6) {
      if (s6 + 1 < MF_KT) {
#pragma unroll
        for (int s = 0; s < 4; ++s) nt_frag<T>(fa[(s6 + 1) & 1][s], w2img + (s6 + 1) * 32 * ROW_BYTES, li, s, lh);
      } else {
        __builtin_amdgcn_sched_barrier(0);
        MF_ST(1)                                            // first product, groups 0-4
        asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)\n\ts_barrier" ::: "memory");        // barrier B(c)
        MF_ST(2)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const u32x4v w = *reinterpret_cast<const u32x4v*>(w13img + ax[s]);             // the stream's first fragments (feature tile 0) ...
          const u32x4v h = *reinterpret_cast<const u32x4v*>(hreg + ax[s]);               // ... and this wave's h13 pieces of chunk c + 1
#pragma unroll
          for (int k = 0; k < 4; ++k) { fa0[4 * s + k] = w[k]; hvv[4 * s + k] = h[k]; }
        }
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) mma32<T>(dg, fa[s6 & 1][s], dyf[s6 * 4 + s]);
      dma_w13(cl(c + 1), (c + 1) & 1, 2 * s6);
      dma_w13(cl(c + 1), (c + 1) & 1, 2 * s6 + 1);
      if (s6 + 1 < MF_KT) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
      else __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    MF_ST(3)                                                // group 5
    // ---- B(c): the generated step
    const unsigned w13a = lds0 + 2 * MF_W2 + (c & 1) * MF_W13, w2a = lds0 + (c & 1) * MF_W2;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      adr[s] = w13a + ax[s];
      adr[12 + s] = w2a + ax[s];
    }
    mlpb_step_asm(dx, dg, fa0, bfc, bfn, hvv, adr, ofs, p.w2t + (int64_t)cl(c + 3) * 32 * p.ldw2t, p.h13 + (int64_t)cl(c + 2) * 64,
                  p.dh13 + (int64_t)(c + 1) * 64, __builtin_amdgcn_readfirstlane(lds0 + ((c + 1) & 1) * MF_W2 + wave * (MF_W2 / MF_NW)), hreg_lds);
    bfc = bfn;
    MF_ST(6)                                                // the generated step (barrier A inside)
  }
  // ---- the last chunk's second product, alone
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");     // W13T(last) and the repeated requests of the tail have landed
  MF_ST(7)
  {
    const char* w13img = w13slot(last & 1);
    constexpr int NT12 = MF_D / 32;
    Frag<T> bf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const u32x4v w = {bfc[4 * s], bfc[4 * s + 1], bfc[4 * s + 2], bfc[4 * s + 3]};
      bf[s].v = __builtin_bit_cast(bf16x8, w);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) nt_frag<T>(fa[0][s], w13img, li, s, lh);
#pragma unroll
    for (int t = 0; t < NT12; ++t) {
      if (t + 1 < NT12) {
#pragma unroll
        for (int s = 0; s < 4; ++s) nt_frag<T>(fa[(t + 1) & 1][s], w13img, 32 * (t + 1) + li, s, lh);
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) mma32<T>(dx[t], fa[t & 1][s], bf[s]);
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  MF_ST(8)                                                  // the last chunk's second product
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // every wave has its last fragments: the LDS is free for the dx staging

  // ---- dx: accumulators -> bf16 rows staged in LDS -> 16-byte row stores (as in the kernel above)
  char* stg = smem + wave * 32 * MF_ROWP;
#pragma unroll
  for (int t = 0; t < MF_D / 32; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bf16x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (T)dx[t][4 * g + e];
      *reinterpret_cast<bf16x4*>(stg + li * MF_ROWP + (32 * t + 8 * g + 4 * lh) * 2) = v;
    }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (lane < MF_D / 8) {
    for (int r = 0; r < 32; ++r) {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(stg + r * MF_ROWP + lane * 16);
      fk_st<true>(reinterpret_cast<bf16x8*>(p.dx + (int64_t)(m0w + r) * p.lddx + lane * 8), v);
    }
  }
  MF_ST(9)                                                  // dx epilogue
  MF_ST_FLUSH
}


// ================================================================================================================================
// The forward up-projection + SwiGLU of the same MLP with the token on the lane (what fk_gemm_nt_swiglu computes: H13 and G).
//   Per wave 32 tokens with x^T stationary (96 registers); per chunk of 32 hidden units the W13 rows arrive as TWO 32-row tiles — tile 0
//   the h1 rows, tile 1 the h3 rows of the same units — so that a lane finds h1 and h3 of a unit in the same register index of its two
//   accumulators and silu(h1) * h3 happens in registers, without the fp32 staging sweep of the tiled kernel.  No accumulator outlives a
//   chunk, so a wave needs ~200 registers: two waves per SIMD, eight per workgroup (256 tokens), and the weight chunk (48 KiB) is shared by
//   twice as many tokens as in the backward kernel above.  h13 (16 bytes per lane and k-group) and g (8 bytes) leave through wave-private LDS
//   tiles as whole row segments; the stores of chunk c are issued at the head of chunk c + 1 (from registers), i.e. BEFORE the next weight
//   request, so that the end-of-chunk wait finds only requests that are a whole chunk old.  Same products and summation order as
//   gemm_nt_ring2_kernel<.., 1>: bit-identical H13 and G.
struct MlpUpArgs {
  const bf16_t* x; const bf16_t* w13; bf16_t* h13; bf16_t* g;
  int64_t ldx, ldw, ldh, ldg;
  int M, H;
};
constexpr int MU_NW = 8, MU_TOK = MU_NW * 32;
constexpr int MU_TILE = MF_KT * 32 * ROW_BYTES;          // one 32-row tile of a chunk: six k-tile images (24 KiB)
constexpr int MU_SLOT = 2 * MU_TILE;                     // h1 rows + h3 rows (48 KiB)
constexpr int MU_HT = 32 * ROW_BYTES, MU_GT = 32 * 64;   // per wave: h13 tile (32 x 128 B), g tile (32 x 64 B)
constexpr int MU_LDS = 2 * MU_SLOT + MU_NW * (MU_HT + MU_GT);
static_assert(MU_LDS <= 160 * 1024, "LDS of a CU");
constexpr int MU_PIECES = MU_SLOT / 1024 / MU_NW;        // 6 requests per wave and chunk

__global__ __launch_bounds__(MU_NW * 64, 2) void mlp_up_fused_kernel(MlpUpArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using T = bf16_t;
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0w = blockIdx.x * MU_TOK + wave * 32;
  const int m = m0w + li, mc = m < p.M ? m : p.M - 1;
  const bool wave_full = m0w + 32 <= p.M;
  const int nchunks = p.H / 32, last = nchunks - 1;
  const int row8 = lane >> 3, ch = lane & 7;
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_void_t*)smem;

  // requests: piece q = wave * 6 + j of the slot: tile q / 24 (0: h1 rows, 1: h3 rows), k-tile image (q % 24) >> 2, image rows (q & 3) * 8 + row8.
  // Image row n of a tile is hidden unit 32 c + n: interleaved W13 row 8 (n / 4) + n % 4 (+ 4 for h3) of the chunk's 64 rows.
  unsigned woff[MU_PIECES];
#pragma unroll
  for (int j = 0; j < MU_PIECES; ++j) {
    const int q = wave * MU_PIECES + j, tile = q / 24, s6 = (q % 24) >> 2, n = (q & 3) * 8 + row8;
    const int wrow = 8 * (n >> 2) + (n & 3) + 4 * tile;
    woff[j] = (unsigned)((wrow * (int)p.ldw + s6 * 64 + ((ch ^ ((n >> 1) & 7)) << 3)) * 2);
  }
  auto issue_w = [&](int c, int slot) __attribute__((always_inline)) {
    const void* gw = p.w13 + (int64_t)c * 64 * p.ldw;
    const unsigned d0 = __builtin_amdgcn_readfirstlane(lds0 + slot * MU_SLOT + wave * (MU_SLOT / MU_NW));
#pragma unroll
    for (int j = 0; j < MU_PIECES; ++j) mf_dma(gw, woff[j], d0 + j * 1024);
  };
  char* htile = smem + 2 * MU_SLOT + wave * (MU_HT + MU_GT);
  char* gtile = htile + MU_HT;
  // output rows of the read-back: h13 eight lanes per row (8 rows per instruction), g four lanes per row (16 rows per instruction)
  T* hdst[4];
  T* gdst[2];
  bool hok[4], gok[2];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int mr = m0w + j * 8 + row8;
    hok[j] = mr < p.M;
    hdst[j] = p.h13 + (int64_t)(hok[j] ? mr : p.M - 1) * p.ldh + (ch << 3);
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int mr = m0w + j * 16 + (lane >> 2);
    gok[j] = mr < p.M;
    gdst[j] = p.g + (int64_t)(gok[j] ? mr : p.M - 1) * p.ldg + ((lane & 3) << 3);
  }

  issue_w(0, 0);
  Frag<T> xf[MF_D / 16];
  const T* xrow = p.x + (int64_t)mc * p.ldx + 8 * lh;
#pragma unroll
  for (int t = 0; t < MF_D / 16; ++t) frag_load_contig<T>(xf[t], xrow + 16 * t);
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");     // chunk 0 and the x fragments have landed

  bf16x8 rbh[4] = {}, rbg[2] = {};
  for (int c = 0; c < nchunks; ++c) {
    const char* t0 = smem + (c & 1) * MU_SLOT;
    const char* t1 = t0 + MU_TILE;
    // the previous chunk's outputs (in registers since its read-back), then the next chunk's weights: the wait at the end of THIS chunk then
    // only covers requests issued a whole chunk earlier
    if (c > 0) {
      if (wave_full) {
#pragma unroll
        for (int j = 0; j < 4; ++j) fk_st<true>(reinterpret_cast<bf16x8*>(hdst[j] + (c - 1) * 64), rbh[j]);
#pragma unroll
        for (int j = 0; j < 2; ++j) fk_st<true>(reinterpret_cast<bf16x8*>(gdst[j] + (c - 1) * 32), rbg[j]);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (hok[j]) fk_st<true>(reinterpret_cast<bf16x8*>(hdst[j] + (c - 1) * 64), rbh[j]);
#pragma unroll
        for (int j = 0; j < 2; ++j)
          if (gok[j]) fk_st<true>(reinterpret_cast<bf16x8*>(gdst[j] + (c - 1) * 32), rbg[j]);
      }
    }
    issue_w(c + 1 < nchunks ? c + 1 : last, (c + 1) & 1);          // that slot was read in chunk c - 1: every wave is past the barrier behind it
    // ---- H^T = W13_c x^T: two accumulator tiles (h1 rows, h3 rows of the same 32 units), 48 MFMAs, fragments read one k-tile ahead
    f32x16 a1, a3;
#pragma unroll
    for (int r = 0; r < 16; ++r) { a1[r] = 0.0f; a3[r] = 0.0f; }
    Frag<T> f1[2][4], f3[2][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) { nt_frag<T>(f1[0][s], t0, li, s, lh); nt_frag<T>(f3[0][s], t1, li, s, lh); }
#pragma unroll
    for (int s6 = 0; s6 < MF_KT; ++s6) {
      if (s6 + 1 < MF_KT) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          nt_frag<T>(f1[(s6 + 1) & 1][s], t0 + (s6 + 1) * 32 * ROW_BYTES, li, s, lh);
          nt_frag<T>(f3[(s6 + 1) & 1][s], t1 + (s6 + 1) * 32 * ROW_BYTES, li, s, lh);
        }
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        mma32<T>(a1, f1[s6 & 1][s], xf[s6 * 4 + s]);
        mma32<T>(a3, f3[s6 & 1][s], xf[s6 * 4 + s]);
      }
      __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- SwiGLU in registers: accumulator register 4 g + i of lane (token, lh) is hidden unit 8 g + 4 lh + i of the chunk in BOTH tiles;
    //      the four units of a register group are interleaved hidden group 2 g + lh: one 16-byte piece of h13, one 8-byte piece of g
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      bf16x8 hp;
      bf16x4 gp;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float h1 = a1[4 * g4 + i], h3 = a3[4 * g4 + i];
        hp[i] = (T)h1;
        hp[4 + i] = (T)h3;
        gp[i] = (T)(h1 * mf_sigmoid<true>(h1) * h3);
      }
      const int q = 2 * g4 + lh;
      *reinterpret_cast<bf16x8*>(htile + nt_off(li, q)) = hp;
      *reinterpret_cast<bf16x4*>(gtile + li * 64 + ((((q >> 1) + li) & 3) << 4) + ((q & 1) << 3)) = gp;      // 16-byte chunks rotated by the row
    }
    // read back as whole row segments (LDS instructions of a wave execute in order); stored at the head of the next chunk
#pragma unroll
    for (int j = 0; j < 4; ++j) rbh[j] = *reinterpret_cast<const bf16x8*>(htile + nt_off(j * 8 + row8, ch));
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = j * 16 + (lane >> 2);
      rbg[j] = *reinterpret_cast<const bf16x8*>(gtile + r * 64 + ((((lane & 3) + r) & 3) << 4));
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");   // chunk c + 1 has landed (requested a chunk ago, like the stores in front of it); everyone is done with this slot
  }
  if (wave_full) {
#pragma unroll
    for (int j = 0; j < 4; ++j) fk_st<true>(reinterpret_cast<bf16x8*>(hdst[j] + last * 64), rbh[j]);
#pragma unroll
    for (int j = 0; j < 2; ++j) fk_st<true>(reinterpret_cast<bf16x8*>(gdst[j] + last * 32), rbg[j]);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (hok[j]) fk_st<true>(reinterpret_cast<bf16x8*>(hdst[j] + last * 64), rbh[j]);
#pragma unroll
    for (int j = 0; j < 2; ++j)
      if (gok[j]) fk_st<true>(reinterpret_cast<bf16x8*>(gdst[j] + last * 32), rbg[j]);
  }
}


// ================================================================================================================================
// The packed q | k | v projection with fused RoPE and pre-scaled queries (what fk_gemm_nt_rope computes) with the token on the lane.
//   Same frame as mlp_up_fused_kernel: x^T stationary, two waves per SIMD, a chunk = 64 output columns = ONE head (two 32-column
//   accumulator tiles), no accumulator outlives a chunk.  A lane owns 4 consecutive columns per register group = two complex pairs, so the
//   rotation is register arithmetic; its (cos, sin) pairs — 16 bytes per register group, 8 groups per chunk, from the lane's own token row
//   of the L2-resident table (the pre-scaled copy for the query heads) — are requested at the head of the chunk, in front of the next
//   weight request, and waited for with a counted vmcnt (loads retire in order: 6 younger requests).  Same products, summation order and
//   rotation arithmetic as gemm_nt_ring2_kernel<.., 3>: bit-identical output.
struct QkvArgs {
  const bf16_t* x; const bf16_t* w; bf16_t* out; const float* table;
  int64_t ldx, ldw, ldo, table_bs, q_off;
  int M, N, T, pos_off, rot_chunks, q_chunks;
};
constexpr int QK_LDS = 2 * MU_SLOT + MU_NW * MU_HT;
static_assert(QK_LDS <= 160 * 1024, "LDS of a CU");

__global__ __launch_bounds__(MU_NW * 64, 2) void qkv_rope_fused_kernel(QkvArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using T = bf16_t;
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0w = blockIdx.x * MU_TOK + wave * 32;
  const int m = m0w + li, mc = m < p.M ? m : p.M - 1;
  const bool wave_full = m0w + 32 <= p.M;
  const int nchunks = p.N / 64, last = nchunks - 1;
  const int row8 = lane >> 3, ch = lane & 7;
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_void_t*)smem;

  // requests: piece q = wave * 6 + j: tile q / 24 (columns 0..31 / 32..63 of the head), k-tile image (q % 24) >> 2, image rows (q & 3) * 8 + row8
  unsigned woff[MU_PIECES];
#pragma unroll
  for (int j = 0; j < MU_PIECES; ++j) {
    const int q = wave * MU_PIECES + j, tile = q / 24, s6 = (q % 24) >> 2, n = (q & 3) * 8 + row8;
    woff[j] = (unsigned)(((32 * tile + n) * (int)p.ldw + s6 * 64 + ((ch ^ ((n >> 1) & 7)) << 3)) * 2);
  }
  auto issue_w = [&](int c, int slot) __attribute__((always_inline)) {
    const void* gw = p.w + (int64_t)c * 64 * p.ldw;
    const unsigned d0 = __builtin_amdgcn_readfirstlane(lds0 + slot * MU_SLOT + wave * (MU_SLOT / MU_NW));
#pragma unroll
    for (int j = 0; j < MU_PIECES; ++j) mf_dma(gw, woff[j], d0 + j * 1024);
  };
  char* htile = smem + 2 * MU_SLOT + wave * MU_HT;
  T* odst[4];
  bool ook[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int mr = m0w + j * 8 + row8;
    ook[j] = mr < p.M;
    odst[j] = p.out + (int64_t)(ook[j] ? mr : p.M - 1) * p.ldo + (ch << 3);
  }
  // this lane's (cos, sin) row: token mc is position pos_off + mc % T of sample mc / T; the lane's pairs start at pair 2 lh (4 floats per piece)
  const float* trow = p.table + (int64_t)(mc / p.T) * p.table_bs + (int64_t)(p.pos_off + mc % p.T) * 64 + 4 * lh;

  issue_w(0, 0);
  Frag<T> xf[MF_D / 16];
  const T* xrow = p.x + (int64_t)mc * p.ldx + 8 * lh;
#pragma unroll
  for (int t = 0; t < MF_D / 16; ++t) frag_load_contig<T>(xf[t], xrow + 16 * t);
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");     // chunk 0 and the x fragments have landed

  bf16x8 rb[4] = {};
  for (int c = 0; c < nchunks; ++c) {
    const char* t0 = smem + (c & 1) * MU_SLOT;
    const char* t1 = t0 + MU_TILE;
    if (c > 0) {                                          // the previous head's rows (in registers since its read-back)
      if (wave_full) {
#pragma unroll
        for (int j = 0; j < 4; ++j) fk_st<true>(reinterpret_cast<bf16x8*>(odst[j] + (c - 1) * 64), rb[j]);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (ook[j]) fk_st<true>(reinterpret_cast<bf16x8*>(odst[j] + (c - 1) * 64), rb[j]);
      }
    }
    const bool rot = c < p.rot_chunks;                    // workgroup-uniform
    f32x4 cs[8];
    if (rot) {
      const float* tq = trow + (c < p.q_chunks ? p.q_off : (int64_t)0);
      asm volatile("global_load_dwordx4 %0, %8, off\n\tglobal_load_dwordx4 %1, %8, off offset:32\n\t"
                   "global_load_dwordx4 %2, %8, off offset:64\n\tglobal_load_dwordx4 %3, %8, off offset:96\n\t"
                   "global_load_dwordx4 %4, %8, off offset:128\n\tglobal_load_dwordx4 %5, %8, off offset:160\n\t"
                   "global_load_dwordx4 %6, %8, off offset:192\n\tglobal_load_dwordx4 %7, %8, off offset:224"
                   : "=&v"(cs[0]), "=&v"(cs[1]), "=&v"(cs[2]), "=&v"(cs[3]), "=&v"(cs[4]), "=&v"(cs[5]), "=&v"(cs[6]), "=&v"(cs[7])
                   : "v"(tq) : "memory");
    }
    issue_w(c + 1 < nchunks ? c + 1 : last, (c + 1) & 1);          // that slot was read in chunk c - 1: every wave is past the barrier behind it
    // ---- two accumulator tiles of the head: 48 MFMAs, four fragments read ahead (tile 0 and tile 1 of a k-tile alternate)
    f32x16 acc[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[0][r] = 0.0f; acc[1][r] = 0.0f; }
    Frag<T> f[2][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) nt_frag<T>(f[0][s], t0, li, s, lh);
#pragma unroll
    for (int gq = 0; gq < 2 * MF_KT; ++gq) {               // group gq: tile gq & 1 of k-tile gq >> 1
      if (gq + 1 < 2 * MF_KT) {
        const char* nx = ((gq + 1) & 1 ? t1 : t0) + ((gq + 1) >> 1) * 32 * ROW_BYTES;
#pragma unroll
        for (int s = 0; s < 4; ++s) nt_frag<T>(f[(gq + 1) & 1][s], nx, li, s, lh);
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) mma32<T>(acc[gq & 1], f[gq & 1][s], xf[(gq >> 1) * 4 + s]);
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (rot)      // the (cos, sin) pieces have landed: six younger requests (the next weight chunk) may still be in flight
      asm volatile("s_waitcnt vmcnt(6)" : "+v"(cs[0]), "+v"(cs[1]), "+v"(cs[2]), "+v"(cs[3]), "+v"(cs[4]), "+v"(cs[5]), "+v"(cs[6]), "+v"(cs[7])::"memory");
    // ---- rotation in registers (the arithmetic of nt_epilogue's fused RoPE: explicit fma shape), rounding, 8-byte pieces into the row tile
#pragma unroll
    for (int tl = 0; tl < 2; ++tl)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        float v[4] = {acc[tl][4 * g4], acc[tl][4 * g4 + 1], acc[tl][4 * g4 + 2], acc[tl][4 * g4 + 3]};
        if (rot) {
          const f32x4 t = cs[4 * tl + g4];                 // (c0, s0, c1, s1) of the pairs 16 tl + 4 g4 + 2 lh, + 1
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const float re = v[2 * j], im = v[2 * j + 1];
            v[2 * j] = __builtin_fmaf(re, t[2 * j], -(im * t[2 * j + 1]));
            v[2 * j + 1] = __builtin_fmaf(re, t[2 * j + 1], im * t[2 * j]);
          }
        }
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (T)v[e];
        *reinterpret_cast<bf16x4*>(htile + nt_off(li, 4 * tl + g4) + 8 * lh) = o;
      }
#pragma unroll
    for (int j = 0; j < 4; ++j) rb[j] = *reinterpret_cast<const bf16x8*>(htile + nt_off(j * 8 + row8, ch));
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");   // chunk c + 1 has landed (requested a chunk ago); everyone is done with this slot
  }
  if (wave_full) {
#pragma unroll
    for (int j = 0; j < 4; ++j) fk_st<true>(reinterpret_cast<bf16x8*>(odst[j] + last * 64), rb[j]);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (ook[j]) fk_st<true>(reinterpret_cast<bf16x8*>(odst[j] + last * 64), rb[j]);
  }
}

}  // namespace

// the token-on-the-lane form of fk_gemm_nt_rope (called from gemm.hip's entry point)
// One 8-wave workgroup of 256 tokens per CU at a time: the launch runs in rounds of 256 workgroups, so its time is a step function of M
// (150 / 293 / 460 us at <= 65 536 / 131 072 / 196 608 rows for the up-projection) where the tiled kernels' is linear.  Measured over
// M = 19 200 ... 196 608 (tools/mlp_size_sweep.py, profiles/r04_mlp_size_sweep.txt): these kernels win wherever the last round is at least
// ~40 % full, i.e. the grid fills >= 70 % of its rounds (49 152: 0.75 wins; 38 400 and 76 800: 0.59 lose by 5-12 %).
static bool mu_grid_fills(int64_t M) {
  const int64_t wg = fk_cdiv(M, MU_TOK), rounds = fk_cdiv(wg, 256);
  return wg * 10 >= rounds * 256 * 7;
}
bool fk_qkv_rope_fused_ok(int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, const void* bias, int64_t T, int64_t D, int64_t rot_cols,
                          int64_t q_cols, int dtype) {
  static const bool off = getenv("FK_QKV_FUSED") != nullptr && getenv("FK_QKV_FUSED")[0] == '0';
  return !off && dtype == FK_BF16 && K == MF_D && D == 64 && N % 64 == 0 && rot_cols % 64 == 0 && q_cols % 64 == 0 && rot_cols <= N && !bias && mu_grid_fills(M) &&
         M < (1LL << 31) && T > 0 && M % T == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0 && 64 * ldb * 2 < (1LL << 32);
}
int fk_qkv_rope_fused_launch(const void* A, int64_t lda, const void* W, int64_t ldb, void* C, int64_t ldc, int64_t M, int64_t N, const float* table,
                             int64_t table_bs, int64_t T, int64_t pos_off, int64_t rot_cols, int64_t q_cols, int64_t q_off, void* stream) {
  QkvArgs a{(const bf16_t*)A, (const bf16_t*)W, (bf16_t*)C, table, lda, ldb, ldc, table_bs, q_off, (int)M, (int)N, (int)T, (int)pos_off,
            (int)(rot_cols / 64), (int)(q_cols / 64)};
  static bool once = (hipFuncSetAttribute(reinterpret_cast<const void*>(qkv_rope_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, QK_LDS) == hipSuccess);
  (void)once;
  hipLaunchKernelGGL(qkv_rope_fused_kernel, dim3((unsigned)fk_cdiv(M, MU_TOK)), dim3(MU_NW * 64), QK_LDS, (hipStream_t)stream, a);
  FK_CHECK_LAUNCH("fk_gemm_nt_rope (token-on-the-lane kernel)");
  return FK_OK;
}

// the token-on-the-lane form of fk_gemm_nt_swiglu for the shapes it is built for (called from gemm.hip's entry point)
bool fk_mlp_up_fused_ok(int64_t M, int64_t H, int64_t K, int64_t lda, int64_t ldb, int64_t ldh, int64_t ldg, int dtype) {
  static const bool off = getenv("FK_MLP_UP_FUSED") != nullptr && getenv("FK_MLP_UP_FUSED")[0] == '0';
  return !off && dtype == FK_BF16 && K == MF_D && H % 32 == 0 && mu_grid_fills(M) && M < (1LL << 31) && lda % 8 == 0 && ldb % 8 == 0 && ldh % 8 == 0 && ldg % 8 == 0 &&
         64 * ldb * 2 < (1LL << 32);
}
int fk_mlp_up_fused_launch(const void* A, int64_t lda, const void* W13, int64_t ldb, void* H13, int64_t ldh, void* G, int64_t ldg, int64_t M, int64_t H,
                           void* stream) {
  MlpUpArgs a{(const bf16_t*)A, (const bf16_t*)W13, (bf16_t*)H13, (bf16_t*)G, lda, ldb, ldh, ldg, (int)M, (int)H};
  static bool once = (hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_up_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, MU_LDS) == hipSuccess);
  (void)once;
  hipLaunchKernelGGL(mlp_up_fused_kernel, dim3((unsigned)fk_cdiv(M, MU_TOK)), dim3(MU_NW * 64), MU_LDS, (hipStream_t)stream, a);
  FK_CHECK_LAUNCH("fk_gemm_nt_swiglu (token-on-the-lane kernel)");
  return FK_OK;
}

#ifdef MF_STAMP
extern "C" int fk_debug_mf_stamps(unsigned long long* out, int reset) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(out, HIP_SYMBOL(mf_stamp_acc), sizeof(unsigned long long) * 16);
  if (reset) { unsigned long long z[16] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(mf_stamp_acc), z, sizeof(z)); }
  return 0;
}
#endif

extern "C" int fk_mlp_bwd_fused(const void* dY, int64_t lddy, const void* W2T, int64_t ldw2t, const void* H13, int64_t ldh, const void* W13T,
                                int64_t ldw13t, void* dH13, int64_t lddh, void* dX, int64_t lddx, int64_t M, int64_t H, int64_t D, int dtype,
                                void* stream) {
  FK_CHECK_ARG(dtype == FK_BF16, "fk_mlp_bwd_fused: bf16 only (dtype %d)", dtype);
  FK_CHECK_ARG(D == MF_D, "fk_mlp_bwd_fused: model dimension %lld (built for %d)", (long long)D, MF_D);
  FK_CHECK_ARG(M > 0 && M < (1LL << 31) && H > 0 && H % 32 == 0 && H < (1 << 24), "fk_mlp_bwd_fused: bad shape M=%lld H=%lld", (long long)M, (long long)H);
  FK_CHECK_ARG(dY && W2T && H13 && W13T && dH13 && dX, "fk_mlp_bwd_fused: null pointer");
  FK_CHECK_ARG(lddy % 8 == 0 && ldw2t % 8 == 0 && ldh % 8 == 0 && ldw13t % 8 == 0 && lddh % 8 == 0 && lddx % 8 == 0 && lddy >= D && ldw2t >= D &&
                   ldh >= 2 * H && lddh >= 2 * H && ldw13t >= 2 * H && lddx >= D,
               "fk_mlp_bwd_fused: leading dimensions must be multiples of 8 elements and cover their rows");
  FK_CHECK_ARG(M * ldh * 2 < (1LL << 32) && (int64_t)D * ldw13t * 2 < (1LL << 32) && 32 * ldw2t * 2 < (1LL << 32),
               "fk_mlp_bwd_fused: the h13 block must be addressable with 32-bit byte offsets (M * ldh < 2^31 elements)");
  FK_CHECK_ARG((((uintptr_t)dY | (uintptr_t)W2T | (uintptr_t)H13 | (uintptr_t)W13T | (uintptr_t)dH13 | (uintptr_t)dX) & 15) == 0,
               "fk_mlp_bwd_fused: pointers must be 16-byte aligned");
  MlpBwdArgs a{(const bf16_t*)dY, (const bf16_t*)W2T, (const bf16_t*)H13, (const bf16_t*)W13T, (bf16_t*)dH13, (bf16_t*)dX,
               lddy, ldw2t, ldh, ldw13t, lddh, lddx, (int)M, (int)H};
  static const bool asm_off = getenv("FK_MLP_BWD_ASM") != nullptr && getenv("FK_MLP_BWD_ASM")[0] == '0';
  static bool once = (hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_bwd_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, MF_LDS) == hipSuccess) &&
                     (hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_bwd_fused_asm_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, MF_LDS) == hipSuccess);
  (void)once;
  // whole 128-token tiles, and the dh13 rows addressable with 32-bit byte offsets from a per-chunk scalar base: the generated-stream kernel
  if (!asm_off && M % MF_TOK == 0 && M * lddh * 2 < (1LL << 32))
    hipLaunchKernelGGL(mlp_bwd_fused_asm_kernel, dim3((unsigned)(M / MF_TOK)), dim3(MF_NW * 64), MF_LDS, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(mlp_bwd_fused_kernel, dim3((unsigned)fk_cdiv(M, MF_TOK)), dim3(MF_NW * 64), MF_LDS, (hipStream_t)stream, a);
  FK_CHECK_LAUNCH("fk_mlp_bwd_fused");
  return FK_OK;
}
