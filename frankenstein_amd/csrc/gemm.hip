// gemm.hip — MFMA GEMMs for the projection / MLP / head layers (SURVEY.md §2.3 K3, K6, K7, K9 head).
//
//   fk_gemm_nt : C[M,N]  = A[M,K] * B[N,K]^T (+ bias[N]) (+ residual)      y = x W^T      (nn.Linear fwd;
//                                                                         dx = dy (W^T)^T with a W^T shadow)
//   fk_gemm_tn : C[N1,N2] (fp32) (+)= sum_m A[m,N1] * B[m,N2]             dW = dy^T x    (nn.Linear wgrad)
//   fk_colsum  : out[c] (+)= sum_r X[r,c]                                  db = sum dy    (bias grads)
//
// Tiling (both GEMMs): 128x128 block tile, 4 waves as 2x2, each wave 2x2 MFMA 32x32 tiles (64 fp32
// accumulators per lane); k-tiles of 128 bytes per row (64 bf16 / 32 fp32), double-buffered in LDS
// (64 KiB -> 2 blocks per CU), global->register->LDS staging with the loads for tile t+1 in flight
// during the MFMAs of tile t, one barrier per k-tile.
// LDS images:
//   NT: [row][128 B], 16-B chunk c of row r stored at chunk c ^ ((r>>1)&7): ds_read_b128 fragment
//       reads (same chunk, 16 different rows per lane group) are bank-conflict free.
//   TN: [k-row][128 cols]; bf16 fragments are k-strided -> ds_read_b64_tr_b16 (4 rows x 16 cols per
//       16-lane group); the 64-B granule g of row r is stored at granule g ^ (r&3) so the 4 rows of a
//       block hit different banks.  fp32 fragments use ds_read_b32 (conflict free unswizzled).
// blockIdx is remapped XCD-aware (xcd_remap) so tiles that share an A row panel share an L2.
#include "fk_common.h"

namespace {

#ifndef FK_EPI_UNROLL
#define FK_EPI_UNROLL 1   // the sweep is executed once per tile: unrolled it is ~40 KiB of straight-line code per kernel and the
#endif                    // instruction fetch (not VALU, LDS or HBM) bounds the epilogue; rolled it stays in the instruction cache
#include "gemm_tile.h"

struct NtArgs {
  const void* A; const void* B; void* C;
  const void* bias; const void* res;
  int64_t lda, ldb, ldc, ldr, res_rows;
  int M, N, K, vec_epi;
  int mode;            // 0 plain, 1 SwiGLU forward (also writes g), 2 SwiGLU backward (acc = dg -> writes dh13)
  void* aux; int64_t ldaux;
  // fused RoPE on the first rope_cols output columns (q and k of a packed qkv projection): row m is token
  // (m % rope_T) of its sample, rotated by table[(rope_off + m % rope_T)][(n % rope_D) / 2] = (cos, sin)
  const float* rope_table; int64_t rope_bs; int rope_T, rope_off, rope_D, rope_cols;
  // the first rope_qcols columns (the queries) read their (cos, sin) pairs rope_qoff floats further on: a second copy of the table
  // pre-multiplied by softmax_scale * log2(e), so Q leaves the projection in the exp2 domain of the attention kernels (FK_ATTN_Q_PRESCALED)
  int rope_qcols; int64_t rope_qoff;
};

// Output stores of the epilogue sweep: every line is written once and not read again by this launch.
#ifndef FK_NT_STORES_GEMM
#define FK_NT_STORES_GEMM 1      // 0: plain stores; 1 (default, -0.4 ms per cfg2 step): all fused-epilogue outputs non-temporal; 2: only the large ones (SwiGLU forward / backward, QKV + RoPE)
#endif
#ifndef FK_NT_LOADS_GEMM
#define FK_NT_LOADS_GEMM 0       // 1: the saved h13 rows of the SwiGLU backward (read once) loaded non-temporal; 2: the residual rows too
#endif
template <int LEVEL, typename V> FK_DEV V ld_once(const V* q) {
  if constexpr (FK_NT_LOADS_GEMM >= LEVEL) return __builtin_nontemporal_load(q);
  else return *q;
}
template <int EPI, typename V> FK_DEV void st_out(V* q, const V& v) {
  fk_st<(FK_NT_STORES_GEMM == 1) || (FK_NT_STORES_GEMM == 2 && EPI >= 1)>(q, v);
}

// sigmoid: the throughput (bf16) mode uses the hardware reciprocal (1 ulp), the fp32 parity mode an exact division
template <typename T> FK_DEV float sigmoid_f(float x) {
  if constexpr (sizeof(T) == 2) return __builtin_amdgcn_rcpf(1.0f + __expf(-x));
  else return 1.0f / (1.0f + __expf(-x));
}

// Epilogue shared by the NT kernels: acc are C^T tiles (lane = output row m, registers = 4 consecutive n).
// EPI < 0: every fused mode decided at run time (one code body); EPI = 0 plain / bias / residual, 1 SwiGLU forward, 2 SwiGLU backward,
// 3 RoPE (+ bias): the same code with the other modes compiled out (the 256 x 256 kernel is instantiated per mode).
template <typename T, typename TO, bool VEC_ONLY = false, int NI = 2, bool PRE_RES = false, int EPI = -1>
FK_DEV void nt_epilogue(const NtArgs& p, f32x16 (&acc)[NI][2], char* stg, int mrow0, int ncol0, int lane, bool sync) {
  // acc = one wave's (32 NI) x 64 sub-tile whose top-left output element is (mrow0, ncol0); stg = that wave's 8 NI KiB of LDS
  const int li = lane & 31, lh = lane >> 5;
  const T* bias = (EPI < 0 || EPI == 0 || EPI == 3) ? (const T*)p.bias : nullptr;
  const T* res = (EPI < 0 || EPI == 0) ? (const T*)p.res : nullptr;
  const int mode = EPI < 0 ? p.mode : (EPI == 1 ? 1 : (EPI == 2 ? 2 : 0));
  const float* rope_table = (EPI < 0 || EPI == 3) ? p.rope_table : nullptr;
  TO* C = (TO*)p.C;
  if (VEC_ONLY || p.vec_epi) {
    // Vector epilogue: accumulators are C^T tiles (lane = output row m, registers = 4 consecutive n), staged as
    // fp32 through this wave's 16 KiB slice of the (now idle) LDS tile buffers, then swept row-wise so every
    // global access is a full 16-byte-per-lane coalesced row segment; bias / residual added in fp32, one rounding.
    // staged rows are 256 B (64 fp32), 16-byte chunk c of row r stored at chunk c ^ (r & 15): conflict-free for the
    // column-of-rows b128 writes and the row-sweep b128 reads; 4 waves x 16 KiB = the 64 KiB already allocated.
    auto eoff = [](int row, int colf) { return row * 256 + ((((colf >> 2) ^ (row & 15)) << 4) | ((colf & 3) << 2)); };
    if (sync) __syncthreads();                      // all waves finished reading the operand tiles
    // The slice is reused by back-to-back calls (one per 32-row group): the previous call's last ds_read_b128s must have
    // returned before this call's ds_write_b128s are issued.  Without the wait, rows 30/31 of a group (the last lane groups of
    // the last pass) occasionally came back with the NEXT group's values — found by a run-to-run determinism check at full size.
    const int col = (lane & 7) * 8, nb = ncol0 + col;
    const bool col_ok = nb < p.N;                   // N % 8 == 0 on this path
    const int r0 = lane >> 3, mb = mrow0 + r0;      // this lane's row in pass 0; pass ps handles row mb + 8 * ps
    // Residual rows of the 4 passes (bf16, 32-row slices): fetched BEFORE the staging writes so that their latency hides behind the
    // LDS round trip instead of being paid once per pass of the rolled sweep (+45 us per N = 384 GEMM of the cfg2 step otherwise).
    constexpr bool PRE = PRE_RES && (NI == 1) && sizeof(T) == 2;   // (on the 256 x 256 kernel the extra live registers cost the fused epilogues 10 %: opt-in)
    bf16x8 rq0 = {}, rq1 = {}, rq2 = {}, rq3 = {};
    if constexpr (PRE) {
      if (res) {
        int rr = p.res_rows > 0 ? mb % (int)p.res_rows : mb;
        auto fetch = [&](int ps) {
          bf16x8 r = {};
          if (mb + 8 * ps < p.M && col_ok) r = ld_once<2>(reinterpret_cast<const bf16x8*>(res + (int64_t)rr * p.ldr + nb));
          rr += 8;
          if (p.res_rows > 0) { while (rr >= (int)p.res_rows) rr -= (int)p.res_rows; }
          return r;
        };
        rq0 = fetch(0); rq1 = fetch(1); rq2 = fetch(2); rq3 = fetch(3);
      }
    }
    // ... and for the (cos, sin) rows of the fused RoPE in the kernels instantiated for it (2 x 16 B per pass from the L2-resident table)
    constexpr bool PROPE = (EPI == 3) && (NI == 1);
    f32x4 ta0 = {}, ta1 = {}, ta2 = {}, ta3 = {}, tc0 = {}, tc1 = {}, tc2 = {}, tc3 = {};
    if constexpr (PROPE) {
      if (rope_table && nb < p.rope_cols && col_ok) {
        int tt = mb % p.rope_T;
        const float* tp = rope_table + (nb < p.rope_qcols ? p.rope_qoff : (int64_t)0) + (int64_t)(mb / p.rope_T) * p.rope_bs + ((int64_t)(p.rope_off + tt) * (p.rope_D / 2) + (nb % p.rope_D) / 2) * 2;
        auto fetch = [&](int ps, f32x4& lo, f32x4& hi) {
#ifdef FK_PROBE_NO_ROPE_TABLE      // timing builds only (wrong results): what the (cos, sin) loads of the fused RoPE cost
          if (mb + 8 * ps < p.M) { lo = f32x4{1.0f, 0.0f, 1.0f, 0.0f}; hi = lo; }
#else
          if (mb + 8 * ps < p.M) { lo = *reinterpret_cast<const f32x4*>(tp); hi = *reinterpret_cast<const f32x4*>(tp + 4); }
#endif
          tt += 8;
          tp += 8 * p.rope_D;
          while (tt >= p.rope_T) { tt -= p.rope_T; tp += p.rope_bs - (int64_t)p.rope_T * p.rope_D; }
        };
        fetch(0, ta0, tc0); fetch(1, ta1, tc1); fetch(2, ta2, tc2); fetch(3, ta3, tc3);
      }
    }
    // ... and for the saved h13 rows of the fused SwiGLU backward (2 x 16 B per pass); that instantiation unrolls the sweep so the
    // prefetch registers are indexed statically
    constexpr bool PAUX = (EPI == 2) && (NI == 1) && sizeof(T) == 2 && sizeof(TO) == 2;
    constexpr int SWEEP_UNROLL = (PAUX || EPI == 1) ? 4 : FK_EPI_UNROLL;
    bf16x8 hq0[4], hq1[4];
    if constexpr (PAUX) {
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) {
        hq0[ps] = bf16x8{};
        hq1[ps] = bf16x8{};
        if (mb + 8 * ps < p.M && col_ok) {
          const T* hp = (const T*)p.aux + (int64_t)(mb + 8 * ps) * p.ldaux + 2 * nb;
          hq0[ps] = ld_once<1>(reinterpret_cast<const bf16x8*>(hp));
          hq1[ps] = ld_once<1>(reinterpret_cast<const bf16x8*>(hp + 8));
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
          *reinterpret_cast<f32x4*>(stg + eoff(i * 32 + li, j * 32 + 8 * g + 4 * lh)) = v;
        }
    // The sweep is VALU-bound (a wave64 VALU op holds the SIMD for 4 cycles and no MFMA runs beside it), so everything
    // row- or column-invariant is hoisted: one division per sub-tile, pointer increments per pass, uniform branches
    // around absent bias / residual / RoPE work, and the bf16 mode uses v_rcp_f32 for the sigmoid.
    float bv[8];
    if (bias && col_ok) {
#pragma unroll
      for (int e = 0; e < 8; ++e) bv[e] = to_f32<T>(bias[nb + e]);
    }
    // residual row of pass 0 (periodic residual: row m % res_rows, advanced by 8 per pass with wrap)
    int rr = 0;
    if (res && !PRE) rr = p.res_rows > 0 ? mb % (int)p.res_rows : mb;
    // RoPE: (cos, sin) pairs of token (m % T) of sample (m / T) for the 4 complex pairs of this lane's 8 columns
    const float* tb = nullptr;
    int tt = 0;
    const bool do_rope = rope_table && nb < p.rope_cols && col_ok;
    if (do_rope && !PROPE) {
      tt = mb % p.rope_T;
      const int bb = mb / p.rope_T, dd = nb % p.rope_D;
      tb = rope_table + (nb < p.rope_qcols ? p.rope_qoff : (int64_t)0) + (int64_t)bb * p.rope_bs + ((int64_t)(p.rope_off + tt) * (p.rope_D / 2) + dd / 2) * 2;
    }
    // Output / auxiliary row pointers advance by 8 rows per pass: computed once here (a 64-bit multiply per pointer and pass is 5
    // quarter-rate VALU instructions, and the sweep is VALU-bound: they were a quarter of the fused SwiGLU epilogue)
    TO* cp_run = C + (int64_t)mb * p.ldc + nb;
    const int64_t cp_step = 8 * p.ldc;
    char* aux_run = nullptr;               // mode 1: g row (T, column nb / 2); mode 2: h13 row (T, column 2 nb), dh13 row = C (T, column 2 nb)
    int64_t aux_step = 0;
    T* dp_run = nullptr;
    if (mode == 1) { aux_run = (char*)((T*)p.aux + (int64_t)mb * p.ldaux + (nb >> 1)); aux_step = 8 * p.ldaux * (int64_t)sizeof(T); }
    if (mode == 2) {
      aux_run = (char*)((T*)p.aux + (int64_t)mb * p.ldaux + 2 * nb); aux_step = 8 * p.ldaux * (int64_t)sizeof(T);
      dp_run = (T*)p.C + (int64_t)mb * p.ldc + 2 * nb;
    }
#pragma unroll SWEEP_UNROLL
    for (int ps = 0; ps < 4 * NI; ++ps) {
      const int row = ps * 8 + r0, m = mrow0 + row;
      f32x4 a = *reinterpret_cast<const f32x4*>(stg + eoff(row, col));
      f32x4 b = *reinterpret_cast<const f32x4*>(stg + eoff(row, col + 4));
      if (m < p.M && col_ok) {
        float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        if (bias) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += bv[e];
        }
        if (res) {
          const T* rp = res + (int64_t)rr * p.ldr + nb;
          if constexpr (PRE) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += (float)rq0[e];
          } else if constexpr (sizeof(T) == 2) {
            bf16x8 r8 = *reinterpret_cast<const bf16x8*>(rp);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += (float)r8[e];
          } else {
            f32x4 q0 = *reinterpret_cast<const f32x4*>(rp), q1 = *reinterpret_cast<const f32x4*>(rp + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] += q0[e]; v[4 + e] += q1[e]; }
          }
        }
        if (do_rope) {
          f32x4 t0, t1;
          if constexpr (PROPE) { t0 = ta0; t1 = tc0; }
          else { t0 = *reinterpret_cast<const f32x4*>(tb); t1 = *reinterpret_cast<const f32x4*>(tb + 4); }
          const float cs[8] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float re = v[2 * j], im = v[2 * j + 1];
            // explicit fma shape: with -ffp-contract=fast hipcc otherwise picks which product is fused per instantiation, and the
            // kernels would differ in the last bit (the GPU suite compares the ring-buffered and the small kernel bit for bit)
            v[2 * j] = __builtin_fmaf(re, cs[2 * j], -(im * cs[2 * j + 1]));
            v[2 * j + 1] = __builtin_fmaf(re, cs[2 * j + 1], im * cs[2 * j]);
          }
        }
        if (mode == 2) {
          // SwiGLU backward fused into the down-projection dgrad: v = dg[m, nb..nb+7]; h13 / dh13 use the interleaved hidden
          // layout (per 4 hidden units: 4 x h1 then 4 x h3), so the 8 units of this lane are 16 contiguous columns.
          if constexpr (sizeof(TO) == sizeof(T)) {
            const T* hp = (const T*)aux_run;
            T* dp = dp_run;
            float hv[16], ov[16];
            if constexpr (PAUX) {
#pragma unroll
              for (int e = 0; e < 8; ++e) { hv[e] = (float)hq0[ps & 3][e]; hv[8 + e] = (float)hq1[ps & 3][e]; }
            } else if constexpr (sizeof(T) == 2) {
              bf16x8 h0 = *reinterpret_cast<const bf16x8*>(hp), h1 = *reinterpret_cast<const bf16x8*>(hp + 8);
#pragma unroll
              for (int e = 0; e < 8; ++e) { hv[e] = (float)h0[e]; hv[8 + e] = (float)h1[e]; }
            } else {
#pragma unroll
              for (int q4 = 0; q4 < 4; ++q4) {
                f32x4 t = *reinterpret_cast<const f32x4*>(hp + 4 * q4);
#pragma unroll
                for (int e = 0; e < 4; ++e) hv[4 * q4 + e] = t[e];
              }
            }
#pragma unroll
            for (int gq = 0; gq < 2; ++gq)
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float a1 = hv[8 * gq + e], a3 = hv[8 * gq + 4 + e], dg = v[4 * gq + e];
                const float sg = sigmoid_f<T>(a1), ds = dg * sg;
                ov[8 * gq + e] = ds * a3 * (1.0f + a1 * (1.0f - sg));
                ov[8 * gq + 4 + e] = ds * a1;
              }
            if constexpr (sizeof(T) == 2) {
              bf16x8 o0, o1;
#pragma unroll
              for (int e = 0; e < 8; ++e) { o0[e] = (bf16_t)ov[e]; o1[e] = (bf16_t)ov[8 + e]; }
              st_out<EPI>(reinterpret_cast<bf16x8*>(dp), o0);
              st_out<EPI>(reinterpret_cast<bf16x8*>(dp + 8), o1);
            } else {
#pragma unroll
              for (int q4 = 0; q4 < 4; ++q4)
                *reinterpret_cast<f32x4*>(dp + 4 * q4) = f32x4{ov[4 * q4], ov[4 * q4 + 1], ov[4 * q4 + 2], ov[4 * q4 + 3]};
            }
          }
        } else {
          TO* cp = cp_run;
          if constexpr (sizeof(TO) == 2) {
            bf16x8 o8;
#pragma unroll
            for (int e = 0; e < 8; ++e) o8[e] = (bf16_t)v[e];
            st_out<EPI>(reinterpret_cast<bf16x8*>(cp), o8);
          } else {
            *reinterpret_cast<f32x4*>(cp) = f32x4{v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>(cp + 4) = f32x4{v[4], v[5], v[6], v[7]};
          }
          if (mode == 1) {
            // SwiGLU forward fused into the up-projection: the 8 columns are (h1[4], h3[4]) of 4 hidden units
            if constexpr (sizeof(TO) == sizeof(T)) {
              T* gp = (T*)aux_run;
              float gq[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) gq[e] = v[e] * sigmoid_f<T>(v[e]) * v[4 + e];
              if constexpr (sizeof(T) == 2) {
                bf16x4 g4;
#pragma unroll
                for (int e = 0; e < 4; ++e) g4[e] = (bf16_t)gq[e];
                st_out<EPI>(reinterpret_cast<bf16x4*>(gp), g4);
              } else {
                *reinterpret_cast<f32x4*>(gp) = f32x4{gq[0], gq[1], gq[2], gq[3]};
              }
            }
          }
        }
      }
      // advance the row-dependent cursors by 8 rows
      cp_run += cp_step;
      aux_run += aux_step;
      if (mode == 2) dp_run += cp_step;
      if (res) {
        if constexpr (PRE) {
          rq0 = rq1; rq1 = rq2; rq2 = rq3;             // next pass's prefetched row
        } else {
          rr += 8;
          if (p.res_rows > 0) { while (rr >= (int)p.res_rows) rr -= (int)p.res_rows; }
        }
      }
      if (do_rope) {
        if constexpr (PROPE) {
          ta0 = ta1; ta1 = ta2; ta2 = ta3; tc0 = tc1; tc1 = tc2; tc2 = tc3;     // next pass's prefetched pair
        } else {
          tt += 8;
          tb += 8 * p.rope_D;
          while (tt >= p.rope_T) { tt -= p.rope_T; tb += p.rope_bs - (int64_t)p.rope_T * p.rope_D; }
        }
      }
    }
    return;
  }
  // scalar epilogue (any N / ldc): lane = output row m, registers = columns n
  if constexpr (!VEC_ONLY) {
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int m = mrow0 + i * 32 + li;
    if (m >= p.M) continue;
    const int64_t rr = p.res_rows > 0 ? (m % p.res_rows) : m;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = ncol0 + j * 32 + acc_row(r, lh);
        if (n >= p.N) continue;
        float v = acc[i][j][r] + (bias ? to_f32<T>(bias[n]) : 0.0f);
        if (res) v += to_f32<T>(res[rr * p.ldr + n]);
        C[(int64_t)m * p.ldc + n] = from_f32<TO>(v);
      }
    }
  }
  }
}

template <typename T, typename TO>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_nt_kernel(NtArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BK = KT<T>::BK, VEC = KT<T>::VEC, STEPS = KT<T>::STEPS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
  const int ntn = (p.N + BN - 1) / BN;
  const unsigned L = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (int)(L / ntn) * BM, n0 = (int)(L % ntn) * BN;
  const T* A = (const T*)p.A;
  const T* B = (const T*)p.B;

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
      ra[i] = (m0 + row < p.M && k < p.K) ? *reinterpret_cast<const u32x4*>(A + (int64_t)(m0 + row) * p.lda + k) : z;
      rb[i] = (n0 + row < p.N && k < p.K) ? *reinterpret_cast<const u32x4*>(B + (int64_t)(n0 + row) * p.ldb + k) : z;
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
        for (int j = 0; j < 2; ++j) mma32<T>(acc[i][j], fb[j], fa[i]);   // C^T tile: rows n, cols m
    }
    if (kt + 1 < nk) lstore((kt + 1) & 1);
    __syncthreads();
  }

  nt_epilogue<T, TO>(p, acc, smem + wave * 16384, m0 + (wave >> 1) * 64, n0 + (wave & 1) * 64, lane, true);
}

// bf16 fast path: operand tiles go global -> LDS directly (global_load_lds_dwordx4, 1 KiB = 8 image rows per wave
// instruction), no staging registers and no ds_write pass (the ds_write_b128 traffic was what bound the register-staged
// loop).  The LDS destination of an LDS-DMA is lane-linear, so the XOR swizzle of the image is applied to the per-lane
// SOURCE address (chunk c' of row r is fetched from logical chunk c' ^ ((r>>1)&7)); reads use nt_off unchanged.
// Rows beyond M / N are clamped (their products are never stored); needs K % 64 == 0.
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

template <typename TO>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_nt_glds_kernel(NtArgs p) {
  using T = bf16_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
  const int ntn = (p.N + BN - 1) / BN;
  const int ntiles = ((p.M + BM - 1) / BM) * ntn;
  // Persistent blocks (the grid is a few blocks per CU, not one per tile: at 128x128 the workgroup dispatch rate, not
  // the math, bounded the short-K GEMMs).  Blocks b, b+8, ... share an XCD (round-robin dispatch); XCD x owns the
  // contiguous tile range [x*T/8, (x+1)*T/8) and its blocks walk it round-robin, so concurrently running blocks of an XCD
  // work on neighbouring tiles (same A row panel in that XCD's L2).  Pure speed: any placement is correct.
  const int nb = gridDim.x, xcd = blockIdx.x & 7, jb = blockIdx.x >> 3, nbx = (nb + 7 - xcd) >> 3;
  const int q8 = ntiles >> 3, r8 = ntiles & 7;
  const int t_beg = xcd * q8 + min(xcd, r8), t_end = t_beg + q8 + (xcd < r8 ? 1 : 0);

  const T* srcA[4];
  const T* srcB[4];
  auto set_src = [&](int tile) {
    const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = (wave * 4 + j) * 8 + (lane >> 3);
      const int ch = (lane & 7) ^ ((row >> 1) & 7);
      srcA[j] = (const T*)p.A + (int64_t)min(m0 + row, p.M - 1) * p.lda + ch * 8;
      srcB[j] = (const T*)p.B + (int64_t)min(n0 + row, p.N - 1) * p.ldb + ch * 8;
    }
  };
  auto stage = [&](int buf, int k0) {
    char* as = smem + buf * 2 * TILE_BYTES + wave * 4096;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      __builtin_amdgcn_global_load_lds((glb_void_t*)(srcA[j] + k0), (lds_void_t*)(as + j * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_void_t*)(srcB[j] + k0), (lds_void_t*)(as + TILE_BYTES + j * 1024), 16, 0, 0);
    }
  };
  const int nk = p.K / 64;
  int tile = t_beg + jb;
  if (tile < t_end) { set_src(tile); stage(0, 0); }
  for (; tile < t_end; tile += nbx) {
    const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    __syncthreads();     // this tile's first k-stage has landed (vmcnt(0)); every wave is done with the previous tile's staging
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) stage((kt + 1) & 1, (kt + 1) * 64);
      const char* as = smem + (kt & 1) * 2 * TILE_BYTES;
      const char* bs = as + TILE_BYTES;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        Frag<T> fa[2], fb[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) nt_frag<T>(fa[i], as, wm * 64 + i * 32 + li, s, lh);
#pragma unroll
        for (int j = 0; j < 2; ++j) nt_frag<T>(fb[j], bs, wn * 64 + j * 32 + li, s, lh);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) mma32<T>(acc[i][j], fb[j], fa[i]);
      }
      __syncthreads();   // drains this wave's LDS-DMA (vmcnt) and orders every wave's reads / DMA writes
    }
    // Epilogue: two 32-row halves through this wave's 8 KiB slice of buffer 1, THEN the next tile's first k-stage into buffer 0.
    // (Issuing that DMA before the epilogue bought nothing measurable and made the RoPE epilogue return a few wrong elements per
    // ~10^8 — rows 30/31 of a group, run-to-run different; tools/determinism_probe.py and the full-size test guard this.)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      f32x16 (&sub)[1][2] = *reinterpret_cast<f32x16 (*)[1][2]>(&acc[i][0]);
      nt_epilogue<T, TO, false, 1>(p, sub, smem + 2 * TILE_BYTES + wave * 8192, m0 + wm * 64 + i * 32, n0 + wn * 64, lane, false);
    }
    if (tile + nbx < t_end) { set_src(tile + nbx); stage(0, 0); }
  }
}

template <int N> FK_DEV void vm_wait_barrier() {
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}
// Short-latency form of gemm_nt_glds_kernel for grids of at most one tile per CU (the perceiver's and the decoder's projections: 1024 rows,
// 24 tiles).  There the launch is one tile's dependent chain, and with two k-stages in the ring every stage paid a full request latency
// behind a vmcnt(0): 6 stages x ~1.5 us for K = 384.  Here the ring has four 32-KiB slots, three stages are requested before the first
// product, and the waits are counted (vmcnt retires in order; 8 requests per wave and stage), one barrier per stage:
//   wait for stage kt (own pieces) -> barrier (everyone's pieces; everyone has left stage kt - 1) -> request stage kt + 3 into the
//   slot of stage kt - 1 -> multiply stage kt.
// The epilogue stages through slot 3 behind one more barrier; the next tile's first stages follow it (see gemm_nt_glds_kernel).
constexpr int G4_LDS = 4 * 2 * TILE_BYTES;
static_assert(G4_LDS <= 160 * 1024, "LDS of a CU");
template <typename TO>
__global__ __launch_bounds__(NTHREADS, 1) void gemm_nt_glds4_kernel(NtArgs p) {
  using T = bf16_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
  const int ntn = (p.N + BN - 1) / BN;
  const int ntiles = ((p.M + BM - 1) / BM) * ntn;
  const int nb = gridDim.x, xcd = blockIdx.x & 7, jb = blockIdx.x >> 3, nbx = (nb + 7 - xcd) >> 3;
  const int q8 = ntiles >> 3, r8 = ntiles & 7;
  const int t_beg = xcd * q8 + min(xcd, r8), t_end = t_beg + q8 + (xcd < r8 ? 1 : 0);

  const T* srcA[4];
  const T* srcB[4];
  auto set_src = [&](int tile) {
    const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = (wave * 4 + j) * 8 + (lane >> 3);
      const int ch = (lane & 7) ^ ((row >> 1) & 7);
      srcA[j] = (const T*)p.A + (int64_t)min(m0 + row, p.M - 1) * p.lda + ch * 8;
      srcB[j] = (const T*)p.B + (int64_t)min(n0 + row, p.N - 1) * p.ldb + ch * 8;
    }
  };
  auto stage = [&](int buf, int k0) __attribute__((always_inline)) {
    char* as = smem + buf * 2 * TILE_BYTES + wave * 4096;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      __builtin_amdgcn_global_load_lds((glb_void_t*)(srcA[j] + k0), (lds_void_t*)(as + j * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_void_t*)(srcB[j] + k0), (lds_void_t*)(as + TILE_BYTES + j * 1024), 16, 0, 0);
    }
  };
  const int nk = p.K / 64;
  auto head = [&]() __attribute__((always_inline)) {
    stage(0, 0);
    if (nk > 1) stage(1, 64);
    if (nk > 2) stage(2, 128);
  };
  int tile = t_beg + jb;
  if (tile < t_end) { set_src(tile); head(); }
  for (; tile < t_end; tile += nbx) {
    const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    for (int kt = 0; kt < nk; ++kt) {
      const int after = nk - 1 - kt;                 // stages requested behind stage kt at this point: min(after, 2)
      if (after >= 2) vm_wait_barrier<16>(); else if (after == 1) vm_wait_barrier<8>(); else vm_wait_barrier<0>();
      if (kt + 3 < nk) stage((kt + 3) & 3, (kt + 3) * 64);
      const char* as = smem + (kt & 3) * 2 * TILE_BYTES;
      const char* bs = as + TILE_BYTES;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        Frag<T> fa[2], fb[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) nt_frag<T>(fa[i], as, wm * 64 + i * 32 + li, s, lh);
#pragma unroll
        for (int j = 0; j < 2; ++j) nt_frag<T>(fb[j], bs, wn * 64 + j * 32 + li, s, lh);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) mma32<T>(acc[i][j], fb[j], fa[i]);
      }
    }
    __syncthreads();                                 // every wave has read the last stage (it may sit in slot 3, the staging area)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      f32x16 (&sub)[1][2] = *reinterpret_cast<f32x16 (*)[1][2]>(&acc[i][0]);
      nt_epilogue<T, TO, false, 1>(p, sub, smem + 3 * 2 * TILE_BYTES + wave * 8192, m0 + wm * 64 + i * 32, n0 + wn * 64, lane, false);
    }
    if (tile + nbx < t_end) { set_src(tile + nbx); head(); }
  }
}

// Large-tile variant for the big projections: 256 x BN_ block tile (BN_ = 256 or 128), 8 waves (2 per SIMD), wave tile
// (256/WROWS) x 64.  A 128x128 tile needs 64 B of L2->LDS traffic per MFMA cycle per CU at full rate, more than the L2
// fabric delivers (measured ~8 TB/s chip-wide on the small-tile kernel); 256x256 halves that (32 B/cycle) and gives each
// LDS-DMA a whole k-tile of two waves' MFMAs (2 x 1024 cycles per SIMD) to land.  Same image / swizzle / epilogue code.
template <typename TO, int BN_>
__global__ __launch_bounds__(512, 2) void gemm_nt_big_kernel(NtArgs p) {
  using T = bf16_t;
  constexpr int BM_ = 256, WCOLS = BN_ / 64, WROWS = 8 / WCOLS, WM = BM_ / WROWS, MT = WM / 32;
  constexpr int A_BYTES = BM_ * ROW_BYTES, B_BYTES = BN_ * ROW_BYTES, STAGE = A_BYTES + B_BYTES;
  constexpr int A_PER_WAVE = BM_ / 64, B_PER_WAVE = BN_ / 64;     // 1-KiB DMA instructions per wave and stage
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WCOLS, wn = wave % WCOLS, li = lane & 31, lh = lane >> 5;
  const int ntn = (p.N + BN_ - 1) / BN_;
  const int ntiles = ((p.M + BM_ - 1) / BM_) * ntn;
  const int nb = gridDim.x, xcd = blockIdx.x & 7, jb = blockIdx.x >> 3, nbx = (nb + 7 - xcd) >> 3;
  const int q8 = ntiles >> 3, r8 = ntiles & 7;
  const int t_beg = xcd * q8 + min(xcd, r8), t_end = t_beg + q8 + (xcd < r8 ? 1 : 0);

  static_assert(STAGE >= 8 * 8192, "the epilogue stages 8 waves x 8 KiB through the second stage buffer");
  const T* srcA[A_PER_WAVE];
  const T* srcB[B_PER_WAVE];
  auto set_src = [&](int tile) {
    const int m0 = (tile / ntn) * BM_, n0 = (tile % ntn) * BN_;
#pragma unroll
    for (int j = 0; j < A_PER_WAVE; ++j) {
      const int row = (wave * A_PER_WAVE + j) * 8 + (lane >> 3);
      srcA[j] = (const T*)p.A + (int64_t)min(m0 + row, p.M - 1) * p.lda + ((lane & 7) ^ ((row >> 1) & 7)) * 8;
    }
#pragma unroll
    for (int j = 0; j < B_PER_WAVE; ++j) {
      const int row = (wave * B_PER_WAVE + j) * 8 + (lane >> 3);
      srcB[j] = (const T*)p.B + (int64_t)min(n0 + row, p.N - 1) * p.ldb + ((lane & 7) ^ ((row >> 1) & 7)) * 8;
    }
  };
  auto stage = [&](int buf, int k0) {
    char* as = smem + buf * STAGE;
#pragma unroll
    for (int j = 0; j < A_PER_WAVE; ++j)
      __builtin_amdgcn_global_load_lds((glb_void_t*)(srcA[j] + k0), (lds_void_t*)(as + (wave * A_PER_WAVE + j) * 1024), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < B_PER_WAVE; ++j)
      __builtin_amdgcn_global_load_lds((glb_void_t*)(srcB[j] + k0), (lds_void_t*)(as + A_BYTES + (wave * B_PER_WAVE + j) * 1024), 16, 0, 0);
  };
  const int nk = p.K / 64;
  int tile = t_beg + jb;
  if (tile < t_end) { set_src(tile); stage(0, 0); }
  for (; tile < t_end; tile += nbx) {
    const int m0 = (tile / ntn) * BM_, n0 = (tile % ntn) * BN_;
    f32x16 acc[MT][2];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    __syncthreads();     // first k-stage landed; previous tile's staging (buffer 1) no longer read by anyone
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) stage((kt + 1) & 1, (kt + 1) * 64);
      const char* as = smem + (kt & 1) * STAGE;
      const char* bs = as + A_BYTES;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        Frag<T> fa[MT], fb[2];
#pragma unroll
        for (int i = 0; i < MT; ++i) nt_frag<T>(fa[i], as, wm * WM + i * 32 + li, s, lh);
#pragma unroll
        for (int j = 0; j < 2; ++j) nt_frag<T>(fb[j], bs, wn * 64 + j * 32 + li, s, lh);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) mma32<T>(acc[i][j], fb[j], fa[i]);
      }
      __syncthreads();
    }
    // the epilogue stages 32-row slices of the wave tile through this wave's 8 KiB of buffer 1; the next tile's first k-stage
    // is issued after it (see gemm_nt_glds_kernel)
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      f32x16 (&sub)[1][2] = *reinterpret_cast<f32x16 (*)[1][2]>(&acc[i][0]);
      nt_epilogue<T, TO, true, 1>(p, sub, smem + STAGE + wave * 8192, m0 + wm * WM + i * 32, n0 + wn * 64, lane, false);
    }
    if (tile + nbx < t_end) { set_src(tile + nbx); stage(0, 0); }
  }
}

// Ring-buffered NT kernels for the wide bf16 projections.  The double-buffered kernels above keep ONE k-stage in flight per block
// and drain it (vmcnt(0)) at every barrier: measured with the MFMAs compiled out (tools/gemm_ksweep.py on a probe build) the fetch
// stream alone takes 83 % of the per-k time, and one stage in flight fetches 1.35x slower than two -- the loop is bound by
// memory latency x bytes in flight, not by MFMA or LDS rate.  Here the k-stages of ALL tiles of a persistent block form one stream
// through NS ring slots: stage g+NS-1 is issued when stage g starts computing, the wait before stage g is the counted
// `s_waitcnt vmcnt((NS-2) PER)` (vmcnt retires in order; PER = LDS-DMA instructions per wave and stage), and the stream runs across
// tile boundaries, so the next tile's first stages land during the epilogue.  The epilogue stages through the slot just computed
// plus the LDS above the ring.  Shapes, all 8 waves and the whole 160 KiB of a CU:
//   N = 384-class : gemm_nt_ring_kernel<128, 128, 3>: 256 x 128 tile, 64-element k-stages (128-B image rows), 2 x 48 KiB in flight
//   N >= 1024     : gemm_nt_ring2_kernel below: 256 x 256 tile with separate A / B rings
// (a 256 x 256 tile with 32-element stages, <256, 64, 4>, also works but its 64-byte requests fetch 1.4x slower per byte).
template <int BN_, int RB, int NS> struct Ring {
  static constexpr int BM_ = 256;
  static constexpr int STAGE = (BM_ + BN_) * RB, A_BYTES = BM_ * RB;
  static constexpr int IN_SLOT = STAGE / 8192 < 8 ? STAGE / 8192 : 8;       // waves whose 8-KiB epilogue slice fits in a ring slot
  static constexpr int LDS = NS * STAGE + (8 - IN_SLOT) * 8192;
  static constexpr int APW = BM_ * RB / 8192, BPW = BN_ * RB / 8192, PER = APW + BPW;
  static_assert(LDS <= 160 * 1024, "ring does not fit the LDS of a CU");
};
// image of a k-stage: row r holds RB bytes; its 16-byte chunk c sits at chunk c ^ swz(r) (conflict-free ds_read_b128 down a column)
template <int RB> FK_DEV int ring_swz(int row) { return RB == 128 ? ((row >> 1) & 7) : ((row >> 2) & 3); }
template <int RB> FK_DEV int ring_off(int row, int chunk) { return row * RB + ((chunk ^ ring_swz<RB>(row)) << 4); }

template <typename TO, int BN_, int RB, int NS, int EPI>
__global__ __launch_bounds__(512, 2) void gemm_nt_ring_kernel(NtArgs p) {
  using T = bf16_t;
  using R = Ring<BN_, RB, NS>;
  constexpr int BM_ = R::BM_, STAGE = R::STAGE, A_BYTES = R::A_BYTES, APW = R::APW, BPW = R::BPW, PER = R::PER;
  constexpr int BKE = RB / 2, SSTEPS = RB / 32, RPI = 1024 / RB, LPR = RB / 16;   // k elements / k16 steps per stage; rows, lanes per row of a DMA
  constexpr int WCOLS = BN_ / 64, WROWS = 8 / WCOLS, WM = BM_ / WROWS, MT = WM / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WCOLS, wn = wave % WCOLS, li = lane & 31, lh = lane >> 5;
  const int ntn = (p.N + BN_ - 1) / BN_;
  const int ntiles = ((p.M + BM_ - 1) / BM_) * ntn;
  const int nb = gridDim.x, xcd = blockIdx.x & 7, jb = blockIdx.x >> 3, nbx = (nb + 7 - xcd) >> 3;
  const int q8 = ntiles >> 3, r8 = ntiles & 7;
  const int t_beg = xcd * q8 + min(xcd, r8), t_end = t_beg + q8 + (xcd < r8 ? 1 : 0);
  const int nk = p.K / BKE;

  const T* srcA[APW];
  const T* srcB[BPW];
  auto set_src = [&](int tile) {
    const int m0 = (tile / ntn) * BM_, n0 = (tile % ntn) * BN_;
#pragma unroll
    for (int j = 0; j < APW; ++j) {
      const int row = (wave * APW + j) * RPI + lane / LPR;
      srcA[j] = (const T*)p.A + (int64_t)min(m0 + row, p.M - 1) * p.lda + ((lane % LPR) ^ ring_swz<RB>(row)) * 8;
    }
#pragma unroll
    for (int j = 0; j < BPW; ++j) {
      const int row = (wave * BPW + j) * RPI + lane / LPR;
      srcB[j] = (const T*)p.B + (int64_t)min(n0 + row, p.N - 1) * p.ldb + ((lane % LPR) ^ ring_swz<RB>(row)) * 8;
    }
  };
  // issue cursor of the stage stream: (itile, ikt) is the next stage to fetch
  int itile = t_beg + jb, ikt = 0;
  if (itile < t_end) set_src(itile);
  // The PER requests of a stage are issued in SSTEPS parts, one behind the MFMAs of every k16-step of the stage being computed (part
  // < 0: all at once).  Issued together right after the barrier, the 48 requests of the eight waves queue up at the CU's one address unit
  // while no MFMA is left in the pipe (worth 1.5-2 % of the long-K launches).
  auto issue = [&](int slot, int part) __attribute__((always_inline)) {
    if (itile >= t_end) return;
    char* as = smem + slot * STAGE;
    const int k0 = ikt * BKE;
#pragma unroll
    for (int j = 0; j < APW; ++j)
      if (part < 0 || (j * SSTEPS) / PER == part)
        __builtin_amdgcn_global_load_lds((glb_void_t*)(srcA[j] + k0), (lds_void_t*)(as + (wave * APW + j) * 1024), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < BPW; ++j)
      if (part < 0 || ((APW + j) * SSTEPS) / PER == part)
        __builtin_amdgcn_global_load_lds((glb_void_t*)(srcB[j] + k0), (lds_void_t*)(as + A_BYTES + (wave * BPW + j) * 1024), 16, 0, 0);
    if (part < 0 || part == SSTEPS - 1) {
      if (++ikt == nk) {
        ikt = 0;
        itile += nbx;
        if (itile < t_end) set_src(itile);
      }
    }
  };
#pragma unroll
  for (int s = 0; s < NS - 1; ++s) issue(s, -1);
  int slot = 0;                                     // ring slot of the stage being computed
  for (int tile = t_beg + jb; tile < t_end; tile += nbx) {
    const int m0 = (tile / ntn) * BM_, n0 = (tile % ntn) * BN_;
    const bool last_tile = tile + nbx >= t_end;
    f32x16 acc[MT][2];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    for (int kt = 0; kt < nk; ++kt) {
      // This wave's part of the stage has landed once at most the NS-2 following stages' instructions are outstanding (fewer at
      // the end of the stream; the epilogue's stores, issued later still, only make the wait stricter).  The barrier extends
      // that to every wave's part and says every wave has finished with the slot that is refilled next.
#ifdef FK_RING_PROBE_DRAIN         // probe builds: one stage in flight, as in the double-buffered kernels
      vm_wait_barrier<0>();
#else
      const int after = last_tile ? nk - 1 - kt : NS;          // stages of the stream behind this one (NS = "plenty")
      if (after >= NS - 2) vm_wait_barrier<(NS - 2) * PER>();
      else if (NS == 4 && after == 1) vm_wait_barrier<PER>();
      else vm_wait_barrier<0>();
#endif
      const int fill = slot == 0 ? NS - 1 : slot - 1;            // the slot every wave left at this barrier
#ifdef FK_RING_BURST_ISSUE
      issue(fill, -1);
#endif
      const char* as = smem + slot * STAGE;
      const char* bs = as + A_BYTES;
#ifndef FK_RING_PROBE_NOMMA        // probe builds (tools/): fetch stream only
#pragma unroll
      for (int s = 0; s < SSTEPS; ++s) {
        Frag<T> fa[MT], fb[2];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const int row = wm * WM + i * 32 + li;
          fa[i].v = *reinterpret_cast<const bf16x8*>(as + ring_off<RB>(row, 2 * s + lh));
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int row = wn * 64 + j * 32 + li;
          fb[j].v = *reinterpret_cast<const bf16x8*>(bs + ring_off<RB>(row, 2 * s + lh));
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) mma32<T>(acc[i][j], fb[j], fa[i]);
#ifndef FK_RING_BURST_ISSUE
        __builtin_amdgcn_sched_barrier(0);
        issue(fill, s);
        __builtin_amdgcn_sched_barrier(0);
#endif
      }
#else
#ifndef FK_RING_BURST_ISSUE
      issue(fill, -1);
#endif
#endif
      slot = slot == NS - 1 ? 0 : slot + 1;
    }
    const int done = slot == 0 ? NS - 1 : slot - 1;  // slot of the tile's last stage: the staging area once every wave has read it
    asm volatile("s_barrier" ::: "memory");
    char* stg = wave < R::IN_SLOT ? smem + done * STAGE + wave * 8192 : smem + NS * STAGE + (wave - R::IN_SLOT) * 8192;
#ifdef FK_RING_PROBE_NOEPI          // probe builds: main loops only (one store per tile keeps the accumulators alive)
    if (acc[0][0][0] == 123.456f) ((float*)p.C)[tid] = acc[MT - 1][1][3];
#else
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      f32x16 (&sub)[1][2] = *reinterpret_cast<f32x16 (*)[1][2]>(&acc[i][0]);
      if (EPI == 0 && BN_ == 128 && p.res) nt_epilogue<T, TO, true, 1, true, EPI>(p, sub, stg, m0 + wm * WM + i * 32, n0 + wn * 64, lane, false);
      else nt_epilogue<T, TO, true, 1, false, EPI>(p, sub, stg, m0 + wm * WM + i * 32, n0 + wn * 64, lane, false);
    }
#endif
  }
}

// 256 x 256 tile with full-line (128-B) requests: three slots do not fit (3 x 64 KiB), so the operands get rings of different depth,
// A (activations, streamed from HBM) three 32-KiB slots and B (weights, L2-resident) two.  Per step the wave issues B(g+1) and then
// A(g+2); the wait before stage g is vmcnt(APW): everything but the 4 newest instructions, i.e. all of A(g) and B(g), has landed
// and A(g+1) stays in flight across the barrier, so the memory pipe never drains.  The epilogue stages through the A and B slots
// just computed (4 waves each).  LDS = 96 + 64 KiB.
constexpr int R2_A = 256 * ROW_BYTES, R2_LDS = 5 * R2_A;
static_assert(R2_LDS <= 160 * 1024, "LDS of a CU");

template <typename TO, int EPI>
__global__ __launch_bounds__(512, 2) void gemm_nt_ring2_kernel(NtArgs p) {
  using T = bf16_t;
  constexpr int BM_ = 256, BN_ = 256, APW = 4, BPW = 4, WM = 128, MT = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  auto aslot = [&](int i) -> char* { return smem + i * R2_A; };
  auto bslot = [&](int i) -> char* { return smem + (3 + i) * R2_A; };
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3, li = lane & 31, lh = lane >> 5;
  const int ntn = (p.N + BN_ - 1) / BN_;
  const int ntiles = ((p.M + BM_ - 1) / BM_) * ntn;
  const int nb = gridDim.x, xcd = blockIdx.x & 7, jb = blockIdx.x >> 3, nbx = (nb + 7 - xcd) >> 3;
  const int q8 = ntiles >> 3, r8 = ntiles & 7;
  const int t_beg = xcd * q8 + min(xcd, r8), t_end = t_beg + q8 + (xcd < r8 ? 1 : 0);
  const int nk = p.K / 64;

  const T* srcA[APW];
  const T* srcB[BPW];
  auto set_a = [&](int tile) {
    const int m0 = (tile / ntn) * BM_;
#pragma unroll
    for (int j = 0; j < APW; ++j) {
      const int row = (wave * APW + j) * 8 + (lane >> 3);
      srcA[j] = (const T*)p.A + (int64_t)min(m0 + row, p.M - 1) * p.lda + ((lane & 7) ^ ((row >> 1) & 7)) * 8;
    }
  };
  auto set_b = [&](int tile) {
    const int n0 = (tile % ntn) * BN_;
#pragma unroll
    for (int j = 0; j < BPW; ++j) {
      const int row = (wave * BPW + j) * 8 + (lane >> 3);
      srcB[j] = (const T*)p.B + (int64_t)min(n0 + row, p.N - 1) * p.ldb + ((lane & 7) ^ ((row >> 1) & 7)) * 8;
    }
  };
  // two issue cursors over the same stage stream: A runs two stages ahead of the compute, B one
  int atile = t_beg + jb, akt = 0, btile = atile, bkt = 0;
  if (atile < t_end) { set_a(atile); set_b(btile); }
  // half = 0 / 1: the first / second two requests of the stage, -1: all four (the steps of the loop issue B in the gaps behind the first
  // two k16-steps and A behind the last two — the order the counted wait relies on — instead of all eight right after the barrier;
  // measured -1.5...-2 % on the 256 x 128 kernel, +-0 here)
  auto issue_a = [&](int slot, int half) __attribute__((always_inline)) {
    if (atile >= t_end) return;
    char* as = aslot(slot);
#pragma unroll
    for (int j = 0; j < APW; ++j)
      if (half < 0 || j / 2 == half)
        __builtin_amdgcn_global_load_lds((glb_void_t*)(srcA[j] + akt * 64), (lds_void_t*)(as + (wave * APW + j) * 1024), 16, 0, 0);
    if (half != 0) { if (++akt == nk) { akt = 0; atile += nbx; if (atile < t_end) set_a(atile); } }
  };
  auto issue_b = [&](int slot, int half) __attribute__((always_inline)) {
    if (btile >= t_end) return;
    char* bs = bslot(slot);
#pragma unroll
    for (int j = 0; j < BPW; ++j)
      if (half < 0 || j / 2 == half)
        __builtin_amdgcn_global_load_lds((glb_void_t*)(srcB[j] + bkt * 64), (lds_void_t*)(bs + (wave * BPW + j) * 1024), 16, 0, 0);
    if (half != 0) { if (++bkt == nk) { bkt = 0; btile += nbx; if (btile < t_end) set_b(btile); } }
  };
  issue_a(0, -1);
  issue_b(0, -1);
  issue_a(1, -1);
  int sa = 0, sb = 0;                               // slots of the stage being computed
  for (int tile = t_beg + jb; tile < t_end; tile += nbx) {
    const int m0 = (tile / ntn) * BM_, n0 = (tile % ntn) * BN_;
    const bool last_tile = tile + nbx >= t_end;
    f32x16 acc[MT][2];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    for (int kt = 0; kt < nk; ++kt) {
      if (last_tile && kt == nk - 1) vm_wait_barrier<0>(); else vm_wait_barrier<APW>();
      const int fill_a = sa == 0 ? 2 : sa - 1, fill_b = sb ^ 1;
      const char* as = aslot(sa);
      const char* bs = bslot(sb);
#ifndef FK_RING_PROBE_NOMMA
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        Frag<T> fa[MT], fb[2];
#pragma unroll
        for (int i = 0; i < MT; ++i) nt_frag<T>(fa[i], as, wm * WM + i * 32 + li, s, lh);
#pragma unroll
        for (int j = 0; j < 2; ++j) nt_frag<T>(fb[j], bs, wn * 64 + j * 32 + li, s, lh);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) mma32<T>(acc[i][j], fb[j], fa[i]);
        __builtin_amdgcn_sched_barrier(0);
        if (s < 2) issue_b(fill_b, s); else issue_a(fill_a, s - 2);
        __builtin_amdgcn_sched_barrier(0);
      }
#else
      issue_b(fill_b, -1);
      issue_a(fill_a, -1);
#endif
      sa = sa == 2 ? 0 : sa + 1;
      sb ^= 1;
    }
    asm volatile("s_barrier" ::: "memory");          // every wave has read the last stage: its two slots become the staging area
    char* stg = wave < 4 ? aslot(sa == 0 ? 2 : sa - 1) + wave * 8192 : bslot(sb ^ 1) + (wave - 4) * 8192;
#ifdef FK_RING_PROBE_NOEPI
    if (acc[0][0][0] == 123.456f) ((float*)p.C)[tid] = acc[MT - 1][1][3];
#else
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      f32x16 (&sub)[1][2] = *reinterpret_cast<f32x16 (*)[1][2]>(&acc[i][0]);
      nt_epilogue<T, TO, true, 1, false, EPI>(p, sub, stg, m0 + wm * WM + i * 32, n0 + wn * 64, lane, false);
    }
#endif
  }
}

// 256 x 192 tile for N = 384 (two column tiles instead of three 128-wide ones).  These launches are bound by the bytes that go through
// LDS-DMA (57 GB/s per CU out of L2, 24 out of HBM): a 256 x 128 tile moves 384 rows per k-stage for 256 x 128 outputs and the
// three column tiles fetch the same A rows three times; 256 x 192 moves 448 rows for 256 x 192 outputs and fetches A twice: 22 % fewer
// bytes for the same product.  Rings as in the 256 x 256 kernel (A: three 32-KiB slots, two stages ahead; B: two 24-KiB slots, one ahead;
// counted wait vmcnt(4)); every wave owns 32 rows and all 192 columns (6 accumulator tiles), so the epilogue is three 64-column sweeps.
constexpr int R192_A = 256 * ROW_BYTES, R192_B = 192 * ROW_BYTES, R192_LDS = 3 * R192_A + 2 * R192_B + 8192;
static_assert(R192_LDS <= 160 * 1024, "LDS of a CU");

template <typename TO, int EPI>
__global__ __launch_bounds__(512, 2) void gemm_nt_ring192_kernel(NtArgs p) {
  using T = bf16_t;
  constexpr int BM_ = 256, BN_ = 192, APW = 4, BPW = 3, NT6 = 6;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  auto aslot = [&](int i) -> char* { return smem + i * R192_A; };
  auto bslot = [&](int i) -> char* { return smem + 3 * R192_A + i * R192_B; };
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int ntn = p.N / BN_;
  const int ntiles = ((p.M + BM_ - 1) / BM_) * ntn;
  const int nb = gridDim.x, xcd = blockIdx.x & 7, jb = blockIdx.x >> 3, nbx = (nb + 7 - xcd) >> 3;
  const int q8 = ntiles >> 3, r8 = ntiles & 7;
  const int t_beg = xcd * q8 + min(xcd, r8), t_end = t_beg + q8 + (xcd < r8 ? 1 : 0);
  const int nk = p.K / 64;

  const T* srcA[APW];
  const T* srcB[BPW];
  auto set_a = [&](int tile) {
    const int m0 = (tile / ntn) * BM_;
#pragma unroll
    for (int j = 0; j < APW; ++j) {
      const int row = (wave * APW + j) * 8 + (lane >> 3);
      srcA[j] = (const T*)p.A + (int64_t)min(m0 + row, p.M - 1) * p.lda + ((lane & 7) ^ ((row >> 1) & 7)) * 8;
    }
  };
  auto set_b = [&](int tile) {
    const int n0 = (tile % ntn) * BN_;
#pragma unroll
    for (int j = 0; j < BPW; ++j) {
      const int row = (wave * BPW + j) * 8 + (lane >> 3);
      srcB[j] = (const T*)p.B + (int64_t)(n0 + row) * p.ldb + ((lane & 7) ^ ((row >> 1) & 7)) * 8;
    }
  };
  int atile = t_beg + jb, akt = 0, btile = atile, bkt = 0;
  if (atile < t_end) { set_a(atile); set_b(btile); }
  // part 0 / 1 of a stage's requests (-1: all): B first, A after it - the order the counted wait relies on
  auto issue_a = [&](int slot, int part) __attribute__((always_inline)) {
    if (atile >= t_end) return;
    char* as = aslot(slot);
#pragma unroll
    for (int j = 0; j < APW; ++j)
      if (part < 0 || j / 2 == part)
        __builtin_amdgcn_global_load_lds((glb_void_t*)(srcA[j] + akt * 64), (lds_void_t*)(as + (wave * APW + j) * 1024), 16, 0, 0);
    if (part != 0) { if (++akt == nk) { akt = 0; atile += nbx; if (atile < t_end) set_a(atile); } }
  };
  auto issue_b = [&](int slot, int part) __attribute__((always_inline)) {
    if (btile >= t_end) return;
    char* bs = bslot(slot);
#pragma unroll
    for (int j = 0; j < BPW; ++j)
      if (part < 0 || (j + 1) / 2 == part)                       // part 0: piece 0, part 1: pieces 1 and 2
        __builtin_amdgcn_global_load_lds((glb_void_t*)(srcB[j] + bkt * 64), (lds_void_t*)(bs + (wave * BPW + j) * 1024), 16, 0, 0);
    if (part != 0) { if (++bkt == nk) { bkt = 0; btile += nbx; if (btile < t_end) set_b(btile); } }
  };
  issue_a(0, -1);
  issue_b(0, -1);
  issue_a(1, -1);
  int sa = 0, sb = 0;
  for (int tile = t_beg + jb; tile < t_end; tile += nbx) {
    const int m0 = (tile / ntn) * BM_, n0 = (tile % ntn) * BN_;
    const bool last_tile = tile + nbx >= t_end;
    f32x16 acc[NT6];
#pragma unroll
    for (int j = 0; j < NT6; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
    for (int kt = 0; kt < nk; ++kt) {
      if (last_tile && kt == nk - 1) vm_wait_barrier<0>(); else vm_wait_barrier<APW>();
      const int fill_a = sa == 0 ? 2 : sa - 1, fill_b = sb ^ 1;
      const char* as = aslot(sa);
      const char* bs = bslot(sb);
#ifndef FK_RING_PROBE_NOMMA
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        Frag<T> fa, fb[NT6];
        nt_frag<T>(fa, as, wave * 32 + li, s, lh);
#pragma unroll
        for (int j = 0; j < NT6; ++j) nt_frag<T>(fb[j], bs, j * 32 + li, s, lh);
#pragma unroll
        for (int j = 0; j < NT6; ++j) mma32<T>(acc[j], fb[j], fa);
        __builtin_amdgcn_sched_barrier(0);
        if (s < 2) issue_b(fill_b, s); else issue_a(fill_a, s - 2);
        __builtin_amdgcn_sched_barrier(0);
      }
#else
      issue_b(fill_b, -1);
      issue_a(fill_a, -1);
#endif
      sa = sa == 2 ? 0 : sa + 1;
      sb ^= 1;
    }
    asm volatile("s_barrier" ::: "memory");          // every wave has read the last stage: its slots (and the spare 8 KiB) are the staging area
    char* stg = wave < 4 ? aslot(sa == 0 ? 2 : sa - 1) + wave * 8192
              : (wave < 7 ? bslot(sb ^ 1) + (wave - 4) * 8192 : smem + 3 * R192_A + 2 * R192_B);
#ifdef FK_RING_PROBE_NOEPI
    if (acc[0][0] == 123.456f) ((float*)p.C)[tid] = acc[NT6 - 1][3];
#else
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      f32x16 (&sub)[1][2] = *reinterpret_cast<f32x16 (*)[1][2]>(&acc[2 * c]);
      if (EPI == 0 && p.res) nt_epilogue<T, TO, true, 1, true, EPI>(p, sub, stg, m0 + wave * 32, n0 + 64 * c, lane, false);
      else nt_epilogue<T, TO, true, 1, false, EPI>(p, sub, stg, m0 + wave * 32, n0 + 64 * c, lane, false);
    }
#endif
  }
}

// Grid of the persistent NT kernels: G workgroups per CU-sized wave (FK_NT_GRID_MULT, default 4), each walking its static share of its
// XCD's tile range; the surplus workgroups start as CUs come free.  With one workgroup per CU (G = 1) a CU that another kernel holds —
// a collective's channel beside the backward — makes its workgroup start late and the launch end a whole range late: beside an occupant
// of 8 workgroups x 48 KiB of LDS the cfg2 launches took x1.33 ... x1.66 their quiet time; with G = 4 x0.91 ... x1.06, at the same quiet
// time and the same step time (tools/occupant_probe.py, profiles/r04_g_occupant_probe.txt, r04_g_grid_mult_step.txt).
static unsigned persistent_grid(int64_t nt) {
  static const int g = [] { const char* e = getenv("FK_NT_GRID_MULT"); const int v = e ? atoi(e) : 4; return v < 1 ? 1 : (v > 64 ? 64 : v); }();
  const int64_t cap = 256LL * g;
  return (unsigned)(nt < cap ? nt : cap);
}

template <typename TO>
static void launch_ring192(const NtArgs& p, int64_t M, int64_t N, hipStream_t s) {
  const int64_t nt = fk_cdiv(M, 256) * (N / 192);
  static bool once = (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_ring192_kernel<TO, 0>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, R192_LDS) == hipSuccess);
  (void)once;
  hipLaunchKernelGGL((gemm_nt_ring192_kernel<TO, 0>), dim3(persistent_grid(nt)), dim3(512), R192_LDS, s, p);
}

template <typename TO, int EPI>
static void launch_ring2_epi(const NtArgs& p, int64_t M, int64_t N, hipStream_t s) {
  const int64_t nt = fk_cdiv(M, 256) * fk_cdiv(N, 256);
  static bool once = (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_ring2_kernel<TO, EPI>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, R2_LDS) == hipSuccess);
  (void)once;
  hipLaunchKernelGGL((gemm_nt_ring2_kernel<TO, EPI>), dim3(persistent_grid(nt)), dim3(512), R2_LDS, s, p);
}
template <typename TO>
static void launch_ring2(const NtArgs& p, int64_t M, int64_t N, hipStream_t s) {
  static const bool generic = getenv("FK_NT_RING2_GENERIC") != nullptr;    // tuning knob: one epilogue body for every mode
  if (generic) launch_ring2_epi<TO, -1>(p, M, N, s);
  else if (p.mode == 1) launch_ring2_epi<TO, 1>(p, M, N, s);
  else if (p.mode == 2) launch_ring2_epi<TO, 2>(p, M, N, s);
  else if (p.rope_table) launch_ring2_epi<TO, 3>(p, M, N, s);
  else launch_ring2_epi<TO, 0>(p, M, N, s);
}

template <typename TO, int BN_, int RB, int NS, int EPI>
static void launch_ring_epi(const NtArgs& p, int64_t M, int64_t N, hipStream_t s) {
  using R = Ring<BN_, RB, NS>;
  const int64_t nt = fk_cdiv(M, R::BM_) * (N / BN_);
  static bool once = (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_ring_kernel<TO, BN_, RB, NS, EPI>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, R::LDS) == hipSuccess);
  (void)once;
  hipLaunchKernelGGL((gemm_nt_ring_kernel<TO, BN_, RB, NS, EPI>), dim3(persistent_grid(nt)), dim3(512), R::LDS, s, p);
}
template <typename TO, int BN_, int RB, int NS>
static void launch_ring(const NtArgs& p, int64_t M, int64_t N, hipStream_t s) {   // one instantiation per fused epilogue mode
  if (p.mode == 1) launch_ring_epi<TO, BN_, RB, NS, 1>(p, M, N, s);
  else if (p.mode == 2) launch_ring_epi<TO, BN_, RB, NS, 2>(p, M, N, s);
  else if (p.rope_table) launch_ring_epi<TO, BN_, RB, NS, 3>(p, M, N, s);
  else launch_ring_epi<TO, BN_, RB, NS, 0>(p, M, N, s);
}

// ------------------------------------------------------------------------------------------------ TN
template <typename T> struct TNT;
template <> struct TNT<bf16_t> { static constexpr int BKM = 64, ROWB = 256, CHUNKS = 16, STEPS = 4; };
template <> struct TNT<float> { static constexpr int BKM = 32, ROWB = 512, CHUNKS = 32, STEPS = 2; };

FK_DEV int tn_off_bf16(int row, int bytecol) {   // bytecol: byte offset inside the 256-B row
  const int g = bytecol >> 6;
  return row * 256 + (((g ^ (row & 3)) << 6) | (bytecol & 63));
}

// fragment of k16-step s for output columns cb..cb+31 of a TN image (k index = image row)
template <typename T> FK_DEV void tn_frag(Frag<T>& f, const char* tile, int cb, int s, int lane);
template <> FK_DEV void tn_frag<bf16_t>(Frag<bf16_t>& f, const char* tile, int cb, int s, int lane) {
  const int g = lane >> 4, i = lane & 15, h = g >> 1;
  const int col = cb + 16 * (g & 1) + 4 * (i & 3);
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int row = 16 * s + 8 * h + 4 * t + (i >> 2);
    bf16x4 v = lds_read_tr4(reinterpret_cast<const bf16_t*>(tile + tn_off_bf16(row, col * 2)));
#pragma unroll
    for (int e = 0; e < 4; ++e) f.v[4 * t + e] = v[e];
  }
}
template <> FK_DEV void tn_frag<float>(Frag<float>& f, const char* tile, int cb, int s, int lane) {
  const int h = lane >> 5, c = cb + (lane & 31);
#pragma unroll
  for (int e = 0; e < 8; ++e)
    f.v[e] = *reinterpret_cast<const float*>(tile + (16 * s + 8 * h + e) * 512 + c * 4);
}

struct TnArgs {
  const void* A; const void* B; float* C; float* ws;
  int64_t lda, ldb, ldc;
  int M, N1, N2, rows_per_split, nsplit, accumulate;
};

template <typename T>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_tn_kernel(TnArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BKM = TNT<T>::BKM, ROWB = TNT<T>::ROWB, CHUNKS = TNT<T>::CHUNKS, STEPS = TNT<T>::STEPS;
  constexpr int VEC = KT<T>::VEC;
  constexpr int LOG_CH = (CHUNKS == 16) ? 4 : 5;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
  // 1-D grid, XCD-aware: logical id = (split, a-tile, b-tile) with b fastest, so one XCD owns whole M-splits: the dY slab
  // shared by the b-tiles and the X chunk shared by the a-tiles are fetched into that XCD's L2 once.
  const int nt2 = (p.N2 + BN - 1) / BN, ntiles = ((p.N1 + BM - 1) / BM) * nt2;
  const unsigned L = xcd_remap(blockIdx.x, gridDim.x);
  const int split = (int)(L / ntiles), tile = (int)(L % ntiles);
  const int a0 = (tile / nt2) * BM, b0 = (tile % nt2) * BN;
  const int mbeg = split * p.rows_per_split;
  const int mend = min(p.M, mbeg + p.rows_per_split);
  const T* A = (const T*)p.A;
  const T* B = (const T*)p.B;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  u32x4 ra[4], rb[4];
  auto gload = [&](int mrow0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int id = tid + NTHREADS * i, row = id >> LOG_CH, ch = id & (CHUNKS - 1);
      const int m = mrow0 + row, ca = a0 + ch * VEC, cb = b0 + ch * VEC;
      u32x4 z = {0u, 0u, 0u, 0u};
      ra[i] = (m < mend && ca < p.N1) ? *reinterpret_cast<const u32x4*>(A + (int64_t)m * p.lda + ca) : z;
      rb[i] = (m < mend && cb < p.N2) ? *reinterpret_cast<const u32x4*>(B + (int64_t)m * p.ldb + cb) : z;
    }
  };
  auto lds_off = [&](int row, int ch) -> int {
    if constexpr (sizeof(T) == 2) return tn_off_bf16(row, ch * 16);
    else return row * ROWB + ch * 16;
  };
  auto lstore = [&](int buf) {
    char* as = smem + buf * 2 * TILE_BYTES;
    char* bs = as + TILE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int id = tid + NTHREADS * i, row = id >> LOG_CH, ch = id & (CHUNKS - 1);
      *reinterpret_cast<u32x4*>(as + lds_off(row, ch)) = ra[i];
      *reinterpret_cast<u32x4*>(bs + lds_off(row, ch)) = rb[i];
    }
  };

  const int nk = (mend - mbeg + BKM - 1) / BKM;
  if (nk > 0) {
    gload(mbeg);
    lstore(0);
  }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) gload(mbeg + (kt + 1) * BKM);
    const char* as = smem + (kt & 1) * 2 * TILE_BYTES;
    const char* bs = as + TILE_BYTES;
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      Frag<T> fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) tn_frag<T>(fa[i], as, wm * 64 + i * 32, s, lane);
#pragma unroll
      for (int j = 0; j < 2; ++j) tn_frag<T>(fb[j], bs, wn * 64 + j * 32, s, lane);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) mma32<T>(acc[i][j], fa[i], fb[j]);
    }
    if (kt + 1 < nk) lstore((kt + 1) & 1);
    __syncthreads();
  }

  float* out = (p.nsplit > 1) ? p.ws + (int64_t)split * p.N1 * p.N2 : p.C;
  const int64_t ldo = (p.nsplit > 1) ? p.N2 : p.ldc;
  const bool accum = (p.nsplit == 1) && p.accumulate;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = b0 + wn * 64 + j * 32 + li;
    if (n >= p.N2) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = a0 + wm * 64 + i * 32 + acc_row(r, lh);
        if (m >= p.N1) continue;
        float* dst = out + (int64_t)m * ldo + n;
        *dst = accum ? (*dst + acc[i][j][r]) : acc[i][j][r];
      }
  }
}

// Large-tile weight-gradient kernel (bf16): 384 (A columns) x 128 (B columns) output tile per 8-wave block, 32 rows of m per
// stage, both operands brought in by LDS-DMA (32-KiB stages in a 4-slot ring, three in flight, one block per CU).  Why not the 128x128
// register-staged kernel above: its m-loop runs one 0.3 us k-tile ahead of loads that take 1-3 us, and every staged byte pays
// the ~79 B/clk ds_write path; here a stage carries 2 x 24 MFMAs per SIMD (~1 us) and nothing goes through ds_write.
// Images are [32 m][row bytes] (A 768 B, B 256 B) with the 64-B granule swizzle of tn_off_bf16 inside every 256-B group,
// applied on the DMA *source* column (the LDS side of an LDS-DMA is lane-linear).  Wave w: rows 96*(w>>1) of the tile's A
// columns (3 MFMA tiles) x columns 64*(w&1) (2 tiles).  Requires M % 64 == 0, N1 % 384 == 0, N2 % 128 == 0.
constexpr int TG_A = 384, TG_B = 128, TG_K = 32;               // 32-row k-stages (32 KiB)
#ifndef FK_TG_NS
#define FK_TG_NS 4
#endif
constexpr int TG_NS = FK_TG_NS;                           // ring slots: TG_NS - 1 stages in flight
constexpr int TG_A_ROW = TG_A * 2, TG_B_ROW = TG_B * 2;
constexpr int TG_A_BYTES = TG_K * TG_A_ROW, TG_B_BYTES = TG_K * TG_B_ROW, TG_STAGE = TG_A_BYTES + TG_B_BYTES;   // 24 + 8 KiB

template <int PITCH> FK_DEV int tg_off(int row, int bytecol) {
  return row * PITCH + (bytecol & ~255) + ((((bytecol >> 6) ^ row) & 3) << 6) + (bytecol & 63);
}
// ds_read_b64_tr_b16 as inline asm: hipcc orders the builtin form behind every pending LDS-DMA (s_waitcnt vmcnt(0) in front
// of the first fragment read of a k-tile), which serialises the prefetch with the MFMAs.  The asm form is invisible to that
// pass, so LDS completion is counted by hand (lgkm_wait) and DMA completion by the __syncthreads() that ends each k-tile.
template <int OFF> FK_DEV bf16x4 tr_read(unsigned base) {
  s16x4 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(base), "n"(OFF) : "memory");
  return __builtin_bit_cast(bf16x4, r);
}
template <int N> FK_DEV void lgkm_wait() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
  __builtin_amdgcn_sched_barrier(0);      // hipcc may otherwise hoist register-only MFMAs above the wait
}
template <int PITCH, int S> FK_DEV void tg_frag(Frag<bf16_t>& f, unsigned base) {
  const bf16x4 v0 = tr_read<(16 * S) * PITCH>(base), v1 = tr_read<(16 * S + 4) * PITCH>(base);
#pragma unroll
  for (int e = 0; e < 4; ++e) { f.v[e] = v0[e]; f.v[4 + e] = v1[e]; }
}
FK_DEV unsigned lds_addr(const char* p) {
  return (unsigned)(__UINTPTR_TYPE__)((__attribute__((address_space(3))) const char*)p);
}

// TB = columns of B per tile: 128 (256-byte image rows) or 192 (two column tiles instead of three for N2 = 384: the launch is bound by the
// bytes that go through LDS-DMA; image rows keep a 512-byte pitch so that the 64-byte-chunk swizzle stays inside a 256-byte
// block, the last 128 bytes of a row are padding that the requests fill with column 0).
template <int TB>
__global__ __launch_bounds__(512, 2) void gemm_tn_big_kernel(TnArgs p) {
  constexpr int B_ROW = TB == 128 ? 256 : 512, B_BYTES = TG_K * B_ROW, STAGE = TG_A_BYTES + B_BYTES, NJ = TB / 64, WB = TB / 2;
  using T = bf16_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
  const int nt2 = p.N2 / TB, ntiles = (p.N1 / TG_A) * nt2;
  const unsigned L = xcd_remap(blockIdx.x, gridDim.x);
  const int split = (int)(L / ntiles), tile = (int)(L % ntiles);
  const int a0 = (tile / nt2) * TG_A, b0 = (tile % nt2) * TB;
  const int mbeg = split * p.rows_per_split;
  const int mend = min(p.M, mbeg + p.rows_per_split);
  const int nk = (mend - mbeg) / TG_K;

  // per-lane DMA sources: piece q of an image = LDS bytes [1024 q, 1024 q + 1024), lane l -> 16 B at 1024 q + 16 l
  constexpr int APW = TG_A_BYTES / 8192, BPW = B_BYTES / 8192, PER = APW + BPW;     // 3 + 1 (or 2) LDS-DMA instructions per wave and stage
  const T* srcA[APW];
  const T* srcB[BPW];
#pragma unroll
  for (int j = 0; j < APW; ++j) {
    const int o = (wave * APW + j) * 1024 + lane * 16, row = o / TG_A_ROW, bc = o % TG_A_ROW;
    const int sc = (bc & ~255) + ((((bc >> 6) ^ row) & 3) << 6) + (bc & 63);
    srcA[j] = (const T*)p.A + (int64_t)(mbeg + row) * p.lda + a0 + sc / 2;
  }
#pragma unroll
  for (int j = 0; j < BPW; ++j) {
    const int o = (wave * BPW + j) * 1024 + lane * 16, row = o / B_ROW, bc = o % B_ROW;
    int sc = (bc & ~255) + ((((bc >> 6) ^ row) & 3) << 6) + (bc & 63);
    if (sc >= TB * 2) sc = 0;                                   // padding of the 512-byte pitch: any valid column
    srcB[j] = (const T*)p.B + (int64_t)(mbeg + row) * p.ldb + b0 + sc / 2;
  }
  const int64_t astep = (int64_t)TG_K * p.lda, bstep = (int64_t)TG_K * p.ldb;
  auto stage = [&](int buf) {
    char* as = smem + buf * STAGE;
#pragma unroll
    for (int j = 0; j < APW; ++j) {
      __builtin_amdgcn_global_load_lds((glb_void_t*)srcA[j], (lds_void_t*)(as + (wave * APW + j) * 1024), 16, 0, 0);
      srcA[j] += astep;
    }
#pragma unroll
    for (int j = 0; j < BPW; ++j) {
      __builtin_amdgcn_global_load_lds((glb_void_t*)srcB[j], (lds_void_t*)(as + TG_A_BYTES + (wave * BPW + j) * 1024), 16, 0, 0);
      srcB[j] += bstep;
    }
  };

  // fragment read bases (stage 0): lane (g = lane>>4, i = lane&15) reads row 8*(g>>1) + (i>>2) (+16 s + 4 t by immediate offset)
  unsigned fbase[3 + NJ];
  {
    const int g = lane >> 4, i = lane & 15, row = 8 * (g >> 1) + (i >> 2), dc = 16 * (g & 1) + 4 * (i & 3);
#pragma unroll
    for (int q = 0; q < 3; ++q) fbase[q] = lds_addr(smem) + tg_off<TG_A_ROW>(row, (wm * 96 + q * 32 + dc) * 2);
#pragma unroll
    for (int q = 0; q < NJ; ++q) fbase[3 + q] = lds_addr(smem) + TG_A_BYTES + tg_off<B_ROW>(row, (wn * WB + q * 32 + dc) * 2);
  }

  f32x16 acc[3][NJ];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  // The k-stages stream through TG_NS ring slots with counted waits (see gemm_nt_ring_kernel): stage kt+TG_NS-1 is issued when stage kt
  // starts computing; `s_waitcnt vmcnt((TG_NS-2) PER)` = this wave's part of stage kt has landed, the barrier = everybody's has and
  // everybody is done reading the slot refilled next.
#pragma unroll
  for (int s = 0; s < TG_NS - 1; ++s)
    if (s < nk) stage(s);
  int slot = 0;                                     // ring slot of the stage being computed
  for (int kt = 0; kt < nk; ++kt) {
    const int after = nk - 1 - kt;                  // stages behind this one: min(after, TG_NS - 2) of them are in flight
    if (after >= TG_NS - 2) vm_wait_barrier<(TG_NS - 2) * PER>();
    else if (after == 3) vm_wait_barrier<3 * PER>();
    else if (after == 2) vm_wait_barrier<2 * PER>();
    else if (after == 1) vm_wait_barrier<PER>();
    else vm_wait_barrier<0>();
    if (kt + TG_NS - 1 < nk) stage(slot == 0 ? TG_NS - 1 : slot - 1);
    const unsigned bo = slot * STAGE;
    slot = slot == TG_NS - 1 ? 0 : slot + 1;
    Frag<T> f0[3 + NJ], f1[3 + NJ];
#pragma unroll
    for (int q = 0; q < 3; ++q) tg_frag<TG_A_ROW, 0>(f0[q], fbase[q] + bo);
#pragma unroll
    for (int q = 0; q < NJ; ++q) tg_frag<B_ROW, 0>(f0[3 + q], fbase[3 + q] + bo);
#pragma unroll
    for (int q = 0; q < 3; ++q) tg_frag<TG_A_ROW, 1>(f1[q], fbase[q] + bo);
#pragma unroll
    for (int q = 0; q < NJ; ++q) tg_frag<B_ROW, 1>(f1[3 + q], fbase[3 + q] + bo);
    lgkm_wait<2 * (3 + NJ)>();                       // the first k16-step's fragments (LDS reads return in order)
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) mma32<T>(acc[i][j], f0[i], f0[3 + j]);
    lgkm_wait<0>();
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) mma32<T>(acc[i][j], f1[i], f1[3 + j]);
  }

  float* out = (p.nsplit > 1) ? p.ws + (int64_t)split * p.N1 * p.N2 : p.C;
  const int64_t ldo = (p.nsplit > 1) ? p.N2 : p.ldc;
  const bool accum = (p.nsplit == 1) && p.accumulate;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int n = b0 + wn * WB + j * 32 + li;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = a0 + wm * 96 + i * 32 + acc_row(r, lh);
        float* dst = out + (int64_t)m * ldo + n;
        *dst = accum ? (*dst + acc[i][j][r]) : acc[i][j][r];
      }
  }
}

// the same sum in the same order (slab 0, 1, 2, ...) on 16-byte vectors with eight slab loads in flight per thread
__global__ __launch_bounds__(256) void reduce_slabs4_kernel(const float* ws, float* C, int64_t ldc, int N1, int N2, int nsplit, int accumulate) {
  const int64_t total = (int64_t)N1 * N2, total4 = total >> 2;
  const int n2v = N2 >> 2;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
    const f32x4* src = reinterpret_cast<const f32x4*>(ws) + i;
    f32x4 s = {0.0f, 0.0f, 0.0f, 0.0f};
    int k = 0;
    for (; k + 8 <= nsplit; k += 8) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(k + u) * total4];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < nsplit; ++k) s += src[(int64_t)k * total4];
    f32x4* dst = reinterpret_cast<f32x4*>(C + (i / n2v) * ldc + (i % n2v) * 4);
    if (accumulate) s += *dst;
    *dst = s;
  }
}
__global__ void reduce_slabs_kernel(const float* ws, float* C, int64_t ldc, int N1, int N2, int nsplit, int accumulate) {
  const int64_t total = (int64_t)N1 * N2;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    float s = 0.0f;
    for (int k = 0; k < nsplit; ++k) s += ws[k * total + i];
    float* dst = C + (i / N2) * ldc + (i % N2);
    *dst = accumulate ? (*dst + s) : s;
  }
}

// ------------------------------------------------------------------------------------------------ colsum
// out[c] (+)= sum_r X[r, c]; two stages through a [nblk, cols] fp32 workspace (deterministic).
// Stage 1: a block owns rows_per_blk rows; thread = (16-byte column vector, row group): 16-byte loads, the row groups of a block are
// combined through LDS.  Stage 2: 64 columns x 4 partial groups per block.
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* X, int64_t ld, float* part, int rows, int cols, int rows_per_blk) {
  constexpr int N = 16 / sizeof(T);
  __shared__ float red[256 * N];
  const int r0 = blockIdx.y * rows_per_blk, r1 = min(rows, r0 + rows_per_blk);
  const int cvt = (cols + N - 1) / N;                     // column vectors in a row (cols % N == 0 on the vector path)
  {
    const int cv0 = blockIdx.x * 256;                    // grid.x walks 256-vector column chunks
    const int cvn = min(256, cvt - cv0), groups = 256 / cvn;
    const int vec = threadIdx.x % cvn, grp = threadIdx.x / cvn;
    float acc[N];
#pragma unroll
    for (int e = 0; e < N; ++e) acc[e] = 0.0f;
    if (grp < groups) {
      const T* p = X + (int64_t)(cv0 + vec) * N;
      for (int r = r0 + grp; r < r1; r += groups) {
        if constexpr (sizeof(T) == 2) {
          bf16x8 v = *reinterpret_cast<const bf16x8*>(p + (int64_t)r * ld);
#pragma unroll
          for (int e = 0; e < N; ++e) acc[e] += (float)v[e];
        } else {
          f32x4 v = *reinterpret_cast<const f32x4*>(p + (int64_t)r * ld);
#pragma unroll
          for (int e = 0; e < N; ++e) acc[e] += v[e];
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < N; ++e) red[threadIdx.x * N + e] = acc[e];
    __syncthreads();
    if (threadIdx.x < cvn) {
#pragma unroll
      for (int e = 0; e < N; ++e) {
        float sm = 0.0f;
        for (int g2 = 0; g2 < groups; ++g2) sm += red[(g2 * cvn + threadIdx.x) * N + e];       // fixed order: deterministic
        part[(int64_t)blockIdx.y * cols + (cv0 + threadIdx.x) * N + e] = sm;
      }
    }
  }
}
template <typename T>
__global__ void colsum_partial_scalar_kernel(const T* X, int64_t ld, float* part, int rows, int cols, int rows_per_blk) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  const int r0 = blockIdx.y * rows_per_blk, r1 = min(rows, r0 + rows_per_blk);
  float s = 0.0f;
  for (int r = r0; r < r1; ++r) s += to_f32<T>(X[(int64_t)r * ld + c]);
  part[(int64_t)blockIdx.y * cols + c] = s;
}
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* part, float* out, int cols, int nblk, int accumulate) {
  __shared__ float red[256];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
  float s = 0.0f;
  if (c < cols)
    for (int k = g; k < nblk; k += 4) s += part[(int64_t)k * cols + c];
  red[threadIdx.x] = s;
  __syncthreads();
  if (g == 0 && c < cols) {
    const float t = (red[threadIdx.x] + red[threadIdx.x + 64]) + (red[threadIdx.x + 128] + red[threadIdx.x + 192]);
    out[c] = accumulate ? out[c] + t : t;
  }
}

int tn_splits(int64_t M, int64_t N1, int64_t N2, int bkm) {
  const int64_t tiles = fk_cdiv(N1, BM) * fk_cdiv(N2, BN);
  int64_t want = fk_cdiv(1024, tiles);                     // ~4 blocks per CU
  const int64_t maxs = fk_cdiv(M, (int64_t)bkm * 4);        // >= 4 k-tiles per split
  if (want >= 8) want = (want + 7) / 8 * 8;                 // whole splits per XCD
  if (want > maxs) want = maxs;
  if (want < 1) want = 1;
  return (int)want;
}
// large-tile TN path: bf16, M % 64 == 0, N1 % 384 == 0, N2 % 128 == 0 and enough rows to be worth one block per CU
bool tn_big_ok(int64_t M, int64_t N1, int64_t N2, int dtype) {
  return dtype == FK_BF16 && M % TG_K == 0 && M >= 16384 && N1 % TG_A == 0 && (N2 % TG_B == 0 || N2 % 192 == 0);
}
// B columns per tile: 192 where it divides N2 and A is wide (fewer bytes through LDS-DMA: A is fetched N2 / TB times).  Measured on the
// cfg2 shapes: 3072 x 384 -7 %, 1152 x 384 and 384 x 1536 +-1 %, 384 x 384 +10 % (more split slabs, 25 % padding in the B requests) -> only
// for N1 >= 2048.
int tn_big_tb(int64_t N1, int64_t N2) {
  static const bool no192 = getenv("FK_TN_NO_192") != nullptr;
  if (N2 % TG_B != 0) return 192;
  return (N2 % 192 == 0 && N1 >= 2048 && !no192) ? 192 : TG_B;
}
int tn_big_splits(int64_t M, int64_t N1, int64_t N2) {
  const int64_t tiles = (N1 / TG_A) * (N2 / tn_big_tb(N1, N2));
  // one wave of blocks, one block per CU; FK_TN_SPLIT_MULT = m (default 1) cuts the contraction m times finer: m waves of shorter units,
  // so that a CU another kernel holds (a collective's channel) costs the launch 1/m of a unit, at the price of m times the slab traffic
  // (bench.py's launcher sets 2 for data-parallel runs; tools/occupant_probe.py measures both sides)
  static const int mult = [] { const char* e = getenv("FK_TN_SPLIT_MULT"); const int v = e ? atoi(e) : 1; return v < 1 ? 1 : (v > 8 ? 8 : v); }();
  int64_t want = 256 / tiles * mult;
  const int64_t maxs = M / 512;                             // >= 512 rows (16 k-stages) per split
  if (want > maxs) want = maxs;
  if (want < 1) want = 1;
  return (int)want;
}
int colsum_blocks(int64_t rows) {
  int64_t b = fk_cdiv(rows, 256);
  if (b > 512) b = 512;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

bool fk_qkv_rope_fused_ok(int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, const void* bias, int64_t T, int64_t D, int64_t rot_cols, int64_t q_cols, int dtype);
int fk_qkv_rope_fused_launch(const void* A, int64_t lda, const void* W, int64_t ldb, void* C, int64_t ldc, int64_t M, int64_t N, const float* table, int64_t table_bs, int64_t T, int64_t pos_off, int64_t rot_cols, int64_t q_cols, int64_t q_off, void* stream);
bool fk_mlp_up_fused_ok(int64_t M, int64_t H, int64_t K, int64_t lda, int64_t ldb, int64_t ldh, int64_t ldg, int dtype);      // mlp_fused.hip
int fk_mlp_up_fused_launch(const void* A, int64_t lda, const void* W13, int64_t ldb, void* H13, int64_t ldh, void* G, int64_t ldg, int64_t M, int64_t H, void* stream);

extern "C" {

struct RopeSpec { const float* table; int64_t bs; int T, off, D, cols; int qcols; int64_t qoff; };

static int launch_nt(const char* name, const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M,
                     int64_t N, int64_t K, const void* bias, const void* residual, int64_t ldr, int64_t res_rows, int dtype,
                     int out_dtype, int mode, void* aux, int64_t ldaux, void* stream, RopeSpec rope = RopeSpec{nullptr, 0, 1, 0, 2, 0, 0, 0}) {
  FK_CHECK_ARG(dtype == FK_F32 || dtype == FK_BF16, "%s: bad dtype %d", name, dtype);
  FK_CHECK_ARG(out_dtype == dtype || out_dtype == FK_F32, "%s: out_dtype must equal dtype or be f32", name);
  const int vec = dtype == FK_BF16 ? 8 : 4;
  FK_CHECK_ARG(M > 0 && N > 0 && K > 0, "%s: empty problem M=%lld N=%lld K=%lld", name, (long long)M, (long long)N, (long long)K);
  FK_CHECK_ARG(M < (1LL << 31) && N < (1LL << 31) && K < (1LL << 31), "%s: dims must fit int32", name);
  FK_CHECK_ARG(K % vec == 0 && lda % vec == 0 && ldb % vec == 0, "%s: K/lda/ldb must be multiples of %d", name, vec);
  FK_CHECK_ARG(A && B && C, "%s: null pointer", name);
  FK_CHECK_ARG(((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0, "%s: A/B must be 16-byte aligned", name);
  const int ovec = 8;   // epilogue handles 8 columns per lane
  const bool vec_epi = (N % ovec == 0) && (ldc % ovec == 0) && (((uintptr_t)C & 15) == 0) &&
                       (!residual || (ldr % ovec == 0 && ((uintptr_t)residual & 15) == 0));
  if (mode != 0)
    FK_CHECK_ARG(vec_epi && out_dtype == dtype && aux && ldaux % ovec == 0 && ((uintptr_t)aux & 15) == 0,
                 "%s: fused SwiGLU epilogue needs N %% 8 == 0 and 16-byte aligned, 8-element strided buffers", name);
  if (rope.table)
    FK_CHECK_ARG(vec_epi && rope.D % 8 == 0 && rope.cols % rope.D == 0 && rope.cols <= N && rope.T > 0 && M % rope.T == 0 &&
                 ((uintptr_t)rope.table & 15) == 0, "%s: fused RoPE needs the vector epilogue, D %% 8 == 0 and M %% T == 0", name);
  NtArgs p{A, B, C, bias, residual, lda, ldb, ldc, ldr, res_rows, (int)M, (int)N, (int)K, vec_epi ? 1 : 0, mode, aux, ldaux,
           rope.table, rope.bs, rope.T, rope.off, rope.D, rope.cols, rope.qcols, rope.qoff};
  const int64_t nwg = fk_cdiv(M, BM) * fk_cdiv(N, BN);
  dim3 grid((unsigned)nwg), block(NTHREADS);
  const size_t sh = 4 * TILE_BYTES;
  hipStream_t s = (hipStream_t)stream;
  const bool glds = dtype == FK_BF16 && (K % 64 == 0);
  static const bool no_ring = getenv("FK_NT_NO_RING") != nullptr;       // tuning knob: the double-buffered kernels instead
  // the persistent ring kernels want several 256-row tiles per CU; below half a tile per CU (SimpleMAE's 4800 visible-token rows at
  // B = 32) the 128 x 128 kernels fill the chip better: graphed cfg5 step 5.9 -> 5.6 ms (profiles/r03_g_other_configs.txt)
  // (read per call, not cached: the GPU suite switches it to send small shapes through the ring kernels)
  const char* ring_min_env = getenv("FK_NT_RING_MIN_TILES");
  const int ring_min = ring_min_env ? atoi(ring_min_env) : 128;
  const bool wide = glds && M >= 4096 && vec_epi && !no_ring && fk_cdiv(M, 256) * fk_cdiv(N, 256) >= ring_min;
  if (wide && (N % 256 == 0 || (N % 128 == 0 && N >= 1024))) {   // 256 x 256 tiles, split A/B rings (N = 1152: the last column tile is
                                                                  // half empty, still 4 % faster than 256 x 128 tiles)
    if (out_dtype == FK_BF16) launch_ring2<bf16_t>(p, M, N, s); else launch_ring2<float>(p, M, N, s);
    FK_CHECK_LAUNCH(name);
    return FK_OK;
  }
  static const bool no_192 = getenv("FK_NT_NO_192") != nullptr;        // tuning knob: 256 x 128 tiles for N = 384 / 768 too
  if (wide && N % 192 == 0 && N % 256 != 0 && N <= 768 && mode == 0 && !rope.table && !no_192) {   // N = 384: two 192-column tiles
    if (out_dtype == FK_BF16) launch_ring192<bf16_t>(p, M, N, s); else launch_ring192<float>(p, M, N, s);
    FK_CHECK_LAUNCH(name);
    return FK_OK;
  }
  if (wide && N % 128 == 0) {                  // N = 128 / 384 / 640 / 896: 256 x 128 ring
    if (out_dtype == FK_BF16) launch_ring<bf16_t, 128, 128, 3>(p, M, N, s); else launch_ring<float, 128, 128, 3>(p, M, N, s);
    FK_CHECK_LAUNCH(name);
    return FK_OK;
  }
  if (glds && M >= 4096 && N % 256 == 0 && vec_epi) {
        // large projections: 256 x 256 tiles, 1 block per CU (the 256x128
                                                                // variant measured slower than 128x128 at N = 384)
    const int64_t nt = fk_cdiv(M, 256) * (N / 256);
    dim3 bgrid(persistent_grid(nt)), bblock(512);
    const size_t bsh = 2 * (size_t)(256 + 256) * ROW_BYTES;
    if (out_dtype == FK_BF16) { static bool once = (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_big_kernel<bf16_t, 256>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072) == hipSuccess); (void)once;
      hipLaunchKernelGGL((gemm_nt_big_kernel<bf16_t, 256>), bgrid, bblock, bsh, s, p); }
    else { static bool once = (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_big_kernel<float, 256>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072) == hipSuccess); (void)once;
      hipLaunchKernelGGL((gemm_nt_big_kernel<float, 256>), bgrid, bblock, bsh, s, p); }
    FK_CHECK_LAUNCH(name);
    return FK_OK;
  }
  static const bool no_g4 = getenv("FK_NT_NO_GLDS4") != nullptr;       // tuning knob
  if (glds && nwg <= 256 && !no_g4) {                      // at most one tile per CU: the short-latency ring
    if (out_dtype == FK_BF16) {
      static bool once = (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_glds4_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, G4_LDS) == hipSuccess); (void)once;
      hipLaunchKernelGGL((gemm_nt_glds4_kernel<bf16_t>), grid, block, G4_LDS, s, p);
    } else {
      static bool once = (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_glds4_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, G4_LDS) == hipSuccess); (void)once;
      hipLaunchKernelGGL((gemm_nt_glds4_kernel<float>), grid, block, G4_LDS, s, p);
    }
    FK_CHECK_LAUNCH(name);
    return FK_OK;
  }
  dim3 pgrid((unsigned)(nwg < 512 ? nwg : 512));           // persistent: 2 blocks per CU (64 KiB LDS each)
  if (glds && out_dtype == FK_BF16) hipLaunchKernelGGL((gemm_nt_glds_kernel<bf16_t>), pgrid, block, sh, s, p);
  else if (glds) hipLaunchKernelGGL((gemm_nt_glds_kernel<float>), pgrid, block, sh, s, p);
  else if (dtype == FK_BF16 && out_dtype == FK_BF16) hipLaunchKernelGGL((gemm_nt_kernel<bf16_t, bf16_t>), grid, block, sh, s, p);
  else if (dtype == FK_BF16) hipLaunchKernelGGL((gemm_nt_kernel<bf16_t, float>), grid, block, sh, s, p);
  else hipLaunchKernelGGL((gemm_nt_kernel<float, float>), grid, block, sh, s, p);
  FK_CHECK_LAUNCH(name);
  return FK_OK;
}

int fk_gemm_nt(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M, int64_t N,
               int64_t K, const void* bias, const void* residual, int64_t ldr, int64_t res_rows, int dtype,
               int out_dtype, void* stream) {
  return launch_nt("fk_gemm_nt", A, lda, B, ldb, C, ldc, M, N, K, bias, residual, ldr, res_rows, dtype, out_dtype, 0, nullptr, 0, stream);
}

int fk_gemm_nt_rope(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M, int64_t N,
                    int64_t K, const void* bias, const float* table, int64_t table_bs, int64_t T, int64_t pos_off, int64_t D,
                    int64_t rot_cols, int64_t q_cols, int64_t q_table_off, int dtype, void* stream) {
  FK_CHECK_ARG(table && T > 0 && D > 0 && rot_cols >= 0, "fk_gemm_nt_rope: bad rope arguments");
  FK_CHECK_ARG(q_cols >= 0 && q_cols <= rot_cols && q_cols % 8 == 0 && q_table_off % 4 == 0, "fk_gemm_nt_rope: bad pre-scaled query table (q_cols %lld, offset %lld)", (long long)q_cols, (long long)q_table_off);
  if (A && B && C && (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C | (uintptr_t)table) & 15) == 0 && M > 0 && pos_off >= 0 &&
      fk_qkv_rope_fused_ok(M, N, K, lda, ldb, ldc, bias, T, D, rot_cols, q_cols, dtype))      // wide bf16 projections at d = 384, head_dim 64: the token-on-the-lane kernel (mlp_fused.hip), same bits
    return fk_qkv_rope_fused_launch(A, lda, B, ldb, C, ldc, M, N, table, table_bs, T, pos_off, rot_cols, q_cols, q_table_off, stream);
  return launch_nt("fk_gemm_nt_rope", A, lda, B, ldb, C, ldc, M, N, K, bias, nullptr, 0, 0, dtype, dtype, 0, nullptr, 0, stream,
                   RopeSpec{table, table_bs, (int)T, (int)pos_off, (int)D, (int)rot_cols, (int)q_cols, q_table_off});
}

int fk_gemm_nt_swiglu(const void* A, int64_t lda, const void* W13, int64_t ldb, void* H13, int64_t ldh, void* G,
                      int64_t ldg, int64_t M, int64_t H, int64_t K, int dtype, void* stream) {
  FK_CHECK_ARG(H > 0 && H % 4 == 0, "fk_gemm_nt_swiglu: hidden size must be a multiple of 4");
  if (A && W13 && H13 && G && (((uintptr_t)A | (uintptr_t)W13 | (uintptr_t)H13 | (uintptr_t)G) & 15) == 0 && M > 0 &&
      fk_mlp_up_fused_ok(M, H, K, lda, ldb, ldh, ldg, dtype))          // wide bf16 MLPs at d = 384: the token-on-the-lane kernel (mlp_fused.hip), same bits
    return fk_mlp_up_fused_launch(A, lda, W13, ldb, H13, ldh, G, ldg, M, H, stream);
  return launch_nt("fk_gemm_nt_swiglu", A, lda, W13, ldb, H13, ldh, M, 2 * H, K, nullptr, nullptr, 0, 0, dtype, dtype, 1, G, ldg, stream);
}

int fk_gemm_nt_dswiglu(const void* dY, int64_t lda, const void* W2T, int64_t ldb, const void* H13, int64_t ldh, void* dH13,
                       int64_t lddh, int64_t M, int64_t H, int64_t K, int dtype, void* stream) {
  FK_CHECK_ARG(H > 0 && H % 8 == 0, "fk_gemm_nt_dswiglu: hidden size must be a multiple of 8");
  FK_CHECK_ARG(lddh % 8 == 0 && ((uintptr_t)dH13 & 15) == 0, "fk_gemm_nt_dswiglu: dH13 must be 16-byte aligned with an 8-element stride");
  // C pointer/ld are the dh13 buffer (2H columns); the accumulator tile (dg) has H columns
  return launch_nt("fk_gemm_nt_dswiglu", dY, lda, W2T, ldb, dH13, lddh, M, H, K, nullptr, nullptr, 0, 0, dtype, dtype, 2,
                   const_cast<void*>(H13), ldh, stream);
}

size_t fk_gemm_tn_workspace_bytes(int64_t M, int64_t N1, int64_t N2, int dtype) {
  const int ns = tn_big_ok(M, N1, N2, dtype) ? tn_big_splits(M, N1, N2) : tn_splits(M, N1, N2, dtype == FK_BF16 ? 64 : 32);
  return ns > 1 ? (size_t)ns * N1 * N2 * sizeof(float) : 0;
}

int fk_gemm_tn(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int64_t N1,
               int64_t N2, int accumulate, int dtype, void* workspace, size_t workspace_bytes, void* stream) {
  FK_CHECK_ARG(dtype == FK_F32 || dtype == FK_BF16, "fk_gemm_tn: bad dtype %d", dtype);
  const int vec = dtype == FK_BF16 ? 8 : 4, bkm = dtype == FK_BF16 ? 64 : 32;
  FK_CHECK_ARG(M > 0 && N1 > 0 && N2 > 0, "fk_gemm_tn: empty problem");
  FK_CHECK_ARG(M < (1LL << 31) && N1 < (1LL << 31) && N2 < (1LL << 31), "fk_gemm_tn: dims must fit int32");
  FK_CHECK_ARG(N1 % vec == 0 && N2 % vec == 0 && lda % vec == 0 && ldb % vec == 0,
               "fk_gemm_tn: N1/N2/lda/ldb must be multiples of %d (N1=%lld N2=%lld)", vec, (long long)N1, (long long)N2);
  FK_CHECK_ARG(((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0, "fk_gemm_tn: A/B must be 16-byte aligned");
  const bool big = tn_big_ok(M, N1, N2, dtype);
  const int ns = big ? tn_big_splits(M, N1, N2) : tn_splits(M, N1, N2, bkm);
  const size_t need = ns > 1 ? (size_t)ns * N1 * N2 * sizeof(float) : 0;
  FK_CHECK_ARG(workspace_bytes >= need && (need == 0 || workspace), "fk_gemm_tn: workspace too small (%zu < %zu)", workspace_bytes, need);
  int64_t rps = fk_cdiv(fk_cdiv(M, ns), bkm) * bkm;
  TnArgs p{A, B, C, (float*)workspace, lda, ldb, ldc, (int)M, (int)N1, (int)N2, (int)rps, ns, accumulate};
  dim3 grid((unsigned)(fk_cdiv(N1, BM) * fk_cdiv(N2, BN) * ns)), block(NTHREADS);
  hipStream_t s = (hipStream_t)stream;
  if (big) {
    const int tb = tn_big_tb(N1, N2);
    dim3 bgrid((unsigned)((N1 / TG_A) * (N2 / tb) * ns));
    if (tb == 192) {
      constexpr int LDS192 = TG_NS * (TG_A_BYTES + TG_K * 512);
      static bool once = (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_big_kernel<192>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS192) == hipSuccess);
      (void)once;
      hipLaunchKernelGGL(gemm_tn_big_kernel<192>, bgrid, dim3(512), LDS192, s, p);
    } else {
      static bool once = (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_big_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize, TG_NS * TG_STAGE) == hipSuccess);
      (void)once;
      hipLaunchKernelGGL(gemm_tn_big_kernel<128>, bgrid, dim3(512), TG_NS * TG_STAGE, s, p);
    }
  } else if (dtype == FK_BF16) hipLaunchKernelGGL((gemm_tn_kernel<bf16_t>), grid, block, 4 * TILE_BYTES, s, p);
  else hipLaunchKernelGGL((gemm_tn_kernel<float>), grid, block, 4 * TILE_BYTES, s, p);
  FK_CHECK_LAUNCH("fk_gemm_tn");
  if (ns > 1) {
    const int64_t total = N1 * N2;
    if (N2 % 4 == 0 && ldc % 4 == 0 && (((uintptr_t)C | (uintptr_t)workspace) & 15) == 0) {
      int nb = (int)fk_cdiv(total / 4, 256);
      if (nb > 2048) nb = 2048;
      hipLaunchKernelGGL(reduce_slabs4_kernel, dim3(nb), dim3(256), 0, s, (const float*)workspace, C, ldc, (int)N1, (int)N2, ns, accumulate);
    } else {
      int nb = (int)fk_cdiv(total, 256);
      if (nb > 2048) nb = 2048;
      hipLaunchKernelGGL(reduce_slabs_kernel, dim3(nb), dim3(256), 0, s, (const float*)workspace, C, ldc, (int)N1, (int)N2, ns, accumulate);
    }
    FK_CHECK_LAUNCH("fk_gemm_tn(reduce)");
  }
  return FK_OK;
}

size_t fk_colsum_workspace_bytes(int64_t rows, int64_t cols) { return (size_t)colsum_blocks(rows) * cols * sizeof(float); }

int fk_colsum(const void* X, int64_t ld, float* out, int64_t rows, int64_t cols, int accumulate, int dtype,
              void* workspace, size_t workspace_bytes, void* stream) {
  FK_CHECK_ARG(dtype == FK_F32 || dtype == FK_BF16, "fk_colsum: bad dtype %d", dtype);
  FK_CHECK_ARG(rows > 0 && cols > 0 && rows < (1LL << 31) && cols < (1LL << 31), "fk_colsum: bad shape");
  const int nb = colsum_blocks(rows);
  FK_CHECK_ARG(workspace && workspace_bytes >= (size_t)nb * cols * sizeof(float), "fk_colsum: workspace too small");
  const int rpb = (int)fk_cdiv(rows, nb);
  hipStream_t s = (hipStream_t)stream;
  const int vec = dtype == FK_BF16 ? 8 : 4;
  if (cols % vec == 0 && ld % vec == 0 && ((uintptr_t)X & 15) == 0) {
    dim3 grid((unsigned)fk_cdiv(fk_cdiv(cols, vec), 256), (unsigned)nb), block(256);
    if (dtype == FK_BF16) hipLaunchKernelGGL(colsum_partial_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)X, ld, (float*)workspace, (int)rows, (int)cols, rpb);
    else hipLaunchKernelGGL(colsum_partial_kernel<float>, grid, block, 0, s, (const float*)X, ld, (float*)workspace, (int)rows, (int)cols, rpb);
  } else {
    dim3 grid((unsigned)fk_cdiv(cols, 256), (unsigned)nb), block(256);
    if (dtype == FK_BF16) hipLaunchKernelGGL(colsum_partial_scalar_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)X, ld, (float*)workspace, (int)rows, (int)cols, rpb);
    else hipLaunchKernelGGL(colsum_partial_scalar_kernel<float>, grid, block, 0, s, (const float*)X, ld, (float*)workspace, (int)rows, (int)cols, rpb);
  }
  FK_CHECK_LAUNCH("fk_colsum(partial)");
  hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)fk_cdiv(cols, 64)), dim3(256), 0, s, (const float*)workspace, out, (int)cols, nb, accumulate);
  FK_CHECK_LAUNCH("fk_colsum(final)");
  return FK_OK;
}

}  // extern "C"
