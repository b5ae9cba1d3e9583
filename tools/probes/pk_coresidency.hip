// pk_coresidency.hip — which instruction form returns wrong values beside which co-resident work?  
//
// Victim: one 2-D rotation per thread and iteration, o0 = x0 c - x1 s, o1 = x0 s + x1 c, written as inline asm in three forms:
//   SCALAR   v_mul_f32 / v_fma_f32
//   SWAPPED  what hipcc's SLP vectorizer emits for the scalar source: v_pk_mul_f32 with op_sel half-swaps feeding two v_pk_fma_f32,
//            each of which computes ONE useful half (the exact sequence of fk_rope / the RoPE epilogues before the fix)
//   PLAIN    packed arithmetic without half-swaps: (x0, x1) * (c, c) and (x1, x0) * (s, s) prepared by v_mov, then v_pk_fma_f32
// Occupants (second stream, 512-thread workgroups with 16 KiB of LDS so that they share CUs with the victim):
//   MFMA     v_mfma_f32_32x32x16_bf16 on register operands        TR     ds_read_b64_tr_b16 only
//   TR+MFMA  both (the inner loop of the small bf16 weight-gradient GEMM)   B128+MFMA  ds_read_b128 + MFMA     VALU   v_fma_f32 chain
// Every (form, occupant) pair: the victim runs REPS times beside the occupant; outputs are compared bit for bit with the quiet run of the
// same form; differing elements are counted per lane quarter and per output (o0 / o1).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/pk_coresidency.hip -o /tmp/pk_coresidency && /tmp/pk_coresidency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <int FORM>
__global__ __launch_bounds__(256) void victim(const float* __restrict__ x, const float* __restrict__ cs, float* __restrict__ out, int n, int iters) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (int it = 0; it < iters; ++it) {
    const int j = (i + it * 7919) % n;
    f32x2 xv = *reinterpret_cast<const f32x2*>(x + 2 * (size_t)j);      // (x0, x1)
    f32x2 cv = *reinterpret_cast<const f32x2*>(cs + 2 * (size_t)j);     // (c, s)
    float o0, o1;
    if (FORM == 0) {
      float t;
      asm volatile("v_mul_f32 %2, %4, %6\n\t"        // t  = x1 * s
                   "v_mul_f32 %1, %3, %6\n\t"        // o1 = x0 * s
                   "v_fma_f32 %0, %3, %5, -%2\n\t"   // o0 = x0 * c - t
                   "v_fma_f32 %1, %4, %5, %1\n\t"    // o1 = x1 * c + o1
                   : "=&v"(o0), "=&v"(o1), "=&v"(t)
                   : "v"(xv[0]), "v"(xv[1]), "v"(cv[0]), "v"(cv[1]));
    } else if (FORM == 1) {
      f32x2 sv = {cv[1], cv[1]}, t, a, b;
      // t = (s * x1, s * x0);  a.lo = c * x0 - t.lo;  b.hi = c * x1 + t.hi   (the other halves are by-products)
      asm volatile("v_pk_mul_f32 %0, %3, %4 op_sel:[0,1] op_sel_hi:[0,0]\n\t"
                   "v_pk_fma_f32 %1, %5, %4, %0 neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t"
                   "v_pk_fma_f32 %2, %5, %4, %0 op_sel_hi:[0,1,1]\n\t"
                   : "=&v"(t), "=&v"(a), "=&v"(b)
                   : "v"(sv), "v"(xv), "v"(cv));
      o0 = a[0];
      o1 = b[1];
    } else {
      f32x2 cc = {cv[0], cv[0]}, ss = {-cv[1], cv[1]}, xs = {xv[1], xv[0]}, t, r;
      asm volatile("v_pk_mul_f32 %0, %2, %3\n\t"        // t = (x1, x0) * (-s, s)
                   "v_pk_fma_f32 %1, %4, %5, %0\n\t"    // r = (x0, x1) * (c, c) + t
                   : "=&v"(t), "=&v"(r)
                   : "v"(xs), "v"(ss), "v"(xv), "v"(cc));
      o0 = r[0];
      o1 = r[1];
    }
    out[2 * (size_t)i] = o0;
    out[2 * (size_t)i + 1] = o1;
    if (it + 1 < iters) asm volatile("" ::"v"(o0), "v"(o1));
  }
}

template <int KIND>
__global__ __launch_bounds__(512, 2) void occupant(float* sink, int iters) {
  __shared__ __attribute__((aligned(16))) char lds[16384];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 4096; i += 512) reinterpret_cast<float*>(lds)[i] = (float)(i & 255) * 0.001f;
  __syncthreads();
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.01f * (lane + e)); b[e] = (__bf16)(0.02f * (lane - e)); }
  float v = 1.0f + lane * 1e-3f;
  const char* p = lds + (lane & 15) * 128 + (lane >> 4) * 8;
  for (int it = 0; it < iters; ++it) {
    if (KIND == 0 || KIND == 2 || KIND == 3) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    if (KIND == 1 || KIND == 2) {
      s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(__attribute__((address_space(3))) void*)(p + (it & 7) * 2048));
      a[0] = __builtin_bit_cast(__bf16, t[0]);
      a[3] = __builtin_bit_cast(__bf16, t[3]);
    }
    if (KIND == 3) {
      bf16x8 t = *reinterpret_cast<const bf16x8*>(lds + ((lane * 16 + (it & 7) * 1024) & 16383));
      b[1] = t[1];
      b[6] = t[6];
    }
    if (KIND == 4) {
#pragma unroll
      for (int q = 0; q < 8; ++q) v = __builtin_fmaf(v, 1.0001f, 0.25f);
    }
  }
  float s = v;
  for (int r = 0; r < 16; ++r) s += acc[r];
  if (s == 123.456f) sink[tid] = s;
}

int main() {
  const int n = 1 << 22, iters = 24, REPS = 6;
  std::vector<float> hx(2 * (size_t)n), hc(2 * (size_t)n);
  srand(1);
  for (size_t i = 0; i < hx.size(); ++i) hx[i] = (float)rand() / (float)RAND_MAX - 0.5f;
  for (int i = 0; i < n; ++i) { const float ang = 6.2831853f * (float)rand() / (float)RAND_MAX; hc[2 * (size_t)i] = cosf(ang); hc[2 * (size_t)i + 1] = sinf(ang); }
  float *dx, *dc, *dout, *dsink;
  CK(hipMalloc(&dx, hx.size() * 4)); CK(hipMalloc(&dc, hc.size() * 4)); CK(hipMalloc(&dout, hx.size() * 4)); CK(hipMalloc(&dsink, 4096));
  CK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dc, hc.data(), hc.size() * 4, hipMemcpyHostToDevice));
  hipStream_t sa, sb;
  CK(hipStreamCreate(&sa)); CK(hipStreamCreate(&sb));
  std::vector<float> base(hx.size()), got(hx.size());
  const char* forms[3] = {"SCALAR (v_mul / v_fma)", "SWAPPED (v_pk_mul op_sel -> v_pk_fma: the SLP form)", "PLAIN (v_pk_mul / v_pk_fma, no half-swaps)"};
  const char* occs[5] = {"MFMA", "TR", "TR+MFMA", "B128+MFMA", "VALU"};
  for (int f = 0; f < 3; ++f) {
    auto run_victim = [&]() {
      const dim3 g((n + 255) / 256), b(256);
      if (f == 0) hipLaunchKernelGGL(victim<0>, g, b, 0, sa, dx, dc, dout, n, iters);
      else if (f == 1) hipLaunchKernelGGL(victim<1>, g, b, 0, sa, dx, dc, dout, n, iters);
      else hipLaunchKernelGGL(victim<2>, g, b, 0, sa, dx, dc, dout, n, iters);
    };
    run_victim();
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(base.data(), dout, base.size() * 4, hipMemcpyDeviceToHost));
    // the three forms agree with a double evaluation to fp32 rounding (sanity of the asm)
    double worst = 0.0;
    for (int i = 0; i < n; i += 97) {
      const int j = (i + (iters - 1) * 7919) % n;
      const double x0 = hx[2 * (size_t)j], x1 = hx[2 * (size_t)j + 1], c = hc[2 * (size_t)j], s = hc[2 * (size_t)j + 1];
      worst = fmax(worst, fabs(base[2 * (size_t)i] - (x0 * c - x1 * s)));
      worst = fmax(worst, fabs(base[2 * (size_t)i + 1] - (x0 * s + x1 * c)));
    }
    printf("== victim %s: quiet max |err| vs double %.2e\n", forms[f], worst);
    for (int k = 0; k < 5; ++k) {
      long bad = 0, quarter[4] = {0, 0, 0, 0}, which[2] = {0, 0};
      int runs_bad = 0;
      for (int rep = 0; rep < REPS; ++rep) {
        const dim3 og(2048), ob(512);
        const int oit = 60000;
        if (k == 0) hipLaunchKernelGGL(occupant<0>, og, ob, 0, sb, dsink, oit);
        else if (k == 1) hipLaunchKernelGGL(occupant<1>, og, ob, 0, sb, dsink, oit);
        else if (k == 2) hipLaunchKernelGGL(occupant<2>, og, ob, 0, sb, dsink, oit);
        else if (k == 3) hipLaunchKernelGGL(occupant<3>, og, ob, 0, sb, dsink, oit);
        else hipLaunchKernelGGL(occupant<4>, og, ob, 0, sb, dsink, oit * 4);
        run_victim();
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(got.data(), dout, got.size() * 4, hipMemcpyDeviceToHost));
        long b0 = 0;
        for (size_t e = 0; e < got.size(); ++e)
          if (memcmp(&got[e], &base[e], 4) != 0) { ++b0; ++quarter[((e / 2) % 64) / 16]; ++which[e & 1]; }
        bad += b0;
        runs_bad += b0 > 0;
      }
      printf("   beside %-10s: %d / %d runs differ, %ld elements; by lane quarter [%ld %ld %ld %ld]; o0 %ld, o1 %ld\n", occs[k], runs_bad, REPS, bad,
             quarter[0], quarter[1], quarter[2], quarter[3], which[0], which[1]);
    }
  }
  return 0;
}
