// gemm_tile.h — the 128 x 128 MFMA tile shared by gemm.hip and head_ce.hip: k-tile geometry, the swizzled LDS image of an operand tile
// and its fragment reads.  Included INSIDE the anonymous namespace of each translation unit.
#pragma once
constexpr int BM = 128, BN = 128, NTHREADS = 256;
constexpr int ROW_BYTES = 128;                 // NT image: bytes per k-tile row
constexpr int TILE_BYTES = BM * ROW_BYTES;     // 16 KiB per operand tile

template <typename T> struct KT;               // elements per k-tile
template <> struct KT<bf16_t> { static constexpr int BK = 64, VEC = 8, STEPS = 4; };
template <> struct KT<float> { static constexpr int BK = 32, VEC = 4, STEPS = 2; };

FK_DEV int nt_off(int row, int chunk) { return row * ROW_BYTES + ((chunk ^ ((row >> 1) & 7)) << 4); }

// fragment of k16-step s for tile row `row`, lane half h, from an NT image
template <typename T> FK_DEV void nt_frag(Frag<T>& f, const char* tile, int row, int s, int h);
template <> FK_DEV void nt_frag<bf16_t>(Frag<bf16_t>& f, const char* tile, int row, int s, int h) {
  f.v = *reinterpret_cast<const bf16x8*>(tile + nt_off(row, 2 * s + h));
}
template <> FK_DEV void nt_frag<float>(Frag<float>& f, const char* tile, int row, int s, int h) {
  f32x4 a = *reinterpret_cast<const f32x4*>(tile + nt_off(row, 4 * s + 2 * h));
  f32x4 b = *reinterpret_cast<const f32x4*>(tile + nt_off(row, 4 * s + 2 * h + 1));
#pragma unroll
  for (int e = 0; e < 4; ++e) { f.v[e] = a[e]; f.v[4 + e] = b[e]; }
}

