// Dense remainder of a trained NodeEdge layer (gfx950).
//
// The reference's NodeEdge is `(w * inci + b) @ x` with w and b DENSE trainable (R x M) matrices (emulator.py:34-45).  On
// the incidence support the product is the fused kernel's CSR aggregation; a trained bias is non-zero everywhere else
// too, and that part -- `rest @ x`, rest = b with the support zeroed -- is a true dense GEMM:
//     rem[s][r][f] = sum_m rest[r][m] * x[s][m][f]          R x M  times  M x (S * h)
// (R, M = nodes / links, h = d/2 = 32 features, S snapshots: 2 * R * M * S * h flops, 0.46 TFLOP per side at the headline
// sizes -- ten times the rest of the layer; the reference pays it on every layer, trained or not).
//
// Here: split-bf16 MFMA as everywhere else (operands a = hi + lo in bf16, hi*hi + lo*hi + hi*lo accumulated in fp32, ~2^-16
// per product).  `rest` is split ONCE per parameter update (k_split_rows_bf16: hi and lo planes, K padded to 32); the
// activations are split and TRANSPOSED per call into (S*h) x M planes (k_split_transpose_bf16), so that both MFMA operands
// read 8 consecutive k for one row / one column with one 16-byte load.  Kernel: 128 x 128 output tile per 256-thread
// workgroup (4 waves, 64 x 64 each), k-steps of 32 through a double-buffered LDS image (80-byte rows: conflict-free
// ds_read_b128), the next k-step's global loads in flight while the current one is multiplied, one barrier per step.  The
// activations are the MFMA A operand, so that a lane ends up with 4 consecutive features of one output row (16-byte stores
// into (S, R, h)).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels_fused.hpp"

namespace uds {

constexpr int GEMM_BM = 128, GEMM_BN = 128, GEMM_BK = 64, GEMM_LDS_ROW = 64;      // LDS rows of 64 bf16 (128 bytes), XOR-swizzled

// fp32 (rows x K) -> bf16 hi / lo planes (rows x Kp), Kp = K rounded up to 64, zero padded
__global__ void k_split_rows_bf16(const float *__restrict__ a, int64_t rows, int64_t K, int64_t Kp, __bf16 *__restrict__ hi,
                                  __bf16 *__restrict__ lo) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // one thread per 8 consecutive k
  const int64_t per_row = Kp / 8;
  if (i >= rows * per_row) return;
  const int64_t r = i / per_row, k0 = (i - r * per_row) * 8;
  bf16x8 h, l;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float v = k0 + j < K ? a[r * K + k0 + j] : 0.f;
    const __bf16 hh = (__bf16)v;
    h[j] = hh;
    l[j] = (__bf16)(v - (float)hh);
  }
  *reinterpret_cast<bf16x8 *>(hi + r * Kp + k0) = h;
  *reinterpret_cast<bf16x8 *>(lo + r * Kp + k0) = l;
}

// x (S, M, h) fp32 -> bf16 hi / lo planes (S * h) x Mp with row (s, f) = x[s][:, f] (transposed), zero padded to Mp.
// One workgroup per (s, 64 values of m): reads 64 x h floats coalesced, writes h runs of 64 bf16 (128 bytes).
__global__ __launch_bounds__(256) void k_split_transpose_bf16(const float *__restrict__ x, int64_t M, int h, int64_t Mp,
                                                              __bf16 *__restrict__ hi, __bf16 *__restrict__ lo) {
  __shared__ float tile[64][65];      // h <= 64
  const int64_t s = blockIdx.y, m0 = (int64_t)blockIdx.x * 64;
  const float *src = x + (s * M + m0) * h;
  for (int i = threadIdx.x; i < 64 * h; i += 256) {
    const int m = i / h, f = i - m * h;
    tile[m][f] = m0 + m < M ? src[(int64_t)m * h + f] : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * h; i += 256) {
    const int f = i / 64, m = i - f * 64;
    if (m0 + m >= Mp) continue;
    const float v = tile[m][f];
    const __bf16 hh = (__bf16)v;
    hi[(s * h + f) * Mp + m0 + m] = hh;
    lo[(s * h + f) * Mp + m0 + m] = (__bf16)(v - (float)hh);
  }
}

// out[s][r][f] = sum_k X[(s, f)][k] * W[r][k]:  X = transposed activations (Nc x Kp, Nc = S * h), W = rest (R x Kp), both
// as hi / lo bf16 planes, Kp % 64 == 0.  grid = n_ctile * ceil(R / 128) workgroups, n_ctile = ceil(Nc / 128).
// Per k-step of 64: [barrier] staged registers -> LDS [barrier] next step's 16 global loads issued, then 2 x (16 fragment
// reads + 48 MFMAs) -- the loads have a whole step of matrix work (and the other workgroup of the CU) to land.
__global__ __launch_bounds__(256, 2) void k_remainder_gemm(const __bf16 *__restrict__ Xh, const __bf16 *__restrict__ Xl,
                                                          const __bf16 *__restrict__ Wh, const __bf16 *__restrict__ Wl, int64_t Nc, int64_t R,
                                                          int64_t Kp, int h, int n_ctile, float *__restrict__ out) {
  __shared__ __attribute__((aligned(16))) __bf16 lds[4][GEMM_BM * GEMM_LDS_ROW];      // [Xh, Xl, Wh, Wl][128 rows x 64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;                   // wave tile: X rows (columns of the result) 64 wr.., W rows 64 wc..
  // Workgroups b, b + 8, b + 16 ... share an XCD (and its 4 MB L2): give each XCD a contiguous run of tiles, column tile
  // fastest, so that the ~64 tiles an XCD has in flight are a few row tiles of W times all column tiles of X -- every
  // k-step's W and X pieces are then fetched into that L2 once and served to the 4-15 workgroups that use them
  const int nblk = gridDim.x, b = blockIdx.x, q8 = nblk / 8, r8 = nblk % 8, xcd = b % 8;
  const int w = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + b / 8;
  const int64_t c0 = (int64_t)(w % n_ctile) * GEMM_BM, r0 = (int64_t)(w / n_ctile) * GEMM_BN;
  // global -> register staging: the 128 x 64 bf16 tile of a plane is 1024 16-byte chunks, four per thread and plane:
  // rows tid / 8 + 32 i, 8 bf16 at k offset 8 (tid % 8) -- eight threads read 128 contiguous bytes of a row
  const int ch_row = tid >> 3, ch_k = (tid & 7) * 8;
  // LDS image: the 16-byte chunk c of row r sits at chunk c ^ (r & 7) -- the four 16-lane groups of a ds_read_b128 fragment
  // read (rows lane & 15, chunk lane >> 4) then touch every bank once (plain 128-byte rows: 4-way; padded to 144: 2-way)
  const int ch_sw = (((tid & 7) ^ (ch_row & 7)) * 8);
  int64_t xoff[4], woff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {       // rows past the matrix are clamped (their results are not stored)
    xoff[i] = min(c0 + ch_row + 32 * i, Nc - 1) * Kp + ch_k;
    woff[i] = min(r0 + ch_row + 32 * i, R - 1) * Kp + ch_k;
  }
  uint4 a0, a1, a2, a3, b0, b1, b2, b3, c0_, c1_, c2_, c3_, d0, d1, d2, d3;      // named scalars: an array here ends up in scratch
#define UDS_GEMM_LOAD1(i, A, B, C, D, k0)                           \
  A = *reinterpret_cast<const uint4 *>(Xh + xoff[i] + (k0));        \
  B = *reinterpret_cast<const uint4 *>(Xl + xoff[i] + (k0));        \
  C = *reinterpret_cast<const uint4 *>(Wh + woff[i] + (k0));        \
  D = *reinterpret_cast<const uint4 *>(Wl + woff[i] + (k0));
#define UDS_GEMM_LOAD(k0)                                           \
  UDS_GEMM_LOAD1(0, a0, b0, c0_, d0, k0) UDS_GEMM_LOAD1(1, a1, b1, c1_, d1, k0) UDS_GEMM_LOAD1(2, a2, b2, c2_, d2, k0) UDS_GEMM_LOAD1(3, a3, b3, c3_, d3, k0)
#define UDS_GEMM_STORE1(i, A, B, C, D)                              \
  {                                                                 \
    const int o = (ch_row + 32 * i) * GEMM_LDS_ROW + ch_sw;         \
    *reinterpret_cast<uint4 *>(&lds[0][o]) = A;                     \
    *reinterpret_cast<uint4 *>(&lds[1][o]) = B;                     \
    *reinterpret_cast<uint4 *>(&lds[2][o]) = C;                     \
    *reinterpret_cast<uint4 *>(&lds[3][o]) = D;                     \
  }
#define UDS_GEMM_STORE UDS_GEMM_STORE1(0, a0, b0, c0_, d0) UDS_GEMM_STORE1(1, a1, b1, c1_, d1) UDS_GEMM_STORE1(2, a2, b2, c2_, d2) UDS_GEMM_STORE1(3, a3, b3, c3_, d3)
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15;                                 // fragment: row / column lane & 15, 8 consecutive k at 8 * (lane >> 4)
  UDS_GEMM_LOAD(0)
  const int64_t n_k = Kp / GEMM_BK;
  for (int64_t kt = 0; kt < n_k; ++kt) {
    __syncthreads();                                         // every wave has read the previous step's fragments
    UDS_GEMM_STORE
    __syncthreads();
    if (kt + 1 < n_k) { UDS_GEMM_LOAD((kt + 1) * GEMM_BK) }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      bf16x8 xh[4], xl[4], wh[4], wl[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int sw = (((half * 4 + (lane >> 4)) ^ (fr & 7)) * 8);       // rows 16 i + fr: (row & 7) = (fr & 7)
        const int xo = (wr * 64 + i * 16 + fr) * GEMM_LDS_ROW + sw, wo = (wc * 64 + i * 16 + fr) * GEMM_LDS_ROW + sw;
        xh[i] = *reinterpret_cast<const bf16x8 *>(&lds[0][xo]);
        xl[i] = *reinterpret_cast<const bf16x8 *>(&lds[1][xo]);
        wh[i] = *reinterpret_cast<const bf16x8 *>(&lds[2][wo]);
        wl[i] = *reinterpret_cast<const bf16x8 *>(&lds[3][wo]);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mfma3(xh[i], xl[i], wh[j], wl[j], acc[i][j]);
    }
  }
  // D[row = 4 (lane >> 4) + q][col = lane & 15] of tile (i, j): result column c = c0 + 64 wr + 16 i + 4 (lane >> 4) + q
  // (= snapshot c / h, feature c % h), result row r = r0 + 64 wc + 16 j + (lane & 15): four consecutive features per lane
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t c = c0 + wr * 64 + i * 16 + (lane >> 4) * 4;
    if (c >= Nc) continue;
    const int64_t s = c / h, f = c - s * h;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t r = r0 + wc * 64 + j * 16 + (lane & 15);
      if (r < R) *reinterpret_cast<f32x4 *>(out + (s * R + r) * h + f) = acc[i][j];
    }
  }
}

#undef UDS_GEMM_LOAD
#undef UDS_GEMM_LOAD1
#undef UDS_GEMM_STORE
#undef UDS_GEMM_STORE1

// ---- second form of the same GEMM (round 3): LDS-DMA staging, 256-row block tiles, 8 waves ------------------------------------
// What bounded k_remainder_gemm (128 x 128, register staging): every k-step moved 4 planes x 16 KB from L2 through 64 VGPRs and
// 16 ds_write_b128 per thread into LDS -- 0.0104 B of L2 traffic per MFMA flop (10 TB/s at its 0.96 PF, against the 17-19 TB/s
// the L2s deliver into LDS, MI355X_MICROARCH.md 'Indexed rows'), an LDS store for every LDS load byte, two barriers per step.
// Here: block tile 256 (X rows) x BN (W rows, 256 or 128) per 512-thread workgroup; k-steps of 32 fetched by
// global_load_lds_dwordx4 straight into a double-buffered LDS image (no staging registers, no LDS stores); one barrier per
// step; wave tile (256 / WRN) x 64.  L2 bytes per flop: (2/3) (1/256 + 1/BN) = 0.0052 (BN 256) / 0.0078 (BN 128).
// LDS image of one buffer: [Xh | Xl | Wh | Wl] planes of 16-row blocks of 1 KiB (a row = 32 bf16 = 4 chunks of 16 B); chunk c of
// row r sits at position c ^ F[(r & 15) >> 2], F = {0, 3, 2, 1}: the 16 lanes ds_read_b128 serves per LDS cycle -- rows 0-3 and
// 12-15 of k-chunk q, rows 4-11 of chunk q ^ 1 -- then cover the sixteen 16-B slots of the 256-B bank row once.  A DMA piece is
// a block: lane l fetches the chunk that belongs at position l & 3 of row l >> 2 (the swizzle is applied on the GLOBAL side:
// LDS-DMA writes lane l's 16 bytes at piece + 16 l).
template <int WRN>
struct Gemm2Cfg {
  static constexpr int BM = 256, NWC = 8 / WRN, BN = 64 * NWC;              // waves: WRN along X rows x NWC along W rows
  static constexpr int TI = BM / WRN / 16;                                   // 16-row X tiles per wave (8, 4 or 2); W tiles per wave: 4
  static constexpr int XI = TI < 4 ? TI : 4;                                 // X tiles whose fragments are held at a time
  static constexpr int X_PLANE = BM * 64, W_PLANE = BN * 64;                 // bytes of one plane of one buffer
  static constexpr int BUF = 2 * X_PLANE + 2 * W_PLANE;                      // 64 KiB (BN 256) / 48 KiB (BN 128)
  static constexpr int LDS_BYTES = 2 * BUF;
};

__device__ __forceinline__ void gemm2_dma(const __bf16 *base, unsigned voff, unsigned lds_byte) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(base), "s"(lds_byte)
               : "memory");
}

// M32: the wave tile as 4 x 2 tiles of 32 x 32 on v_mfma_f32_32x32x16_bf16 (two per 32-wide k-step) instead of 8 x 4 tiles of 16 x 16 on
// v_mfma_f32_16x16x32_bf16: the same matrix-pipe cycles and LDS reads, half the MFMA instructions and half the operand-register reads
// per flop (WRN = 2 only).  Lane l: row l & 31 of the tile, k half l >> 5; accumulator register 4 g + e <-> X row 8 g + 4 (l >> 5) + e.
template <int WRN, bool M32 = false>
__global__ __launch_bounds__(512) void k_remainder_gemm2(const __bf16 *__restrict__ Xh, const __bf16 *__restrict__ Xl,
                                                         const __bf16 *__restrict__ Wh, const __bf16 *__restrict__ Wl, int64_t Nc, int64_t R,
                                                         int64_t Kp, int h, int n_ctile, float *__restrict__ out, int tile_base, int ksplit,
                                                         float *__restrict__ partial) {
  // The tiles that fill whole rounds of 256 workgroups are computed whole (results straight to `out`); the remaining ones are cut
  // along K into ksplit pieces each so that they fill one more (partial) round: their accumulators go to `partial` in register
  // order and k_remainder_gemm2_reduce adds the pieces in a fixed order (uds_remainder_forward).
  using C = Gemm2Cfg<WRN>;
  constexpr int64_t r_base = 0;
  const int64_t r_lim = R;
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds2[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / C::NWC, wc = wave % C::NWC;
  // workgroups [0, tile_base): whole tiles, straight to `out`; [tile_base, gridDim.x): the cut tiles, ksplit pieces each, in the
  // SAME launch so that they start as the whole tiles finish (two launches serialise on the first one's ragged end).
  // XCD-aware order inside each part as in k_remainder_gemm (tile_base is a multiple of 256).
  const bool cut = (int)blockIdx.x >= tile_base;
  const int nblk = cut ? (int)gridDim.x - tile_base : tile_base, b = cut ? (int)blockIdx.x - tile_base : (int)blockIdx.x;
  const int q8 = nblk / 8, r8 = nblk % 8, xcd = b % 8;
  const int wi = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + b / 8;
  const int w = cut ? tile_base + wi / ksplit : wi, piece = cut ? wi % ksplit : 0;
  if (!cut) ksplit = 1;
  const int64_t c0 = (int64_t)(w % n_ctile) * C::BM, r0 = r_base + (int64_t)(w / n_ctile) * C::BN;
  // DMA duty of this wave: X blocks 2 wave, 2 wave + 1 and W blocks NWB wave + {0, ..} (while there are any) of both planes
  constexpr int NXB = 2, WBLK = C::BN / 16, NWB = (WBLK + 7) / 8;
  const int d_row = lane >> 2, d_pos = lane & 3;
  const int d_chunk = d_pos ^ ((4 - (lane >> 4)) & 3);                     // F[(row & 15) >> 2] with row = lane >> 2
  unsigned xvoff[NXB], wvoff[NWB];
#pragma unroll
  for (int i = 0; i < NXB; ++i)      // rows past the matrix are clamped (their results are not stored); host: Nc * Kp * 2 < 2^32
    xvoff[i] = (unsigned)(min(c0 + 16 * (NXB * wave + i) + d_row, Nc - 1) * Kp * 2 + d_chunk * 16);
#pragma unroll
  for (int i = 0; i < NWB; ++i)
    wvoff[i] = (unsigned)(min(r0 + 16 * (NWB * wave + i) + d_row, r_lim - 1) * Kp * 2 + d_chunk * 16);
  const unsigned lds0 = lds_addr(lds2);
  // part p of this wave's DMA duty for k-step kt: p < NXB -> X block p (both planes), else W block p - NXB
  auto issue_part = [&](int64_t kt, int p) __attribute__((always_inline)) {
    const unsigned buf = lds0 + (unsigned)(kt & 1) * C::BUF;
    const int64_t ko = kt * 32;                                             // bf16 elements
    if (p < NXB) {
      gemm2_dma(Xh + ko, xvoff[p], buf + (NXB * wave + p) * 1024);
      gemm2_dma(Xl + ko, xvoff[p], buf + C::X_PLANE + (NXB * wave + p) * 1024);
    } else if (p - NXB < NWB && NWB * wave + (p - NXB) < WBLK) {
      gemm2_dma(Wh + ko, wvoff[p - NXB], buf + 2 * C::X_PLANE + (NWB * wave + p - NXB) * 1024);
      gemm2_dma(Wl + ko, wvoff[p - NXB], buf + 2 * C::X_PLANE + C::W_PLANE + (NWB * wave + p - NXB) * 1024);
    }
  };
  auto issue = [&](int64_t kt) __attribute__((always_inline)) {
#pragma unroll
    for (int p = 0; p < NXB + NWB; ++p) issue_part(kt, p);
  };
  static_assert(!M32 || WRN == 2, "the 32 x 32 form is written for the 128 x 64 wave tile");
  typedef float f32x16 __attribute__((ext_vector_type(16)));
  f32x4 acc[M32 ? 1 : C::TI][4];
  f32x16 acc32[M32 ? 4 : 1][2];
  if constexpr (M32) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc32[i][j][r] = 0.f;
  } else {
#pragma unroll
    for (int i = 0; i < C::TI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int fr = lane & 15, q = lane >> 4;
  const int f_off = fr * 64 + ((q ^ ((4 - (fr >> 2)) & 3)) << 4);          // this lane's 16 bytes inside a 1-KiB block
  const unsigned char *xbase = lds2 + (C::TI * wr) * 1024 + f_off;
  const unsigned char *wbase = lds2 + 2 * C::X_PLANE + (4 * wc) * 1024 + f_off;
  const int64_t n_k_all = Kp / 32, k_beg = n_k_all * piece / ksplit, n_k = n_k_all * (piece + 1) / ksplit;
  // (buffer = parity of the absolute k-step: consistent between issue() and the reads below)
  issue(k_beg);
  for (int64_t kt = k_beg; kt < n_k; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's pieces of step kt have landed
    __syncthreads();                                         // ... everybody's have, and everybody is done reading the other buffer
    // The next step's pieces are issued BEHIND the first tile rows' MFMAs, one part per tile row: a DMA instruction holds the
    // wave at issue for 100+ cycles under load, and issued right here (as at first) eight of them kept the matrix pipe idle at
    // the start of every step; behind the MFMAs the pipe works through them meanwhile.
    const bool more = kt + 1 < n_k;
    const int bo = (int)(kt & 1) * C::BUF;
    if constexpr (M32) {
      const int n32 = lane & 31, hf = lane >> 5;
      const int off0 = (n32 >> 4) * 1024 + fr * 64 + ((hf ^ ((4 - (fr >> 2)) & 3)) << 4);      // k chunk hf of row n32; chunk 2 + hf: ^ 32
      const unsigned char *xb = lds2 + bo + (8 * wr) * 1024, *wb = lds2 + bo + 2 * C::X_PLANE + (4 * wc) * 1024;
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) {
        const int off = off0 ^ (kh << 5);
        bf16x8 wh2[2], wl2[2], xh4[4], xl4[4];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          wh2[j] = *reinterpret_cast<const bf16x8 *>(wb + j * 2048 + off);
          wl2[j] = *reinterpret_cast<const bf16x8 *>(wb + C::W_PLANE + j * 2048 + off);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          xh4[i] = *reinterpret_cast<const bf16x8 *>(xb + i * 2048 + off);
          xl4[i] = *reinterpret_cast<const bf16x8 *>(xb + C::X_PLANE + i * 2048 + off);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            acc32[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh4[i], wl2[j], acc32[i][j], 0, 0, 0);
            acc32[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl4[i], wh2[j], acc32[i][j], 0, 0, 0);
            acc32[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh4[i], wh2[j], acc32[i][j], 0, 0, 0);
          }
          if (kh == 0 && i < NXB + NWB) {
            __builtin_amdgcn_sched_barrier(0);
            if (more) issue_part(kt + 1, i);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      continue;
    }
    bf16x8 wh[4], wl[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      wh[j] = *reinterpret_cast<const bf16x8 *>(wbase + bo + j * 1024);
      wl[j] = *reinterpret_cast<const bf16x8 *>(wbase + bo + C::W_PLANE + j * 1024);
    }
#pragma unroll
    for (int i0 = 0; i0 < C::TI; i0 += C::XI) {
      bf16x8 xh[C::XI], xl[C::XI];
#pragma unroll
      for (int i = 0; i < C::XI; ++i) {
        xh[i] = *reinterpret_cast<const bf16x8 *>(xbase + bo + (i0 + i) * 1024);
        xl[i] = *reinterpret_cast<const bf16x8 *>(xbase + bo + C::X_PLANE + (i0 + i) * 1024);
      }
#pragma unroll
      for (int i = 0; i < C::XI; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i0 + i][j] = mfma3(xh[i], xl[i], wh[j], wl[j], acc[i0 + i][j]);
        if (i0 == 0 && i < NXB + NWB) {
          __builtin_amdgcn_sched_barrier(0);
          if (more) issue_part(kt + 1, i);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if (C::XI < NXB + NWB) {      // (wave tiles of fewer than four tile rows: the parts not issued above)
#pragma unroll
      for (int p = C::XI; p < NXB + NWB; ++p)
        if (more) issue_part(kt + 1, p);
    }
  }
  if constexpr (M32) {
    const int n32 = lane & 31, hf = lane >> 5;
    if (cut) {
      const int n_t = nblk / ksplit;
      f32x4 *dst = reinterpret_cast<f32x4 *>(partial) + ((int64_t)piece * n_t + wi / ksplit) * (32 * 512) + tid;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g)
            dst[((i * 2 + j) * 4 + g) * 512] = f32x4{acc32[i][j][4 * g], acc32[i][j][4 * g + 1], acc32[i][j][4 * g + 2], acc32[i][j][4 * g + 3]};
      return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int64_t c = c0 + 128 * wr + 32 * i + 8 * g + 4 * hf;
        if (c >= Nc) continue;
        const int64_t sn = c / h, f = c - sn * h;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int64_t r = r0 + wc * 64 + j * 32 + n32;
          if (r < r_lim)
            *reinterpret_cast<f32x4 *>(out + (sn * R + r) * h + f) =
                f32x4{acc32[i][j][4 * g], acc32[i][j][4 * g + 1], acc32[i][j][4 * g + 2], acc32[i][j][4 * g + 3]};
        }
      }
    return;
  }
  if (cut) {          // piece `piece` of cut tile wi / ksplit: [(piece, tile)][i][j][thread] float4
    const int n_t = nblk / ksplit;
    f32x4 *dst = reinterpret_cast<f32x4 *>(partial) + ((int64_t)piece * n_t + wi / ksplit) * (C::TI * 4 * 512) + tid;
#pragma unroll
    for (int i = 0; i < C::TI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) dst[(i * 4 + j) * 512] = acc[i][j];
    return;
  }
  // D[row = 4 (lane >> 4) + e][col = lane & 15] of tile (i, j): result column c = c0 + 16 (TI wr + i) + 4 (lane >> 4) + e
  // (= snapshot c / h, feature c % h), result row r = r0 + 64 wc + 16 j + (lane & 15): four consecutive features per lane
#pragma unroll
  for (int i = 0; i < C::TI; ++i) {
    const int64_t c = c0 + (C::TI * wr + i) * 16 + (lane >> 4) * 4;
    if (c >= Nc) continue;
    const int64_t s = c / h, f = c - s * h;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t r = r0 + wc * 64 + j * 16 + (lane & 15);
      if (r < r_lim) *reinterpret_cast<f32x4 *>(out + (s * R + r) * h + f) = acc[i][j];
    }
  }
}

// Sum of the ksplit accumulator pieces of the cut tiles (tile t of the n_t of that launch = tile tile_base + t of the matrix), in
// piece order, stored as k_remainder_gemm2's own epilogue would (same thread -> element mapping).
template <int WRN, bool M32 = false>
__global__ __launch_bounds__(512) void k_remainder_gemm2_reduce(const float *__restrict__ partial, int ksplit, int tile_base, int64_t Nc, int64_t R,
                                                                int h, int n_ctile, float *__restrict__ out) {
  using C = Gemm2Cfg<WRN>;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave / C::NWC, wc = wave % C::NWC;
  const int t = blockIdx.x, n_t = gridDim.x, w = tile_base + t;
  const int64_t c0 = (int64_t)(w % n_ctile) * C::BM, r0 = (int64_t)(w / n_ctile) * C::BN;
  const f32x4 *src = reinterpret_cast<const f32x4 *>(partial) + (int64_t)t * (C::TI * 4 * 512) + tid;
  if constexpr (M32) {
    const int n32 = lane & 31, hf = lane >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int64_t c = c0 + 128 * wr + 32 * i + 8 * g + 4 * hf;
        if (c >= Nc) continue;
        const int64_t sn = c / h, f = c - sn * h;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int64_t r = r0 + wc * 64 + j * 32 + n32;
          if (r >= R) continue;
          const int idx = ((i * 2 + j) * 4 + g) * 512;
          f32x4 v = src[idx];
          for (int p = 1; p < ksplit; ++p) {
            const f32x4 u = src[(int64_t)p * n_t * (32 * 512) + idx];
            v[0] += u[0]; v[1] += u[1]; v[2] += u[2]; v[3] += u[3];
          }
          *reinterpret_cast<f32x4 *>(out + (sn * R + r) * h + f) = v;
        }
      }
    return;
  }
#pragma unroll
  for (int i = 0; i < C::TI; ++i) {
    const int64_t c = c0 + (C::TI * wr + i) * 16 + (lane >> 4) * 4;
    if (c >= Nc) continue;
    const int64_t s = c / h, f = c - s * h;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t r = r0 + wc * 64 + j * 16 + (lane & 15);
      if (r >= R) continue;
      f32x4 v = src[(i * 4 + j) * 512];
      for (int p = 1; p < ksplit; ++p) {
        const f32x4 u = src[(int64_t)p * n_t * (C::TI * 4 * 512) + (i * 4 + j) * 512];
        v[0] += u[0]; v[1] += u[1]; v[2] += u[2]; v[3] += u[3];
      }
      *reinterpret_cast<f32x4 *>(out + (s * R + r) * h + f) = v;
    }
  }
}

// x_e = act(e W + b) straight into the GEMM's operand image: the transposed bf16 hi / lo planes of k_split_transpose_bf16, without the
// fp32 tensor in between.  In the trained-bias layer x_e exists only to be multiplied by `rest` (the fused kernel recomputes its own
// secondary MLP), so Dense -> (S, M, h) fp32 -> split / transpose was a write and a read of the tensor for nothing.
// One 256-thread workgroup per (snapshot, 64 rows): wave w multiplies rows 16 w .. + 15 (MFMA 16x16x32, split-bf16, the weight
// fragments of uds_rowgemm_pack read from L2), turns its 16 x h block through LDS and the workgroup writes whole 128-byte runs
// (64 consecutive rows of one feature).  Rows past M are written as zeros (the planes' padding must be zero).
template <int KT>
__global__ __launch_bounds__(256) void k_dense_split_planes(const float *__restrict__ e, int64_t M, int h, int64_t Mp, const uint4 *__restrict__ wq,
                                                            const float *__restrict__ bias, int act, __bf16 *__restrict__ hi, __bf16 *__restrict__ lo) {
  constexpr int F = 32 * KT;
  __shared__ __attribute__((aligned(16))) __bf16 th[64][72], tl[64][72];      // [feature][row], 144-byte rows (16-B aligned, 4-bank skew)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, qd = lane >> 4;
  const int64_t s = blockIdx.y, m0 = (int64_t)blockIdx.x * 64;
  const int64_t m = m0 + 16 * wave + fr;
  const float *row = e + (s * M + min(m, M - 1)) * F;
  const int MB = h / 16;
  f32x4 acc[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) acc[b] = bias && b < MB ? *reinterpret_cast<const f32x4 *>(bias + 16 * b + 4 * qd) : f32x4{0.f, 0.f, 0.f, 0.f};
  const int mbp = h <= 16 ? 1 : (h <= 32 ? 2 : 4);                            // block count of the packed image (rowgemm_mb)
#pragma unroll
  for (int t = 0; t < KT; ++t) {
    const f32x4 u0 = *reinterpret_cast<const f32x4 *>(row + 32 * t + 4 * qd), u1 = *reinterpret_cast<const f32x4 *>(row + 32 * t + 16 + 4 * qd);
    bf16x8 dh, dl;
    split8(make_float4(u0[0], u0[1], u0[2], u0[3]), make_float4(u1[0], u1[1], u1[2], u1[3]), dh, dl);
#pragma unroll
    for (int b = 0; b < 4; ++b)
      if (b < MB) {
        const bf16x8 wh = __builtin_bit_cast(bf16x8, wq[((t * mbp + b) * 2 + 0) * 64 + lane]), wl = __builtin_bit_cast(bf16x8, wq[((t * mbp + b) * 2 + 1) * 64 + lane]);
        acc[b] = mfma3(wh, wl, dh, dl, acc[b]);
      }
  }
  const bool live = m < M;
#pragma unroll
  for (int b = 0; b < 4; ++b)
    if (b < MB) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float v = live ? apply_act(acc[b][j], act) : 0.f;
        const __bf16 vh = (__bf16)v;
        th[16 * b + 4 * qd + j][16 * wave + fr] = vh;
        tl[16 * b + 4 * qd + j][16 * wave + fr] = (__bf16)(v - (float)vh);
      }
    }
  __syncthreads();
  // 8 threads per feature run of 64 rows (128 bytes), 32 features per pass
  for (int f = tid >> 3; f < h; f += 32) {
    const int c = (tid & 7) * 8;
    const int64_t o = (s * h + f) * Mp + m0 + c;
    *reinterpret_cast<uint4 *>(hi + o) = *reinterpret_cast<const uint4 *>(&th[f][c]);
    *reinterpret_cast<uint4 *>(lo + o) = *reinterpret_cast<const uint4 *>(&tl[f][c]);
  }
}

}  // namespace uds
