// One spatial layer at embed_size d = 128 (h = 64, 128-wide node and link rows: the reference's default model size,
// utils/config.yaml) as ONE launch -- the same tile / snapshot-chunk / LDS-DMA scheme as k_fused_tile (kernels_fused.hpp),
// re-organised because at d = 128 the weight fragments (128 + 384 VGPRs per lane) fit no wave:
//
//   * the GEMMs are split by OUTPUT COLUMNS (and row-block parity) across the 8 waves: a wave owns 32 of the 128 columns
//     of hx (6 k-steps x 2 x hi/lo = 96 VGPRs of weights) and, in the fusion MLP, 16 of its 64 columns (4 k-steps = 32
//     VGPRs), each for every second row block: weights stay register-resident, read once per workgroup;
//   * every wave therefore needs EVERY row block's operand fragments: the wave that DMA'd a block splits it once into
//     bf16 hi / lo fragments IN PLACE in the stage (phase P0), all waves read the shared fragments;
//   * phases per snapshot:  P0 split | P1 fusion MLP (columns x row-block parity) | P1.5 NodeEdge aggregation -> fragments |
//     P2 hx columns + partial attention scores per wave | P3 softmax + neighbour sum (16 lanes x 2 float4 per row).
//     The secondary stage is refilled after P1, the primary stage after P2 (next snapshot's rows, LDS-DMA).
//   * the attention scores are sums over the 8 column slices: each wave stores its partial, P3 adds the 8 partials in a
//     fixed order (bitwise reproducible, no LDS atomics).
//
// Tiles: <= 64 own rows (P3: 8 waves x 2 groups x 4 rows), <= 64 primary, <= 80 secondary rows (LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels_fused.hpp"

namespace uds {

constexpr int F128_H = 64, F128_D = 128, F128_F = 128;
constexpr int F128_U = 2;                       // P3 row groups per wave
constexpr int F128_SEC_STRIDE = F128_H + 4;     // floats: 272-B rows, conflict-free 16-B fragment writes

// LDS bytes of k_fused128 for the caps of a plan (mirrors the layout in the kernel)
inline int64_t fused128_lds_bytes(int p_cap, int q_cap, int meta_cap, int fp = F128_F, int fs = F128_F) {
  return 4 * ((int64_t)meta_cap + 2 * 8 * p_cap + (2 * F128_D + F128_H) + (int64_t)q_cap * F128_SEC_STRIDE + (int64_t)p_cap * F128_H +
              (int64_t)p_cap * F128_D + (int64_t)q_cap * fs + (int64_t)p_cap * fp);
}

// FP / FS: widths of the primary / secondary input rows, 128 or 64 (the first layer of block 2 without actions has 128-wide
// node rows [temporal output | boundary embedding] and 64-wide link rows, emulator.py:260-262)
template <int FP, int FS, int ACT>
__global__ __launch_bounds__(FUSED_WAVES * 64, 2) void k_fused128(FusedArgs a) {
  static_assert((FP == 64 || FP == 128) && (FS == 64 || FS == 128), "row widths 64 or 128");
  constexpr int NW = FUSED_WAVES, NT = FUSED_WAVES * 64, U = F128_U;
  constexpr int KT_S = FS / 32, KT_X = FP / 32, KT_A = F128_H / 32, KT_B = KT_X + KT_A;     // 4|2, 4|2, 2, 6|4
  constexpr int MB_S = F128_H / 16, MB_B = F128_D / 16;                                              // 4, 8
  extern __shared__ __attribute__((aligned(16))) int32_t smem[];
#ifdef UDS_PHASE_TIMING
  unsigned long long tq_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tp_ = clock64();
#define UDS_STAMP128(k) do { const unsigned long long n_ = clock64(); tq_[k] += n_ - tp_; tp_ = n_; } while (0)
#else
#define UDS_STAMP128(k) do { } while (0)
#endif
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, qd = lane >> 4;

  const int W = gridDim.x, b = blockIdx.x;
  const int q8 = W / 8, r8 = W % 8, xcd = b % 8;
  const int w = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + b / 8;
  const int tile = w % a.n_tiles, chunk_id = w / a.n_tiles;
  const int32_t *hd = a.hdr + tile * TILE_HDR_INTS;
  const int n_own = hd[0], n_prim = hd[1], n_sec = hd[2], n_inc = hd[3], pool_off = hd[5], sd = hd[6], meta_len = hd[7];
  if (!((a.side_mask >> sd) & 1)) return;
  const FusedSide &S_ = a.side[sd];

  int32_t *meta = smem;
  float *sp_self = reinterpret_cast<float *>(smem + a.meta_cap);       // [p_cap][8]: per-wave partial <hx, a_self>
  float *sp_nbr = sp_self + 8 * a.p_cap;                                // [p_cap][8]
  float *attn = sp_nbr + 8 * a.p_cap;                                   // a_self[128] | a_nbr[128] | b_small[64]
  float *sec = attn + 2 * F128_D + F128_H;                              // [q_cap][68]
  float *aggf = sec + a.q_cap * F128_SEC_STRIDE;                        // (p_cap/16) blocks x 2 k-steps x (hi 1 KiB | lo 1 KiB)
  float *hx = aggf + a.p_cap * F128_H;                                  // [p_cap][128], 16-B chunks XOR (row & 7)
  float *stage_s = hx + a.p_cap * F128_D;                               // (q_cap/16) blocks x KT_S k-steps x 2 x 1 KiB
  float *stage_p = stage_s + a.q_cap * FS;                              // (p_cap/16) blocks x KT_X k-steps x 2 x 1 KiB

  for (int i = tid; i < meta_len; i += NT) meta[i] = a.pool[pool_off + i];
  __syncthreads();
  const int32_t *prim_ids = meta;
  const int32_t *sec_ids = prim_ids + n_prim;
  const int32_t *inc_ptr = sec_ids + n_sec;
  const int32_t *inc_loc = inc_ptr + n_prim + 1;
  int32_t *inc_w = const_cast<int32_t *>(inc_loc) + n_inc;
  const int32_t *adj_ptr = inc_w + n_inc;
  const int32_t *adj_loc = adj_ptr + n_own + 1;
  const float *inc_val = reinterpret_cast<const float *>(inc_w);
  const int c16 = lane & 15, rs = lane >> 4;
  const int nb_sec = (n_sec + 15) / 16, nb_prim = (n_prim + 15) / 16;

  // block ownership (DMA issue, P0 split): secondary block b -> wave b % 8, primary block b -> wave 7 - b % 8
  auto sec_owner = [&](int blk) { return blk & 7; };
  auto prim_owner = [&](int blk) { return 7 - (blk & 7); };
  auto dma_block = [&](auto KT_, const float *base, int row, float *stage, int blk) {      // 16 rows x 32 KT floats = 2 KT pieces of 1 KiB
    constexpr int KT = decltype(KT_)::value;
    const float *src = base + (int64_t)row * (32 * KT) + 4 * qd;
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_addr(stage) + (unsigned)blk * (2 * KT * 1024));
    const float *p0[4] = {src, src + 16, src + 32, src + 48};
    glds16_run<4>(p0, dst);
    if constexpr (KT == 4) {
      const float *p1[4] = {src + 64, src + 80, src + 96, src + 112};
      glds16_run<4>(p1, dst + 4096);
    }
  };
  auto dma_sec_all = [&](int s) {
    for (int blk = 0; blk < nb_sec; ++blk)
      if (sec_owner(blk) == wave)
        dma_block(std::integral_constant<int, KT_S>{}, S_.sec_in + (int64_t)s * S_.n_sec_glob * FS, sec_ids[min(blk * 16 + r16, n_sec - 1)], stage_s, blk);
  };
  auto dma_prim_all = [&](int s) {
    for (int blk = 0; blk < nb_prim; ++blk)
      if (prim_owner(blk) == wave)
        dma_block(std::integral_constant<int, KT_X>{}, S_.prim_in + (int64_t)s * S_.n_prim_glob * FP, prim_ids[min(blk * 16 + r16, n_prim - 1)], stage_p, blk);
  };

  int p3_dmax[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int i = wave * 4 + 4 * NW * u + rs;
    const int ic = min(i, n_own - 1);
    int dmx = i < n_own ? adj_ptr[ic + 1] - adj_ptr[ic] : 0;
    dmx = max(dmx, __shfl_xor(dmx, 16));
    dmx = max(dmx, __shfl_xor(dmx, 32));
    p3_dmax[u] = __builtin_amdgcn_readfirstlane(dmx);
  }

  const int s_begin = chunk_id * a.chunk, s_end = min(a.S, (chunk_id + 1) * a.chunk);
  if (s_begin < s_end) {
    dma_sec_all(s_begin);
    dma_prim_all(s_begin);
  }
  // everything below overlaps with the first snapshot's DMA
  for (int i = tid; i < n_inc; i += NT) reinterpret_cast<float *>(inc_w)[i] = S_.ne_val[inc_w[i]];
  if (tid < F128_D) {
    attn[tid] = S_.a_self[tid];
    attn[F128_D + tid] = S_.a_nbr[tid];
    if (tid < F128_H) attn[2 * F128_D + tid] = S_.b_small ? S_.b_small[tid] : 0.f;
  }
  f32x4 bo[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  if (S_.b_out) {
    bo[0] = *reinterpret_cast<const f32x4 *>(S_.b_out + 4 * c16);
    bo[1] = *reinterpret_cast<const f32x4 *>(S_.b_out + 64 + 4 * c16);
  }
  // this wave's weight columns: fusion MLP slice wave & 3 (16 columns), hx slices 2 (wave & 3) and 2 (wave & 3) + 1 (32
  // columns) -- both for the row blocks of parity wave >> 2: a fragment read from LDS then feeds two column blocks
  const int cs = wave & 3, par = wave >> 2;
  bf16x8 wsh[KT_S], wsl[KT_S], wbh[KT_B][2], wbl[KT_B][2];
#pragma unroll
  for (int t = 0; t < KT_S; ++t) {
    wsh[t] = __builtin_bit_cast(bf16x8, S_.w_small[((t * MB_S + cs) * 2 + 0) * 64 + lane]);
    wsl[t] = __builtin_bit_cast(bf16x8, S_.w_small[((t * MB_S + cs) * 2 + 1) * 64 + lane]);
  }
#pragma unroll
  for (int t = 0; t < KT_B; ++t)
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      wbh[t][m] = __builtin_bit_cast(bf16x8, S_.w_big[((t * MB_B + 2 * cs + m) * 2 + 0) * 64 + lane]);
      wbl[t][m] = __builtin_bit_cast(bf16x8, S_.w_big[((t * MB_B + 2 * cs + m) * 2 + 1) * 64 + lane]);
    }
  __syncthreads();
  int n_st = 0;
  UDS_STAMP128(0);

  for (int s = s_begin; s < s_end; ++s) {
    wait_all_but(n_st);       // this wave's DMA pieces of snapshot s have landed (the P3 stores are younger)
    UDS_STAMP128(1);
    // ---------------- P0: raw fp32 rows -> bf16 hi / lo fragments, in place, by the wave that fetched them ----------------
    auto split_block = [&](auto KT_, float *stage, int blk) {
      constexpr int KT = decltype(KT_)::value;
      float4 *st = reinterpret_cast<float4 *>(stage + blk * (2 * KT * 256)) + lane;
      float4 v[2 * KT];
#pragma unroll
      for (int i = 0; i < 2 * KT; ++i) v[i] = st[i * 64];
#pragma unroll
      for (int t = 0; t < KT; ++t) {
        bf16x8 hi, lo;
        split8(v[2 * t], v[2 * t + 1], hi, lo);
        st[(2 * t) * 64] = __builtin_bit_cast(float4, hi);
        st[(2 * t + 1) * 64] = __builtin_bit_cast(float4, lo);
      }
    };
#ifndef UDS_F128_NOP0
    for (int blk = 0; blk < nb_sec; ++blk)
      if (sec_owner(blk) == wave) split_block(std::integral_constant<int, KT_S>{}, stage_s, blk);
    for (int blk = 0; blk < nb_prim; ++blk)
      if (prim_owner(blk) == wave) split_block(std::integral_constant<int, KT_X>{}, stage_p, blk);
#else
    (void)split_block;
#endif
    UDS_STAMP128(2);
    lds_barrier();
    UDS_STAMP128(3);
    // ---------------- P1: fusion MLP, 16 columns (cs) x the row blocks of this wave's parity -> sec ----------------
    for (int blk = par; blk < nb_sec; blk += 2) {
      const float4 *st = reinterpret_cast<const float4 *>(stage_s + blk * (2 * KT_S * 256)) + lane;
      f32x4 acc = *reinterpret_cast<const f32x4 *>(attn + 2 * F128_D + 16 * cs + 4 * qd);
#pragma unroll
      for (int t = 0; t < KT_S; ++t) {
#ifndef UDS_F128_NOP0
        acc = mfma3(wsh[t], wsl[t], __builtin_bit_cast(bf16x8, st[(2 * t) * 64]), __builtin_bit_cast(bf16x8, st[(2 * t + 1) * 64]), acc);
#else
        bf16x8 dh, dl;
        split8(st[(2 * t) * 64], st[(2 * t + 1) * 64], dh, dl);
        acc = mfma3(wsh[t], wsl[t], dh, dl, acc);
#endif
      }
      const int lrow = blk * 16 + r16;
      if (lrow < n_sec) {
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = fused_act<ACT>(acc[j], a.act);
        *reinterpret_cast<f32x4 *>(sec + lrow * F128_SEC_STRIDE + 16 * cs + 4 * qd) = o;
      }
    }
    UDS_STAMP128(4);
    lds_barrier();
    UDS_STAMP128(5);
    if (s + 1 < s_end) dma_sec_all(s + 1);        // the secondary fragments are consumed: fetch the next snapshot's rows
    // ---------------- P1.5: NodeEdge aggregation of the primary rows -> fragments (block wave/2, k-step wave&1) ----------------
    for (int unit = wave; unit < 2 * nb_prim; unit += NW) {
      const int blk = unit >> 1, half = unit & 1;
      const int lr = min(blk * 16 + r16, n_prim - 1);
      float4 g0 = make_float4(0.f, 0.f, 0.f, 0.f), g1 = g0;
      for (int p = inc_ptr[lr]; p < inc_ptr[lr + 1]; ++p) {
        const float wv = inc_val[p];
        const float *row = sec + inc_loc[p] * F128_SEC_STRIDE + 32 * half + 4 * qd;
        const float4 u0 = *reinterpret_cast<const float4 *>(row);
        const float4 u1 = *reinterpret_cast<const float4 *>(row + 16);
        g0.x = fmaf(wv, u0.x, g0.x); g0.y = fmaf(wv, u0.y, g0.y); g0.z = fmaf(wv, u0.z, g0.z); g0.w = fmaf(wv, u0.w, g0.w);
        g1.x = fmaf(wv, u1.x, g1.x); g1.y = fmaf(wv, u1.y, g1.y); g1.z = fmaf(wv, u1.z, g1.z); g1.w = fmaf(wv, u1.w, g1.w);
      }
      bf16x8 hi, lo;
      split8(g0, g1, hi, lo);
      float4 *dst = reinterpret_cast<float4 *>(aggf + unit * 512) + lane;
      dst[0] = __builtin_bit_cast(float4, hi);
      dst[64] = __builtin_bit_cast(float4, lo);
    }
    UDS_STAMP128(6);
    lds_barrier();
    UDS_STAMP128(7);
    // ---------------- P2: hx columns [32 cs, 32 cs + 32) of the primary blocks of this wave's parity + partial scores ----------------
    {
      f32x4 as4[2], an4[2];
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        as4[m] = *reinterpret_cast<const f32x4 *>(attn + 16 * (2 * cs + m) + 4 * qd);
        an4[m] = *reinterpret_cast<const f32x4 *>(attn + F128_D + 16 * (2 * cs + m) + 4 * qd);
      }
      for (int blk = par; blk < nb_prim; blk += 2) {
        const float4 *st = reinterpret_cast<const float4 *>(stage_p + blk * (2 * KT_X * 256)) + lane;
        const float4 *ag = reinterpret_cast<const float4 *>(aggf + blk * 1024) + lane;
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int t = 0; t < KT_X; ++t) {
#ifndef UDS_F128_NOP0
          const bf16x8 dh = __builtin_bit_cast(bf16x8, st[(2 * t) * 64]), dl = __builtin_bit_cast(bf16x8, st[(2 * t + 1) * 64]);
#else
          bf16x8 dh, dl;
          split8(st[(2 * t) * 64], st[(2 * t + 1) * 64], dh, dl);
#endif
#pragma unroll
          for (int m = 0; m < 2; ++m) acc[m] = mfma3(wbh[t][m], wbl[t][m], dh, dl, acc[m]);
        }
#pragma unroll
        for (int t = 0; t < KT_A; ++t) {
          const bf16x8 dh = __builtin_bit_cast(bf16x8, ag[t * 128]), dl = __builtin_bit_cast(bf16x8, ag[t * 128 + 64]);
#pragma unroll
          for (int m = 0; m < 2; ++m) acc[m] = mfma3(wbh[KT_X + t][m], wbl[KT_X + t][m], dh, dl, acc[m]);
        }
        const int lrow = blk * 16 + r16;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          float ps = 0.f, pn = 0.f;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            ps = fmaf(acc[m][j], as4[m][j], ps);
            pn = fmaf(acc[m][j], an4[m][j], pn);
          }
          ps = quarters_sum(ps);
          pn = quarters_sum(pn);
          if (lrow < n_prim) {
            if (qd == 0) {
              sp_self[lrow * 8 + 2 * cs + m] = ps;
              sp_nbr[lrow * 8 + 2 * cs + m] = pn;
            }
            *reinterpret_cast<f32x4 *>(hx + lrow * F128_D + (((4 * (2 * cs + m) + qd) ^ (lrow & 7)) << 2)) = acc[m];
          }
        }
      }
    }
    UDS_STAMP128(8);
    lds_barrier();
    UDS_STAMP128(9);
    if (s + 1 < s_end) dma_prim_all(s + 1);       // the primary fragments are consumed
    // ---------------- P3: segmented softmax + neighbour sum -> HBM (16 lanes x 2 float4 per output row) ----------------
    n_st = 0;
    {
      int deg[U], jn[U], orow[U];
      float ss[U], sn[U], wgt[U], den[U];
      bool ok[U];
      int dm = 0;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = wave * 4 + 4 * NW * u + rs;
        const int ic = min(i, n_own - 1);
        ok[u] = i < n_own;
        const int b0 = adj_ptr[ic];
        deg[u] = ok[u] ? adj_ptr[ic + 1] - b0 : 0;
        jn[u] = c16 < deg[u] ? adj_loc[b0 + c16] : 0;
        orow[u] = prim_ids[ic] * F128_D + 4 * c16;
        // the 8 per-wave partials of a score, added in index order: lanes 0..7 of the row group hold one partial each
        ss[u] = row16_sum(c16 < 8 ? sp_self[ic * 8 + c16] : 0.f);
        const f32x4 q0 = *reinterpret_cast<const f32x4 *>(sp_nbr + jn[u] * 8), q1 = *reinterpret_cast<const f32x4 *>(sp_nbr + jn[u] * 8 + 4);
        sn[u] = ((q0[0] + q0[1]) + (q0[2] + q0[3])) + ((q1[0] + q1[1]) + (q1[2] + q1[3]));
        dm = max(dm, p3_dmax[u]);
      }
      f32x4 acc[U][2];
#pragma unroll
      for (int u = 0; u < U; ++u) acc[u][0] = acc[u][1] = f32x4{0.f, 0.f, 0.f, 0.f};
      const char *hxb = reinterpret_cast<const char *>(hx);
      if (dm <= 16) {
        int joff[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const float lg = c16 < deg[u] ? leaky02(ss[u] + sn[u]) : -INFINITY;
          const float mx = row16_max(lg);
          const float ex = __builtin_amdgcn_exp2f((lg - mx) * 1.44269504088896340736f);
          wgt[u] = c16 < deg[u] ? ex : 0.f;
          den[u] = row16_sum(wgt[u]);
          joff[u] = jn[u] * (F128_D * 4) + ((jn[u] & 7) << 4);       // row byte offset with the swizzle key in bits 4-6
        }
        const int cx = c16 << 4;
        auto step = [&](auto K_, auto A_) {
          constexpr int K = decltype(K_)::value, A = decltype(A_)::value;
          f32x4 h0[A][2], h1[A][2];
#pragma unroll
          for (int u = 0; u < A; ++u) {
            const int a0 = row16_bcast<K>(joff[u]) ^ cx, a1 = row16_bcast<K + 1>(joff[u]) ^ cx;
            h0[u][0] = *reinterpret_cast<const f32x4 *>(hxb + a0);
            h0[u][1] = *reinterpret_cast<const f32x4 *>(hxb + a0 + 256);
            h1[u][0] = *reinterpret_cast<const f32x4 *>(hxb + a1);
            h1[u][1] = *reinterpret_cast<const f32x4 *>(hxb + a1 + 256);
          }
#pragma unroll
          for (int u = 0; u < A; ++u) {
            const float w0 = __int_as_float(row16_bcast<K>(__float_as_int(wgt[u])));
            const float w1 = __int_as_float(row16_bcast<K + 1>(__float_as_int(wgt[u])));
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              acc[u][0][q] = fmaf(w1, h1[u][0][q], fmaf(w0, h0[u][0][q], acc[u][0][q]));
              acc[u][1][q] = fmaf(w1, h1[u][1][q], fmaf(w0, h0[u][1][q], acc[u][1][q]));
            }
          }
        };
        const int e1 = p3_dmax[1], e0 = max(e1, p3_dmax[0]);
        bool more = e0 > 0;
        static_for<8>([&](auto t_) {
          constexpr int K = decltype(t_)::value * 2;
          if (more) {
            if (K < e1) step(std::integral_constant<int, K>{}, std::integral_constant<int, 2>{});
            else step(std::integral_constant<int, K>{}, std::integral_constant<int, 1>{});
            more = e0 > K + 2;
          }
        });
      } else {   // some row has more than 16 neighbours: every lane walks its row's whole list
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int i = wave * 4 + 4 * NW * u + rs;
          const int b0 = i < n_own ? adj_ptr[i] : 0;
          auto s_nbr_of = [&](int j) {
            float t = 0.f;
            for (int k = 0; k < 8; ++k) t += sp_nbr[j * 8 + k];
            return t;
          };
          float mx = -INFINITY;
          for (int p = b0; p < b0 + deg[u]; ++p) mx = fmaxf(mx, leaky02(ss[u] + s_nbr_of(adj_loc[p])));
          den[u] = 0.f;
          for (int p = b0; p < b0 + deg[u]; ++p) {
            const int jj = adj_loc[p];
            const float wv = __builtin_amdgcn_exp2f((leaky02(ss[u] + s_nbr_of(jj)) - mx) * 1.44269504088896340736f);
            const f32x4 hv0 = *reinterpret_cast<const f32x4 *>(hx + jj * F128_D + ((c16 ^ (jj & 7)) << 2));
            const f32x4 hv1 = *reinterpret_cast<const f32x4 *>(hx + jj * F128_D + (((16 + c16) ^ (jj & 7)) << 2));
            den[u] += wv;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              acc[u][0][q] = fmaf(wv, hv0[q], acc[u][0][q]);
              acc[u][1][q] = fmaf(wv, hv1[q], acc[u][1][q]);
            }
          }
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (p3_dmax[u] > 0) n_st += 2;       // wave-uniform: two store instructions per row group that has a valid row
        if (ok[u]) {
          const float inv = __builtin_amdgcn_rcpf(den[u]);
          f32x4 o0, o1;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            o0[q] = fused_act<ACT>(fmaf(acc[u][0][q], inv, bo[0][q]), a.act);
            o1[q] = fused_act<ACT>(fmaf(acc[u][1][q], inv, bo[1][q]), a.act);
          }
          float *dst = S_.out + ((int64_t)s * S_.n_prim_glob * F128_D + orow[u]);
          *reinterpret_cast<f32x4 *>(dst) = o0;
          *reinterpret_cast<f32x4 *>(dst + 64) = o1;
        }
      }
    }
    UDS_STAMP128(10);
    // no barrier here: the next P0 touches only this wave's own stage blocks, whose readers all passed the barriers above
  }
#ifdef UDS_PHASE_TIMING
  if (a.dbg && lane == 0) {
    unsigned long long *o = a.dbg + ((size_t)blockIdx.x * NW + wave) * 16;
    for (int k = 0; k < 11; ++k) o[k] = tq_[k];
    o[12] = wave;
    o[13] = 1ull | ((unsigned long long)sd << 8) | ((unsigned long long)n_own << 16) | ((unsigned long long)n_prim << 32) | ((unsigned long long)n_sec << 48);
  }
#endif
}

}  // namespace uds
