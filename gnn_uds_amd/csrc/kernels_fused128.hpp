// The COLUMN-SPLIT fused spatial layer.  First written for embed_size d = 128 (h = 64, 128-wide node and link rows: the
// reference's default model size, utils/config.yaml) as ONE launch -- the same tile / snapshot-chunk / LDS-DMA scheme as k_fused_tile (kernels_fused.hpp),
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

#include <algorithm>

#include "kernels_fused.hpp"

namespace uds {

constexpr int F128_H = 64, F128_D = 128, F128_F = 128;
constexpr int F128_U = 2;                       // P3 row groups per wave

// LDS bytes of the column-split kernel for embed size d and the caps of a plan (mirrors the layout in the kernel)
inline int64_t fused_cs_lds_bytes(int d, int p_cap, int q_cap, int meta_cap, int fp, int fs) {
  const int h = d / 2;
  // the score partials (2 x d/16 floats per primary row) live in the sec region: sec is dead between P1.5 and the next P1
  const int64_t sec = std::max<int64_t>((int64_t)q_cap * (h + 4), 2 * (d / 16) * (int64_t)p_cap);
  return 4 * ((int64_t)meta_cap + (2 * d + h) + sec + (int64_t)p_cap * h + (int64_t)p_cap * d + (int64_t)q_cap * fs + (int64_t)p_cap * fp);
}
inline int64_t fused128_lds_bytes(int p_cap, int q_cap, int meta_cap, int fp = F128_F, int fs = F128_F) {
  return fused_cs_lds_bytes(F128_D, p_cap, q_cap, meta_cap, fp, fs);
}

// 16 bytes per lane from base (wave-uniform) + voff (per-lane byte offset) + IMM, invisible to the compiler's vmcnt bookkeeping:
// the value is there after the next wait_all_but() of the snapshot loop (the loads are older than the P3 stores it leaves
// in flight), then f128_pin2 ties the registers to that point
template <int IMM>
__device__ __forceinline__ void f128_gld16(f32x4 &dst, const float *base, unsigned voff) {
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(base), "n"(IMM) : "memory");
}
__device__ __forceinline__ void f128_pin2(f32x4 &a0, f32x4 &a1) { asm volatile("" : "+v"(a0), "+v"(a1)); }

// FP / FS: widths of the primary / secondary input rows, 128 or 64 (the first layer of block 2 without actions has 128-wide
// node rows [temporal output | boundary embedding] and 64-wide link rows, emulator.py:260-262)
// D = embed size (128 with 8 waves; 64 with 16 waves: four waves per SIMD hide the LDS / MFMA latencies the 8-wave
// register-resident kernel k_fused_tile exposes), NW = waves per workgroup.
template <int D, int FP, int FS, int NW, int ACT>
__global__ __launch_bounds__(NW * 64) void k_fused_cs(FusedArgs a) {
  static_assert(D == 64 || D == 128, "embed size 64 or 128");
  static_assert(FP % 32 == 0 && FS % 32 == 0 && FP <= 128 && FS <= 128, "row widths: multiples of 32 up to 128");
  constexpr int H = D / 2, NT = NW * 64, U = F128_U;
  constexpr int KT_S = FS / 32, KT_X = FP / 32, KT_A = H / 32, KT_B = KT_X + KT_A;
  constexpr int MB_S = H / 16, MB_B = D / 16;
  constexpr int MPW = MB_B / 4;                 // hx column blocks per wave (four column groups)
  constexpr int PAR_B = NW / 4;                 // row-block parities of the hx GEMM
  constexpr int PAR_S = NW / MB_S;              // ... of the fusion MLP
  constexpr int NSP = MB_B;                     // attention-score partials per row
  constexpr int CH = D / 64;                    // 16-B chunks per lane and output row in P3
  constexpr int SECS = H + 4;                   // sec row stride (floats): conflict-free 16-B fragment writes
  static_assert(NW % 4 == 0 && NW % MB_S == 0, "waves split into column groups");
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
  float *attn = reinterpret_cast<float *>(smem + a.meta_cap);           // a_self[D] | a_nbr[D] | b_small[H]
  float *sec = attn + 2 * D + H;                                        // [q_cap][H + 4]
  // the per-column-block partials <hx, a_self>, <hx, a_nbr> ([p_cap][NSP] each) are written in P2 and read in P3, when the
  // sec rows are dead (their last reader is P1.5, their next writer the P1 after the next P0 barrier): they share the region
  float *sp_self = sec;
  float *sp_nbr = sec + NSP * a.p_cap;
  const int sec_floats = max(a.q_cap * SECS, 2 * NSP * a.p_cap);
  float *aggf = sec + sec_floats;                                       // (p_cap/16) blocks x KT_A k-steps x (hi 1 KiB | lo 1 KiB)
  float *hx = aggf + a.p_cap * H;                                       // [p_cap][D], 16-B chunks XOR (row & 7)
  float *stage_s = hx + a.p_cap * D;                               // (q_cap/16) blocks x KT_S k-steps x 2 x 1 KiB
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

  // block ownership (DMA issue, P0 split): secondary block b -> wave b % NW, primary block b -> wave NW - 1 - b % NW
  auto sec_owner = [&](int blk) { return blk % NW; };
  auto prim_owner = [&](int blk) { return NW - 1 - blk % NW; };
  auto dma_block = [&](auto KT_, const float *base, int row, float *stage, int blk) {      // 16 rows x 32 KT floats = 2 KT pieces of 1 KiB
    constexpr int KT = decltype(KT_)::value;
    const float *src = base + (int64_t)row * (32 * KT) + 4 * qd;
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_addr(stage) + (unsigned)blk * (2 * KT * 1024));
    if constexpr (KT == 3) {
      const float *p0[6] = {src, src + 16, src + 32, src + 48, src + 64, src + 80};
      glds16_run<6>(p0, dst);
    } else {
      const float *p0[4] = {src, src + 16, src + 32, src + 48};
      glds16_run<4>(p0, dst);
      if constexpr (KT == 4) {
        const float *p1[4] = {src + 64, src + 80, src + 96, src + 112};
        glds16_run<4>(p1, dst + 4096);
      }
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

  // P3 statics of this lane's row groups: degree, the local row of the neighbour this lane scores (slot c16, clamped), the output
  // row.  Per tile, not per snapshot: read inside the snapshot loop they were two dependent LDS round trips (adj_ptr -> adj_loc) in
  // front of the score gather of every snapshot.
  int p3_dmax[U], p3_deg[U], p3_jn[U], p3_orow[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int i = wave * 4 + 4 * NW * u + rs;
    const int ic = min(i, n_own - 1);
    const int b0 = adj_ptr[ic];
    int dmx = i < n_own ? adj_ptr[ic + 1] - b0 : 0;
    p3_deg[u] = dmx;
    p3_jn[u] = adj_loc[b0 + min(lane & 15, max(dmx - 1, 0))];
    p3_orow[u] = prim_ids[ic] * D + 4 * (lane & 15);
    dmx = max(dmx, __shfl_xor(dmx, 16));
    dmx = max(dmx, __shfl_xor(dmx, 32));
    p3_dmax[u] = __builtin_amdgcn_readfirstlane(dmx);
  }

  // The dense remainder of a trained NodeEdge (FusedSide::rem): P1.5's unit `wave` (one per wave: KT_A * nb_prim <= NW, checked by
  // the host) adds its 8 floats per lane to the aggregate.  Loaded one snapshot ahead, right behind the primary rows' DMA.
  const bool has_rem = S_.rem != nullptr && wave < KT_A * nb_prim;      // wave-uniform
  f32x4 rm0 = f32x4{0.f, 0.f, 0.f, 0.f}, rm1 = rm0;
  unsigned rem_off = 0;
  if (has_rem) rem_off = (unsigned)((prim_ids[min((wave / KT_A) * 16 + r16, n_prim - 1)] * H + 32 * (wave % KT_A) + 4 * qd) * 4);
  auto rem_issue = [&](int s) __attribute__((always_inline)) {
    if (has_rem) {
      const float *base = S_.rem + (int64_t)s * S_.n_prim_glob * H;
      f128_gld16<0>(rm0, base, rem_off);
      f128_gld16<64>(rm1, base, rem_off);
    }
  };
  const int s_begin = chunk_id * a.chunk, s_end = min(a.S, (chunk_id + 1) * a.chunk);
  if (s_begin < s_end) {
    dma_sec_all(s_begin);
    dma_prim_all(s_begin);
    rem_issue(s_begin);
  }
  // everything below overlaps with the first snapshot's DMA
  for (int i = tid; i < n_inc; i += NT) reinterpret_cast<float *>(inc_w)[i] = S_.ne_val[inc_w[i]];
  if (tid < D) {
    attn[tid] = S_.a_self[tid];
    attn[D + tid] = S_.a_nbr[tid];
    if (tid < H) attn[2 * D + tid] = S_.b_small ? S_.b_small[tid] : 0.f;
  }
  f32x4 bo[CH];
#pragma unroll
  for (int ch = 0; ch < CH; ++ch) bo[ch] = S_.b_out ? *reinterpret_cast<const f32x4 *>(S_.b_out + 64 * ch + 4 * c16) : f32x4{0.f, 0.f, 0.f, 0.f};
  // this wave's weight columns: fusion MLP slice wave & 3 (16 columns), hx slices 2 (wave & 3) and 2 (wave & 3) + 1 (32
  // columns) -- both for the row blocks of parity wave >> 2: a fragment read from LDS then feeds two column blocks
  const int cs = wave & 3, par = wave >> 2;              // hx GEMM: column group, row-block parity
  const int cs_s = wave % MB_S, par_s = wave / MB_S;     // fusion MLP
  bf16x8 wsh[KT_S], wsl[KT_S], wbh[KT_B][MPW], wbl[KT_B][MPW];
#pragma unroll
  for (int t = 0; t < KT_S; ++t) {
    wsh[t] = __builtin_bit_cast(bf16x8, S_.w_small[((t * MB_S + cs_s) * 2 + 0) * 64 + lane]);
    wsl[t] = __builtin_bit_cast(bf16x8, S_.w_small[((t * MB_S + cs_s) * 2 + 1) * 64 + lane]);
  }
#pragma unroll
  for (int t = 0; t < KT_B; ++t)
#pragma unroll
    for (int m = 0; m < MPW; ++m) {
      wbh[t][m] = __builtin_bit_cast(bf16x8, S_.w_big[((t * MB_B + MPW * cs + m) * 2 + 0) * 64 + lane]);
      wbl[t][m] = __builtin_bit_cast(bf16x8, S_.w_big[((t * MB_B + MPW * cs + m) * 2 + 1) * 64 + lane]);
    }
  __syncthreads();
  // P1.5 operands of this wave's first unit (see there): NodeEdge values are in LDS since the barrier above
  unsigned ag_locs = 0;
  f32x4 ag_vals = f32x4{0.f, 0.f, 0.f, 0.f};
  int ag_beg = 0, ag_end = 0;
  bool ag_wide = false;
  if (wave < KT_A * nb_prim) {
    const int lr = min((wave / KT_A) * 16 + r16, n_prim - 1);
    ag_beg = inc_ptr[lr];
    ag_end = inc_ptr[lr + 1];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (ag_beg + k < ag_end) {
        ag_locs |= (unsigned)inc_loc[ag_beg + k] << (8 * k);      // local secondary rows: q_cap <= 255
        ag_vals[k] = inc_val[ag_beg + k];
      }
    ag_wide = __any(ag_end - ag_beg > 2);
  }
  int n_st = 0;
  UDS_STAMP128(0);

  // the snapshot's output base as a running pointer: formed from the kernel arguments at the stores, the compiler reloads the pointer
  // and the row count inside EACH exec-masked store branch (scalar load + s_waitcnt lgkmcnt(0) + a 64-bit multiply per row group)
  float *out_snap = S_.out + (int64_t)s_begin * S_.n_prim_glob * D;
  const int64_t out_stride = (int64_t)S_.n_prim_glob * D;
  for (int s = s_begin; s < s_end; ++s, out_snap += out_stride) {
    wait_all_but(n_st);       // this wave's DMA pieces of snapshot s have landed (the P3 stores are younger)
    f128_pin2(rm0, rm1);      // ... and its remainder piece
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
    for (int blk = par_s; blk < nb_sec; blk += PAR_S) {
      const float4 *st = reinterpret_cast<const float4 *>(stage_s + blk * (2 * KT_S * 256)) + lane;
      f32x4 acc = *reinterpret_cast<const f32x4 *>(attn + 2 * D + 16 * cs_s + 4 * qd);
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
        *reinterpret_cast<f32x4 *>(sec + lrow * SECS + 16 * cs_s + 4 * qd) = o;
      }
    }
    UDS_STAMP128(4);
    lds_barrier();
    UDS_STAMP128(5);
    if (s + 1 < s_end) dma_sec_all(s + 1);        // the secondary fragments are consumed: fetch the next snapshot's rows
    // (issued from inside P2, behind its first block's MFMAs, the issue time of these DMA instructions -- ~700 cycles per wave and
    // snapshot -- just moves there: the launch takes the same time)
    // ---------------- P1.5: NodeEdge aggregation of the primary rows -> fragments (block wave/2, k-step wave&1) ----------------
    for (int unit = wave; unit < KT_A * nb_prim; unit += NW) {
      const int blk = unit / KT_A, half = unit % KT_A;
      const int lr = min(blk * 16 + r16, n_prim - 1);
      float4 g0 = make_float4(0.f, 0.f, 0.f, 0.f), g1 = g0;
      auto add_row = [&](int loc, float wv) __attribute__((always_inline)) {
        const float *row = sec + loc * SECS + 32 * half + 4 * qd;
        const float4 u0 = *reinterpret_cast<const float4 *>(row);
        const float4 u1 = *reinterpret_cast<const float4 *>(row + 16);
        g0.x = fmaf(wv, u0.x, g0.x); g0.y = fmaf(wv, u0.y, g0.y); g0.z = fmaf(wv, u0.z, g0.z); g0.w = fmaf(wv, u0.w, g0.w);
        g1.x = fmaf(wv, u1.x, g1.x); g1.y = fmaf(wv, u1.y, g1.y); g1.z = fmaf(wv, u1.z, g1.z); g1.w = fmaf(wv, u1.w, g1.w);
      };
      if (unit == wave) {
        // the wave's first unit (its only one when KT_A * nb_prim <= NW): the first four (row, weight) pairs of the lane's row are
        // static per tile and sit in registers -- four independent pairs of LDS reads per snapshot instead of a loop of dependent
        // (weight, index, row) reads per entry, which was 13 % of the kernel for ~1 % of its arithmetic
        add_row((int)(ag_locs & 0xffu), ag_vals[0]);
        add_row((int)((ag_locs >> 8) & 0xffu), ag_vals[1]);
        if (ag_wide) {      // wave-uniform: some row of the unit has more than two entries
          add_row((int)((ag_locs >> 16) & 0xffu), ag_vals[2]);
          add_row((int)(ag_locs >> 24), ag_vals[3]);
        }
        for (int p = ag_beg + 4; p < ag_end; ++p) add_row(inc_loc[p], inc_val[p]);
      } else {
        for (int p = inc_ptr[lr]; p < inc_ptr[lr + 1]; ++p) add_row(inc_loc[p], inc_val[p]);
      }
      if (has_rem) {          // unit == wave
        g0.x += rm0[0]; g0.y += rm0[1]; g0.z += rm0[2]; g0.w += rm0[3];
        g1.x += rm1[0]; g1.y += rm1[1]; g1.z += rm1[2]; g1.w += rm1[3];
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
      f32x4 as4[MPW], an4[MPW];
#pragma unroll
      for (int m = 0; m < MPW; ++m) {
        as4[m] = *reinterpret_cast<const f32x4 *>(attn + 16 * (MPW * cs + m) + 4 * qd);
        an4[m] = *reinterpret_cast<const f32x4 *>(attn + D + 16 * (MPW * cs + m) + 4 * qd);
      }
      for (int blk = par; blk < nb_prim; blk += PAR_B) {
        const float4 *st = reinterpret_cast<const float4 *>(stage_p + blk * (2 * KT_X * 256)) + lane;
        const float4 *ag = reinterpret_cast<const float4 *>(aggf + blk * (KT_A * 512)) + lane;
        f32x4 acc[MPW];
#pragma unroll
        for (int m = 0; m < MPW; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < KT_X; ++t) {
#ifndef UDS_F128_NOP0
          const bf16x8 dh = __builtin_bit_cast(bf16x8, st[(2 * t) * 64]), dl = __builtin_bit_cast(bf16x8, st[(2 * t + 1) * 64]);
#else
          bf16x8 dh, dl;
          split8(st[(2 * t) * 64], st[(2 * t + 1) * 64], dh, dl);
#endif
#pragma unroll
          for (int m = 0; m < MPW; ++m) acc[m] = mfma3(wbh[t][m], wbl[t][m], dh, dl, acc[m]);
        }
#pragma unroll
        for (int t = 0; t < KT_A; ++t) {
          const bf16x8 dh = __builtin_bit_cast(bf16x8, ag[t * 128]), dl = __builtin_bit_cast(bf16x8, ag[t * 128 + 64]);
#pragma unroll
          for (int m = 0; m < MPW; ++m) acc[m] = mfma3(wbh[KT_X + t][m], wbl[KT_X + t][m], dh, dl, acc[m]);
        }
        const int lrow = blk * 16 + r16;
#pragma unroll
        for (int m = 0; m < MPW; ++m) {
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
              sp_self[lrow * NSP + MPW * cs + m] = ps;
              sp_nbr[lrow * NSP + MPW * cs + m] = pn;
            }
            *reinterpret_cast<f32x4 *>(hx + lrow * D + (((4 * (MPW * cs + m) + qd) ^ (lrow & 7)) << 2)) = acc[m];
          }
        }
      }
    }
    UDS_STAMP128(8);
    lds_barrier();
    UDS_STAMP128(9);
    if (s + 1 < s_end) {      // the primary fragments are consumed
      dma_prim_all(s + 1);
      rem_issue(s + 1);
    }
    // ---------------- P3: segmented softmax + neighbour sum -> HBM (16 lanes x CH float4 per output row) ----------------
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
        deg[u] = p3_deg[u];
        jn[u] = p3_jn[u];          // slot c16 of the row, clamped: slots past the degree get weight 0
        orow[u] = p3_orow[u];
        // the NSP per-column-block partials of a score, added in a fixed order: lanes 0..NSP-1 of the row group hold one each
        ss[u] = row16_sum(c16 < NSP ? sp_self[ic * NSP + c16] : 0.f);
        const f32x4 q0 = *reinterpret_cast<const f32x4 *>(sp_nbr + jn[u] * NSP);
        sn[u] = (q0[0] + q0[1]) + (q0[2] + q0[3]);
        if constexpr (NSP == 8) {
          const f32x4 q1 = *reinterpret_cast<const f32x4 *>(sp_nbr + jn[u] * NSP + 4);
          sn[u] += (q1[0] + q1[1]) + (q1[2] + q1[3]);
        }
        dm = max(dm, p3_dmax[u]);
      }
      f32x4 acc[U][CH];
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int ch = 0; ch < CH; ++ch) acc[u][ch] = f32x4{0.f, 0.f, 0.f, 0.f};
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
          joff[u] = jn[u] * (D * 4) + ((jn[u] & 7) << 4);       // row byte offset with the swizzle key in bits 4-6
        }
        const int cx = c16 << 4;
        auto step = [&](auto K_, auto A_) {
          constexpr int K = decltype(K_)::value, A = decltype(A_)::value;
          f32x4 h0[A][CH], h1[A][CH];
#pragma unroll
          for (int u = 0; u < A; ++u) {
            const int a0 = row16_bcast<K>(joff[u]) ^ cx, a1 = row16_bcast<K + 1>(joff[u]) ^ cx;
#pragma unroll
            for (int ch = 0; ch < CH; ++ch) {
              h0[u][ch] = *reinterpret_cast<const f32x4 *>(hxb + a0 + 256 * ch);
              h1[u][ch] = *reinterpret_cast<const f32x4 *>(hxb + a1 + 256 * ch);
            }
          }
#pragma unroll
          for (int u = 0; u < A; ++u) {
            const float w0 = __int_as_float(row16_bcast<K>(__float_as_int(wgt[u])));
            const float w1 = __int_as_float(row16_bcast<K + 1>(__float_as_int(wgt[u])));
#pragma unroll
            for (int ch = 0; ch < CH; ++ch)
#pragma unroll
              for (int q = 0; q < 4; ++q) acc[u][ch][q] = fmaf(w1, h1[u][ch][q], fmaf(w0, h0[u][ch][q], acc[u][ch][q]));
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
            for (int k = 0; k < NSP; ++k) t += sp_nbr[j * NSP + k];
            return t;
          };
          float mx = -INFINITY;
          for (int p = b0; p < b0 + deg[u]; ++p) mx = fmaxf(mx, leaky02(ss[u] + s_nbr_of(adj_loc[p])));
          den[u] = 0.f;
          for (int p = b0; p < b0 + deg[u]; ++p) {
            const int jj = adj_loc[p];
            const float wv = __builtin_amdgcn_exp2f((leaky02(ss[u] + s_nbr_of(jj)) - mx) * 1.44269504088896340736f);
            den[u] += wv;
#pragma unroll
            for (int ch = 0; ch < CH; ++ch) {
              const f32x4 hv = *reinterpret_cast<const f32x4 *>(hx + jj * D + (((16 * ch + c16) ^ (jj & 7)) << 2));
#pragma unroll
              for (int q = 0; q < 4; ++q) acc[u][ch][q] = fmaf(wv, hv[q], acc[u][ch][q]);
            }
          }
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (p3_dmax[u] > 0) n_st += CH;       // wave-uniform: CH store instructions per row group that has a valid row
        if (ok[u]) {
          const float inv = __builtin_amdgcn_rcpf(den[u]);
          float *dst = out_snap + orow[u];
#pragma unroll
          for (int ch = 0; ch < CH; ++ch) {
            f32x4 o;
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = fused_act<ACT>(fmaf(acc[u][ch][q], inv, bo[ch][q]), a.act);
            *reinterpret_cast<f32x4 *>(dst + 64 * ch) = o;
          }
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
