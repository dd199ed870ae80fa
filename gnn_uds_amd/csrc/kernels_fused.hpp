// Fused spatial-layer kernel (gfx950): one workgroup = one tile of one side of the network,
// looping over a chunk of snapshots.  Replaces, per side, the whole chain
//   Dense(d/2,relu) on the secondary rows -> NodeEdge aggregation -> concat -> GAT linear ->
//   attention logits / segmented softmax / neighbour sum -> bias -> activation
// (emulator.py:225-230) with ONE pass over HBM: inputs are read, the only global writes are the
// layer outputs; x_e / e_x / agg / hx / attention scalars live in LDS or registers.
//
// Per snapshot, per tile:
//   P1  secondary MLP   sec[q]  = relu(in_sec[q] @ Wsmall + b)        rows gathered from HBM -> MFMA -> LDS
//   P2  primary linear  hx[p]   = [in_prim[p] | sum_q w_pq sec[q]] @ Wbig ; s_self, s_nbr -> LDS
//   P3  GAT aggregate   out[i]  = act(sum_j softmax_j(leaky(s_self_i+s_nbr_j)) hx[j] + bias)  LDS -> HBM
//
// GEMMs: v_mfma_f32_16x16x32_bf16 with the WEIGHTS as the A operand and 16 data rows as the B operand
// (result lane layout = 4 consecutive features of one row -> 16-B LDS / HBM accesses).  fp32 operands
// are split into bf16 hi + lo and three products are accumulated in fp32 (hi*hi + lo*hi + hi*lo,
// error ~2^-16 relative per product: MORE mantissa than the TF32 path TensorFlow uses by default on
// the reference's GPUs, cheaper than the 157 TF fp32 MFMA that would cap the layer below the HBM
// roofline -- SURVEY.md section 7).  Weight fragments (32 + 96 VGPRs per lane at F=64) are loaded once
// per workgroup and stay in registers across the snapshot loop; data fragments are loaded straight
// from HBM in fragment shape (each row read once, 64 B per lane per k-step pair), never via LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels_dense.hpp"
#include "kernels_sparse.hpp"
#include "tile_plan.hpp"

namespace uds {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int FUSED_H = 32;             // d/2
constexpr int FUSED_D = 64;             // d
constexpr int SEC_STRIDE = FUSED_H + 4; // floats; 144-B rows make the 16-B fragment writes conflict-free

struct FusedSide {
  const float *prim_in, *sec_in;
  float *out;
  const uint4 *w_small, *w_big;   // packed bf16 hi/lo fragments (k_pack_weight_frags)
  const float *b_small, *a_self, *a_nbr, *b_out, *ne_val;
  int n_prim_glob, n_sec_glob;
};

struct FusedArgs {
  FusedSide side[2];
  const int32_t *hdr, *pool;
  int n_tiles, S, chunk, p_cap, q_cap, meta_cap, act, side_mask;
};

// k index a lane's element jj (0..7) of k-step t stands for: two 16-B pieces per lane so that one
// load instruction covers 64 contiguous bytes of each of the 16 rows.
__device__ __forceinline__ int frag_k(int t, int qd, int jj) { return 32 * t + (jj < 4 ? 4 * qd + jj : 16 + 4 * qd + (jj - 4)); }

// Pack a row-major fp32 weight matrix W (K x F_out, K % 32 == 0, F_out % 16 == 0) into MFMA A-operand
// fragments: out[((t*MB + m)*2 + hl)*64 + lane] = 8 bf16 {W[frag_k(t,qd,jj)][16m + (lane&15)]}, hl 0 = hi, 1 = lo.
__global__ void k_pack_weight_frags(const float *__restrict__ W, int K, int F_out, uint4 *__restrict__ out) {
  const int MB = F_out / 16, KT = K / 32;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= KT * MB * 64) return;
  const int lane = idx & 63, m = (idx >> 6) % MB, t = (idx >> 6) / MB;
  const int qd = lane >> 4, f = 16 * m + (lane & 15);
  bf16x8 hi, lo;
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) {
    const float w = W[(int64_t)frag_k(t, qd, jj) * F_out + f];
    const __bf16 h = (__bf16)w;
    hi[jj] = h;
    lo[jj] = (__bf16)(w - (float)h);
  }
  out[((t * MB + m) * 2 + 0) * 64 + lane] = __builtin_bit_cast(uint4, hi);
  out[((t * MB + m) * 2 + 1) * 64 + lane] = __builtin_bit_cast(uint4, lo);
}

__device__ __forceinline__ void split8(const float4 &a, const float4 &b, bf16x8 &hi, bf16x8 &lo) {
  const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const __bf16 h = (__bf16)v[j];
    hi[j] = h;
    lo[j] = (__bf16)(v[j] - (float)h);
  }
}

__device__ __forceinline__ f32x4 mfma3(const bf16x8 &wh, const bf16x8 &wl, const bf16x8 &dh, const bf16x8 &dl, f32x4 acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, dl, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, dh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, dh, acc, 0, 0, 0);
  return acc;
}

template <int FP, int FS>
__global__ __launch_bounds__(256, (FP + FS <= 128) ? 2 : 1) void k_fused_side(FusedArgs a) {
  constexpr int KT_S = FS / 32, MB_S = FUSED_H / 16;          // small GEMM: FS -> 32
  constexpr int KT_X = FP / 32, KT_B = KT_X + 1, MB_B = FUSED_D / 16;   // big GEMM: FP + 32 -> 64
  extern __shared__ __attribute__((aligned(16))) int32_t smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, qd = lane >> 4;

  // XCD-aware bijective remap: workgroups b, b+8, b+16.. share an XCD (round-robin dispatch), give each XCD
  // a contiguous range of work items so neighbouring tiles of one snapshot chunk meet in one L2.
  const int W = gridDim.x, b = blockIdx.x;
  const int q8 = W / 8, r8 = W % 8, xcd = b % 8;
  const int w = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + b / 8;
  const int tile = w % a.n_tiles, chunk_id = w / a.n_tiles;
  const int32_t *hd = a.hdr + tile * TILE_HDR_INTS;
  const int n_own = hd[0], n_prim = hd[1], n_sec = hd[2], n_inc = hd[3], pool_off = hd[5], sd = hd[6], meta_len = hd[7];
  if (!((a.side_mask >> sd) & 1)) return;
  const FusedSide &S_ = a.side[sd];

  int32_t *meta = smem;
  float *s_self = reinterpret_cast<float *>(smem + a.meta_cap);
  float *s_nbr = s_self + a.p_cap;
  float *sec = s_nbr + a.p_cap;
  float *hx = sec + a.q_cap * SEC_STRIDE;

  for (int i = tid; i < meta_len; i += 256) meta[i] = a.pool[pool_off + i];
  __syncthreads();
  const int32_t *prim_ids = meta;
  const int32_t *sec_ids = prim_ids + n_prim;
  const int32_t *inc_ptr = sec_ids + n_sec;
  const int32_t *inc_loc = inc_ptr + n_prim + 1;
  int32_t *inc_w = const_cast<int32_t *>(inc_loc) + n_inc;
  const int32_t *adj_ptr = inc_w + n_inc;
  const int32_t *adj_loc = adj_ptr + n_own + 1;
  for (int i = tid; i < n_inc; i += 256) reinterpret_cast<float *>(inc_w)[i] = S_.ne_val[inc_w[i]];
  const float *inc_val = reinterpret_cast<const float *>(inc_w);

  // weights -> registers, once per workgroup
  bf16x8 wsh[KT_S][MB_S], wsl[KT_S][MB_S], wbh[KT_B][MB_B], wbl[KT_B][MB_B];
#pragma unroll
  for (int t = 0; t < KT_S; ++t)
#pragma unroll
    for (int m = 0; m < MB_S; ++m) {
      wsh[t][m] = __builtin_bit_cast(bf16x8, S_.w_small[((t * MB_S + m) * 2 + 0) * 64 + lane]);
      wsl[t][m] = __builtin_bit_cast(bf16x8, S_.w_small[((t * MB_S + m) * 2 + 1) * 64 + lane]);
    }
#pragma unroll
  for (int t = 0; t < KT_B; ++t)
#pragma unroll
    for (int m = 0; m < MB_B; ++m) {
      wbh[t][m] = __builtin_bit_cast(bf16x8, S_.w_big[((t * MB_B + m) * 2 + 0) * 64 + lane]);
      wbl[t][m] = __builtin_bit_cast(bf16x8, S_.w_big[((t * MB_B + m) * 2 + 1) * 64 + lane]);
    }
  f32x4 bs[MB_S], as4[MB_B], an4[MB_B];
#pragma unroll
  for (int m = 0; m < MB_S; ++m) {
    bs[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (S_.b_small) bs[m] = *reinterpret_cast<const f32x4 *>(S_.b_small + 16 * m + 4 * qd);
  }
#pragma unroll
  for (int m = 0; m < MB_B; ++m) {
    as4[m] = *reinterpret_cast<const f32x4 *>(S_.a_self + 16 * m + 4 * qd);
    an4[m] = *reinterpret_cast<const f32x4 *>(S_.a_nbr + 16 * m + 4 * qd);
  }
  const int c16 = lane & 15, rs = lane >> 4;     // P3 mapping: 16 lanes x float4 per output row, 4 rows per wave
  f32x4 bo = f32x4{0.f, 0.f, 0.f, 0.f};
  if (S_.b_out) bo = *reinterpret_cast<const f32x4 *>(S_.b_out + 4 * c16);
  __syncthreads();

  const int s_end = min(a.S, (chunk_id + 1) * a.chunk);
  for (int s = chunk_id * a.chunk; s < s_end; ++s) {
    // ---------------- P1: secondary MLP -> LDS ----------------
    for (int blk = wave; blk * 16 < n_sec; blk += 4) {
      const int lrow = blk * 16 + r16;
      const int id = sec_ids[min(lrow, n_sec - 1)];
      const float *src = S_.sec_in + ((int64_t)s * S_.n_sec_glob + id) * FS + 4 * qd;
      bf16x8 dh[KT_S], dl[KT_S];
#pragma unroll
      for (int t = 0; t < KT_S; ++t) {
        const float4 v0 = *reinterpret_cast<const float4 *>(src + 32 * t);
        const float4 v1 = *reinterpret_cast<const float4 *>(src + 32 * t + 16);
        split8(v0, v1, dh[t], dl[t]);
      }
      f32x4 acc[MB_S];
#pragma unroll
      for (int m = 0; m < MB_S; ++m) acc[m] = bs[m];
#pragma unroll
      for (int t = 0; t < KT_S; ++t)
#pragma unroll
        for (int m = 0; m < MB_S; ++m) acc[m] = mfma3(wsh[t][m], wsl[t][m], dh[t], dl[t], acc[m]);
      if (lrow < n_sec) {
#pragma unroll
        for (int m = 0; m < MB_S; ++m) {
          f32x4 o;
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = apply_act(acc[m][j], a.act);
          *reinterpret_cast<f32x4 *>(sec + lrow * SEC_STRIDE + 16 * m + 4 * qd) = o;
        }
      }
    }
    __syncthreads();
    // ---------------- P2: [prim | agg] @ Wbig -> hx, attention scalars -> LDS ----------------
    for (int blk = wave; blk * 16 < n_prim; blk += 4) {
      const int lrow = blk * 16 + r16;
      const bool valid = lrow < n_prim;
      const int lr = min(lrow, n_prim - 1);
      const int id = prim_ids[lr];
      const float *src = S_.prim_in + ((int64_t)s * S_.n_prim_glob + id) * FP + 4 * qd;
      bf16x8 dh[KT_B], dl[KT_B];
#pragma unroll
      for (int t = 0; t < KT_X; ++t) {
        const float4 v0 = *reinterpret_cast<const float4 *>(src + 32 * t);
        const float4 v1 = *reinterpret_cast<const float4 *>(src + 32 * t + 16);
        split8(v0, v1, dh[t], dl[t]);
      }
      float4 g0 = make_float4(0.f, 0.f, 0.f, 0.f), g1 = g0;   // this lane's 8 aggregate features (fragment shape)
      for (int p = inc_ptr[lr]; p < inc_ptr[lr + 1]; ++p) {
        const float wv = inc_val[p];
        const float *row = sec + inc_loc[p] * SEC_STRIDE + 4 * qd;
        const float4 u0 = *reinterpret_cast<const float4 *>(row);
        const float4 u1 = *reinterpret_cast<const float4 *>(row + 16);
        g0.x = fmaf(wv, u0.x, g0.x); g0.y = fmaf(wv, u0.y, g0.y); g0.z = fmaf(wv, u0.z, g0.z); g0.w = fmaf(wv, u0.w, g0.w);
        g1.x = fmaf(wv, u1.x, g1.x); g1.y = fmaf(wv, u1.y, g1.y); g1.z = fmaf(wv, u1.z, g1.z); g1.w = fmaf(wv, u1.w, g1.w);
      }
      split8(g0, g1, dh[KT_X], dl[KT_X]);
      f32x4 acc[MB_B];
#pragma unroll
      for (int m = 0; m < MB_B; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < KT_B; ++t)
#pragma unroll
        for (int m = 0; m < MB_B; ++m) acc[m] = mfma3(wbh[t][m], wbl[t][m], dh[t], dl[t], acc[m]);
      float ps = 0.f, pn = 0.f;
#pragma unroll
      for (int m = 0; m < MB_B; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          ps = fmaf(acc[m][j], as4[m][j], ps);
          pn = fmaf(acc[m][j], an4[m][j], pn);
        }
      ps += __shfl_xor(ps, 16); pn += __shfl_xor(pn, 16);
      ps += __shfl_xor(ps, 32); pn += __shfl_xor(pn, 32);
      if (valid) {
        if (qd == 0) {
          s_self[lrow] = ps;
          s_nbr[lrow] = pn;
        }
#pragma unroll
        for (int m = 0; m < MB_B; ++m)   // chunk index XOR (row & 7): the 8 lanes of a write group hit 8 different slots
          *reinterpret_cast<f32x4 *>(hx + lrow * FUSED_D + (((4 * m + qd) ^ (lrow & 7)) << 2)) = acc[m];
      }
    }
    __syncthreads();
    // ---------------- P3: segmented softmax + neighbour sum -> HBM ----------------
    for (int i0 = wave * 4; i0 < n_own; i0 += 16) {
      const int i = i0 + rs;
      if (i < n_own) {
        const float ss = s_self[i];
        const int beg = adj_ptr[i], end = adj_ptr[i + 1];
        float mx = -INFINITY;
        for (int p = beg; p < end; ++p) mx = fmaxf(mx, leaky02(ss + s_nbr[adj_loc[p]]));
        float den = 0.f;
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int p = beg; p < end; ++p) {
          const int j = adj_loc[p];
          const float wgt = expf(leaky02(ss + s_nbr[j]) - mx);
          const f32x4 hv = *reinterpret_cast<const f32x4 *>(hx + j * FUSED_D + ((c16 ^ (j & 7)) << 2));
          den += wgt;
#pragma unroll
          for (int k = 0; k < 4; ++k) acc[k] = fmaf(wgt, hv[k], acc[k]);
        }
        const float inv = 1.0f / den;
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = apply_act(fmaf(acc[k], inv, bo[k]), a.act);
        *reinterpret_cast<f32x4 *>(S_.out + ((int64_t)s * S_.n_prim_glob + prim_ids[i]) * FUSED_D + 4 * c16) = o;
      }
    }
    // no barrier needed here: the next snapshot's P1 only writes `sec`, whose readers (P2) all passed the barrier
    // above; its P2 writes hx / s_* only after the next P1->P2 barrier, which every wave reaches after its own P3.
  }
}

}  // namespace uds
