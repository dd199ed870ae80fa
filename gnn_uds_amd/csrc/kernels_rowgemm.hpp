// Row-local GEMM on the matrix cores (gfx950): out[r,:] = act(A[r,:] @ W + bias), where A[r,:] is either the row
// itself (keras Dense) or the concatenation of `taps` time-shifted rows (causal dilated Conv1D on a (B,T,R,F) tensor,
// emulator.py:155-157).  Same numerics as the fused spatial kernel: operands split into bf16 hi + lo, three
// v_mfma_f32_16x16x32_bf16 products, fp32 accumulation (~2^-16 relative per product).
//
// HBM-bound by design: every wave owns NB consecutive 16-row blocks and keeps their NB x MB accumulators in registers
// while it walks K in 64-wide chunks; a chunk's weight fragments (<= 64 VGPRs, pre-split, L2-resident) are loaded once
// per chunk and reused for all NB blocks; data rows are loaded straight from HBM in fragment shape (each 16-B piece
// once per tap), with the next block's loads in flight while the current block is split and multiplied.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels_fused.hpp"

namespace uds {

struct RowGemmArgs {
  const float *x, *bias;
  const uint4 *packed;      // k_pack_weight_frags layout, F_out padded to 16*MB
  float *out;
  int64_t rows;
  int F, taps, dil, T, t_rows, fo, act;   // A row = taps x F floats, K = taps * F (multiple of 32)
};

// Pack with zero padding of the output features up to mb*16 (heads have 1..3 outputs).
__global__ void k_pack_weight_frags_padded(const float *__restrict__ W, int K, int F_out, int MB, uint4 *__restrict__ out) {
  const int KT = K / 32;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= KT * MB * 64) return;
  const int lane = idx & 63, m = (idx >> 6) % MB, t = (idx >> 6) / MB;
  const int qd = lane >> 4, f = 16 * m + (lane & 15);
  bf16x8 hi, lo;
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) {
    const float w = f < F_out ? W[(int64_t)frag_k(t, qd, jj) * F_out + f] : 0.0f;
    const __bf16 h = (__bf16)w;
    hi[jj] = h;
    lo[jj] = (__bf16)(w - (float)h);
  }
  out[((t * MB + m) * 2 + 0) * 64 + lane] = __builtin_bit_cast(uint4, hi);
  out[((t * MB + m) * 2 + 1) * 64 + lane] = __builtin_bit_cast(uint4, lo);
}

template <int MB, int NB>
__global__ __launch_bounds__(256, 2) void k_rowgemm_mfma(RowGemmArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r16 = lane & 15, qd = lane >> 4;
  const int64_t base = ((int64_t)blockIdx.x * 4 + wave) * (NB * 16);
  if (base >= a.rows) return;
  const int KT = a.taps * a.F / 32;

  f32x4 acc[NB][MB];
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int m = 0; m < MB; ++m) acc[b][m] = f32x4{0.f, 0.f, 0.f, 0.f};

  // per block: this lane's row and its time index (for the causal zero padding)
  int row[NB], tix[NB];        // rows < 2^31 (checked by the launcher)
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    row[b] = (int)min(base + b * 16 + r16, a.rows - 1);
    tix[b] = (row[b] / a.t_rows) % a.T;
  }
  auto load = [&](int b, int t, float4 &v0, float4 &v1) {
    const int k0 = 32 * t;
    const int j = k0 / a.F, f0 = k0 - j * a.F;
    const int shift = (a.taps - 1 - j) * a.dil;
    const bool live = tix[b] >= shift;
    const float *src = a.x + (int64_t)(row[b] - (live ? shift * a.t_rows : 0)) * a.F + f0 + 4 * qd;
    v0 = *reinterpret_cast<const float4 *>(src);
    v1 = *reinterpret_cast<const float4 *>(src + 16);
    if (!live) v0 = v1 = make_float4(0.f, 0.f, 0.f, 0.f);
  };

  for (int t0 = 0; t0 < KT; t0 += 2) {           // 64-wide K chunk (the last one may be 32 wide)
    const int nt = min(2, KT - t0);
    bf16x8 wh[2][MB], wl[2][MB];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int m = 0; m < MB; ++m) {
        const int t = min(t0 + tt, KT - 1);
        wh[tt][m] = __builtin_bit_cast(bf16x8, a.packed[((t * MB + m) * 2 + 0) * 64 + lane]);
        wl[tt][m] = __builtin_bit_cast(bf16x8, a.packed[((t * MB + m) * 2 + 1) * 64 + lane]);
      }
    float4 cur[4], nxt[4];
    load(0, t0, cur[0], cur[1]);
    if (nt == 2) load(0, t0 + 1, cur[2], cur[3]);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if (b + 1 < NB) {                          // next block's rows are in flight while this one is multiplied
        load(b + 1, t0, nxt[0], nxt[1]);
        if (nt == 2) load(b + 1, t0 + 1, nxt[2], nxt[3]);
      }
      bf16x8 dh, dl;
      split8(cur[0], cur[1], dh, dl);
#pragma unroll
      for (int m = 0; m < MB; ++m) acc[b][m] = mfma3(wh[0][m], wl[0][m], dh, dl, acc[b][m]);
      if (nt == 2) {
        split8(cur[2], cur[3], dh, dl);
#pragma unroll
        for (int m = 0; m < MB; ++m) acc[b][m] = mfma3(wh[1][m], wl[1][m], dh, dl, acc[b][m]);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) cur[i] = nxt[i];
    }
  }

#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int64_t r = base + b * 16 + r16;
    if (r >= a.rows) continue;
#pragma unroll
    for (int m = 0; m < MB; ++m) {
      const int c0 = 16 * m + 4 * qd;
      f32x4 o = acc[b][m];
      if ((a.fo & 3) == 0) {
        if (c0 < a.fo) {
          if (a.bias) {
            const f32x4 bb = *reinterpret_cast<const f32x4 *>(a.bias + c0);
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] += bb[j];
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = apply_act(o[j], a.act);
          *reinterpret_cast<f32x4 *>(a.out + r * a.fo + c0) = o;
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (c0 + j < a.fo) a.out[r * a.fo + c0 + j] = apply_act(o[j] + (a.bias ? a.bias[c0 + j] : 0.f), a.act);
      }
    }
  }
}

template <int MB, int NB>
inline hipError_t launch_rowgemm_t(const RowGemmArgs &a, hipStream_t st) {
  const int64_t rows_per_wg = 4 * NB * 16;
  hipLaunchKernelGGL((k_rowgemm_mfma<MB, NB>), dim3((unsigned)((a.rows + rows_per_wg - 1) / rows_per_wg)), dim3(256), 0, st, a);
  return hipGetLastError();
}

inline int rowgemm_mb(int fo) { return fo <= 16 ? 1 : (fo <= 32 ? 2 : 4); }

inline hipError_t launch_rowgemm(const RowGemmArgs &a, hipStream_t st) {
  switch (rowgemm_mb(a.fo)) {
    case 1: return launch_rowgemm_t<1, 8>(a, st);
    case 2: return launch_rowgemm_t<2, 8>(a, st);
    default: return launch_rowgemm_t<4, 6>(a, st);
  }
}

}  // namespace uds
