// Reverse mode of the keras GRU / LSTM(64, return_sequences=True) time recurrence (emulator.py:158-161 under the
// GradientTape of fit_eval, emulator.py:457-484): back-propagation through time as ONE time-streaming kernel on the matrix
// cores (gfx950), the mirror image of k_recurrent_mfma (kernels_recurrent.hpp).
//
// Forward (uds_recurrent_forward, exact fp32; gate order and reset_after=True form as in kernels_dense.hpp: k_recurrent):
//     a_rec = h[t-1] U + b_rec ;  GRU:  z = sig(xp_z + a_z), r = sig(xp_r + a_r), cand = tanh(xp_h + r a_h), h[t] = z h[t-1] + (1 - z) cand
//                                 LSTM: i, f, g, o from xp + a_rec ;  c[t] = f c[t-1] + i g ;  h[t] = o tanh(c[t])
// Backward: a wave owns 16 rows (series) of one batch element and walks t = T-1 .. 0 with the carried gradient dh (and dc):
//     1. a_rec is RECOMPUTED from the saved h[t-1] (one 16 x 64 x G*64 product, split-bf16, three MFMA products), the gates
//        from it and the saved input projection xp[t]: nothing but the layer output (and the LSTM's cell states) is kept
//        from the forward pass;
//     2. the gate derivatives give d_xp[t] (gradient of the input projection: handed to the Dense backward) and d_arec[t]
//        (gradient of the recurrent pre-activation; differs from d_xp only in the GRU's candidate gate: r * d_cand);
//     3. dh[t-1] = direct path + d_arec[t] U^T (one 16 x G*64 x 64 product).  The accumulator layout of step 2's values is
//        the B-operand fragment layout of step 3's product (the identity the forward kernel uses): no LDS round trip.
// d_arec is written gate-major, (G, B, T, R, 64), so that dU_g = h[t-1]^T d_arec_g is one call of the split-K weight-
// gradient kernel per gate (uds_wgrad with a time shift of one) and d b_rec its bias row.
// LDS: U as G packed 64 x 64 slices (uds_rowgemm_pack layout) and U^T as G more: 96 KB (GRU) / 128 KB (LSTM).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels_recurrent.hpp"

namespace uds {

struct RecurrentBwdArgs {
  const float *xp, *b_rec;             // xp (B, T, R, G*64) input projection incl. input bias; b_rec (G*64) or NULL
  const uint4 *packed;                 // G slices of U (64 x 64 each), then G slices of U_g^T: [(kt * 4 + m) * 2 + hl] * 64 + lane
  const float *h, *c, *gh;             // h (B, T, R, 64) forward output; c (B, T, R, 64) cell states (LSTM) or NULL; gh = dL/dh
  float *dxp, *darec;                  // dxp (B, T, R, G*64); darec (G, B, T, R, 64)
  int B, T, R, n_blocks;
};

template <int G>
__global__ __launch_bounds__(RC_WAVES * 64) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_recurrent_bwd(RecurrentBwdArgs a) {
  constexpr int MB = 4, KT = 2, SLICE = KT * MB * 2 * 64;      // uint4 per packed 64 x 64 slice
  extern __shared__ __attribute__((aligned(16))) uint4 wl_rb[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, qd = lane >> 4;
  for (int i = tid; i < 2 * G * SLICE; i += RC_WAVES * 64) wl_rb[i] = a.packed[i];
  __syncthreads();
  const int unit = blockIdx.x * RC_WAVES + wave;       // (batch element, 16-row block)
  if (unit >= a.B * a.n_blocks) return;
  const int b = unit / a.n_blocks, nb = unit - b * a.n_blocks;
  const int n_valid = min(16, a.R - nb * 16);
  const bool live = r16 < n_valid;
  const int64_t row0 = (int64_t)b * a.T * a.R + nb * 16 + min(r16, n_valid - 1);      // this lane's row at t = 0
  const int64_t t_rows = a.R;
  const int64_t gate_stride = (int64_t)a.B * a.T * a.R * 64;                           // floats between the gate planes of darec

  f32x4 br[G][MB];                                     // recurrent bias in accumulator layout: feature 16 m + 4 qd + q of gate g
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int m = 0; m < MB; ++m)
      br[g][m] = a.b_rec ? *reinterpret_cast<const f32x4 *>(a.b_rec + g * 64 + 16 * m + 4 * qd) : f32x4{0.f, 0.f, 0.f, 0.f};

  int wl_lane = lane;      // laundered once per step: the weight fragments are re-read from LDS every step, not hoisted out of the loop
  auto wfrag = [&](int slice, int kt, int m, int hl) __attribute__((always_inline)) {
    return __builtin_bit_cast(bf16x8, wl_rb[slice * SLICE + ((kt * MB + m) * 2 + hl) * 64 + wl_lane]);
  };
  // acc[m] += W_slice (k-step kt) x data fragment: three MFMA products per feature block
  auto mma = [&](int slice, int kt, f32x4 (&acc)[MB], const bf16x8 &dh_, const bf16x8 &dl_) __attribute__((always_inline)) {
#pragma unroll
    for (int m = 0; m < MB; ++m) {
      const bf16x8 wh = wfrag(slice, kt, m, 0), wl = wfrag(slice, kt, m, 1);
      acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, dl_, acc[m], 0, 0, 0);
      acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, dh_, acc[m], 0, 0, 0);
      acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, dh_, acc[m], 0, 0, 0);
    }
  };
  auto frag = [&](const f32x4 &u0, const f32x4 &u1, bf16x8 &hi, bf16x8 &lo) __attribute__((always_inline)) {
    split8(make_float4(u0[0], u0[1], u0[2], u0[3]), make_float4(u1[0], u1[1], u1[2], u1[3]), hi, lo);
  };

  f32x4 dh[MB], dc[MB];                                // carried gradients (zero beyond the last step)
#pragma unroll
  for (int m = 0; m < MB; ++m) dh[m] = dc[m] = f32x4{0.f, 0.f, 0.f, 0.f};

  // registers of the step being processed, loaded one step ahead
  f32x4 xq[G][MB], hp[MB], cp[MB], gq[MB];
  auto load_step = [&](int t) __attribute__((always_inline)) {
    const int64_t row = row0 + (int64_t)t * t_rows;
    const float *xr = a.xp + row * (G * 64) + 4 * qd;
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int m = 0; m < MB; ++m) xq[g][m] = *reinterpret_cast<const f32x4 *>(xr + g * 64 + 16 * m);
    const float *gr = a.gh + row * 64 + 4 * qd;
#pragma unroll
    for (int m = 0; m < MB; ++m) gq[m] = *reinterpret_cast<const f32x4 *>(gr + 16 * m);
    if (t > 0) {
      const float *hr = a.h + (row - t_rows) * 64 + 4 * qd;
#pragma unroll
      for (int m = 0; m < MB; ++m) hp[m] = *reinterpret_cast<const f32x4 *>(hr + 16 * m);
      if (G == 4) {
        const float *cr = a.c + (row - t_rows) * 64 + 4 * qd;
#pragma unroll
        for (int m = 0; m < MB; ++m) cp[m] = *reinterpret_cast<const f32x4 *>(cr + 16 * m);
      }
    } else {
#pragma unroll
      for (int m = 0; m < MB; ++m) hp[m] = cp[m] = f32x4{0.f, 0.f, 0.f, 0.f};      // zero initial state
    }
  };

  for (int t = a.T - 1; t >= 0; --t) {
    asm volatile("" : "+v"(wl_lane));
    load_step(t);
    // ---- 1. a_rec = h[t-1] U + b_rec ----
    bf16x8 hh[KT], hl[KT];
    frag(hp[0], hp[1], hh[0], hl[0]);
    frag(hp[2], hp[3], hh[1], hl[1]);
    f32x4 ar[G][MB];
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
      for (int m = 0; m < MB; ++m) ar[g][m] = br[g][m];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) mma(g, kt, ar[g], hh[kt], hl[kt]);
    }
    // ---- 2. gates and their derivatives ----
    f32x4 dx[G][MB], da[G][MB], direct[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float dht = gq[m][q] + dh[m][q];
        if (G == 3) {
          const float z = rc_sigmoid(xq[0][m][q] + ar[0][m][q]), r = rc_sigmoid(xq[1][m][q] + ar[1][m][q]);
          const float ah = ar[2][m][q];
          const float cand = rc_tanh(xq[2][m][q] + r * ah);
          const float dcand = dht * (1.0f - z) * (1.0f - cand * cand);      // d / d(pre-activation of the candidate)
          const float dz = dht * (hp[m][q] - cand) * z * (1.0f - z);
          const float dr = dcand * ah * r * (1.0f - r);
          dx[0][m][q] = dz; dx[1][m][q] = dr; dx[2][m][q] = dcand;
          da[0][m][q] = dz; da[1][m][q] = dr; da[2][m][q] = dcand * r;
          direct[m][q] = dht * z;
        } else {
          const float ig = rc_sigmoid(xq[0][m][q] + ar[0][m][q]), fg = rc_sigmoid(xq[1][m][q] + ar[1][m][q]);
          const float gg = rc_tanh(xq[2][m][q] + ar[2][m][q]), og = rc_sigmoid(xq[G - 1][m][q] + ar[G - 1][m][q]);
          const float ct = fg * cp[m][q] + ig * gg, tc = rc_tanh(ct);
          const float dct = dc[m][q] + dht * og * (1.0f - tc * tc);
          dx[0][m][q] = dct * gg * ig * (1.0f - ig);
          dx[1][m][q] = dct * cp[m][q] * fg * (1.0f - fg);
          dx[2][m][q] = dct * ig * (1.0f - gg * gg);
          dx[G - 1][m][q] = dht * tc * og * (1.0f - og);
#pragma unroll
          for (int g = 0; g < G; ++g) da[g][m][q] = dx[g][m][q];
          dc[m][q] = dct * fg;
          direct[m][q] = 0.f;
        }
      }
    if (live) {
      const int64_t row = row0 + (int64_t)t * t_rows;
      float *xo = a.dxp + row * (G * 64) + 4 * qd;
      float *ao = a.darec + row * 64 + 4 * qd;
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int m = 0; m < MB; ++m) {
          *reinterpret_cast<f32x4 *>(xo + g * 64 + 16 * m) = dx[g][m];
          *reinterpret_cast<f32x4 *>(ao + g * gate_stride + 16 * m) = da[g][m];
        }
    }
    // ---- 3. dh[t-1] = direct + sum_g d_arec_g U_g^T ----
#pragma unroll
    for (int m = 0; m < MB; ++m) dh[m] = direct[m];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      bf16x8 fh[KT], fl[KT];
      frag(da[g][0], da[g][1], fh[0], fl[0]);
      frag(da[g][2], da[g][3], fh[1], fl[1]);
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) mma(G + g, kt, dh, fh[kt], fl[kt]);
    }
  }
}

inline size_t recurrent_bwd_lds(int G) { return (size_t)2 * G * (2 * 4 * 2 * 64) * sizeof(uint4); }

template <int G>
inline hipError_t launch_recurrent_bwd_t(const RecurrentBwdArgs &a, hipStream_t st) {
  static unsigned long long attr_done = 0;
  const size_t lds = recurrent_bwd_lds(G);
  if (hipError_t e = set_max_lds_once(reinterpret_cast<const void *>(&k_recurrent_bwd<G>), (int)lds, attr_done); e != hipSuccess) return e;
  const int units = a.B * a.n_blocks;
  hipLaunchKernelGGL((k_recurrent_bwd<G>), dim3((unsigned)((units + RC_WAVES - 1) / RC_WAVES)), dim3(RC_WAVES * 64), lds, st, a);
  return hipGetLastError();
}

}  // namespace uds
