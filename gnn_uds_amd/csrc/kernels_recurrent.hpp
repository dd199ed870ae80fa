// keras GRU / LSTM(64, return_sequences=True) on 64-wide rows as ONE time-streaming kernel per layer on the matrix cores
// (gfx950).  emulator.py:158-161: `recurrent: GRU` is the reference's default (`main.py` / utils/config.yaml), 6 of its 86
// shipped model configurations use LSTM.
//
// A wave owns 16 consecutive rows (node or link series) of one batch element and walks the T steps.  Per step
//     a = x[t] W + h U + biases          two 16 x 64 x (G*64) products on MFMA, split-bf16 (3 products, fp32 accumulate)
//     gates, new state h' (and c' for the LSTM) in registers, h' stored as the layer output
// With weights as the MFMA A operand and the 16 data rows as B, a lane ends up with features 16m + 4qd + {0..3} of row
// lane & 15 for every 16-feature block m -- which, under this library's k ordering (frag_k: a lane's 8 operand elements are
// the two 16-byte pieces 32t + 4qd and 32t + 16 + 4qd), is exactly the B-operand fragment of k-step t = m / 2: the new
// state feeds the next step's `h U` product straight from the accumulator registers, no LDS round trip, no transpose.
// x[t] is read from HBM in the same fragment shape (64 contiguous bytes per row and instruction), the next step's rows are
// in flight while the current one is multiplied; the input projection is never written to memory.
// Weights: W and U as 2*G packed 64 x 64 slices (uds_rowgemm_pack layout, 16 KB each: 96 KB for the GRU, 128 KB for the
// LSTM) staged once per workgroup in LDS; 4 waves per workgroup, one per SIMD (the gate state + biases take ~200 VGPRs).
// Gate conventions (TF 2.10 / Keras): GRU reset_after=True, order z, r, h; LSTM order i, f, c, o; sigmoid / tanh.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels_fused.hpp"

namespace uds {

struct RecurrentMfmaArgs {
  const float *x, *b_in, *b_rec;      // x (B, T, R, 32 KTX) -- or, KTX = 0, the input projection incl. bias (B, T, R, G*64); biases (G*64)
  const uint4 *packed;                 // G slices of W (KTX k-steps each), then G slices of U (2 k-steps): [(kt * 4 + m) * 2 + hl] * 64 + lane
  float *out;                          // (B, T, R, 64)
  int B, T, R, n_blocks;               // n_blocks = ceil(R / 16)
};

constexpr int RC_WAVES = 4;

// sigmoid / tanh on v_exp_f32 + v_rcp_f32 (1 ulp each): an IEEE fp32 division costs ten instructions, and a step has 48 of them
__device__ __forceinline__ float rc_sigmoid(float v) { return __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
__device__ __forceinline__ float rc_tanh(float v) {
  const float e = __expf(-2.0f * fabsf(v));          // tanh(|v|) = (1 - e) / (1 + e): no overflow
  return copysignf((1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e), v);
}

// KTX: k-steps of the input rows (2: 64-wide, 4: 128-wide -- the first temporal layer of a d = 128 model), 0: the input projection
// is given (computed by the row-GEMM kernel: any input width, and the LSTM at 128 whose W + U do not fit the LDS together)
template <int G, int KTX>
__global__ __launch_bounds__(RC_WAVES * 64) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_recurrent_mfma(RecurrentMfmaArgs a) {
  constexpr int MB = 4, KT = 2, USLICE = KT * MB * 2 * 64, WSLICE = KTX * MB * 2 * 64;      // uint4 per packed slice
  constexpr int FX = KTX ? 32 * KTX : G * 64, NX = KTX ? 2 * KTX : G * MB;                  // floats per input row, float4 per lane and step
  extern __shared__ __attribute__((aligned(16))) uint4 wl_rc[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, qd = lane >> 4;
  for (int i = tid; i < G * (WSLICE + USLICE); i += RC_WAVES * 64) wl_rc[i] = a.packed[i];
  __syncthreads();
  const int unit = blockIdx.x * RC_WAVES + wave;       // (batch element, 16-row block)
  if (unit >= a.B * a.n_blocks) return;
  const int b = unit / a.n_blocks, nb = unit - b * a.n_blocks;
  const int n_valid = min(16, a.R - nb * 16);
  const bool live = r16 < n_valid;
  const int64_t row0 = (int64_t)b * a.T * a.R + nb * 16 + min(r16, n_valid - 1);      // this lane's row at t = 0
  const int64_t t_stride = (int64_t)a.R * 64, x_stride = (int64_t)a.R * FX;
  const float *xl = a.x + row0 * FX + 4 * qd;
  float *ol = a.out + row0 * 64 + 4 * qd;

  // biases in accumulator layout: feature 16 m + 4 qd + q of gate g
  f32x4 bi[G][MB], br[MB];                             // br: recurrent bias of the candidate gate (GRU: multiplied by r)
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int m = 0; m < MB; ++m) {
      f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};             // KTX = 0: the given projection carries its biases (b_in is not read)
      if constexpr (KTX != 0) {
        v = *reinterpret_cast<const f32x4 *>(a.b_in + g * 64 + 16 * m + 4 * qd);
        if (G == 3 && g < 2 && a.b_rec) v += *reinterpret_cast<const f32x4 *>(a.b_rec + g * 64 + 16 * m + 4 * qd);
      }
      bi[g][m] = v;
    }
#pragma unroll
  for (int m = 0; m < MB; ++m)
    br[m] = (G == 3 && a.b_rec) ? *reinterpret_cast<const f32x4 *>(a.b_rec + 2 * 64 + 16 * m + 4 * qd) : f32x4{0.f, 0.f, 0.f, 0.f};

  int wl_lane = lane;      // laundered once per step: the weight fragments are re-read from LDS every step instead of being
                           // hoisted out of the time loop (2 * G * 16 fragments = 384+ VGPRs: they would spill)
  auto wfrag = [&](int mat, int g, int kt, int m, int hl) __attribute__((always_inline)) {
    return __builtin_bit_cast(bf16x8, wl_rc[(mat ? G * WSLICE + g * USLICE : g * WSLICE) + ((kt * MB + m) * 2 + hl) * 64 + wl_lane]);
  };
  // One GROUP = one k-step of one 64 x 64 slice: 8 weight fragments (4 feature blocks x hi / lo) and 12 MFMAs on four
  // independent accumulators (hi*lo for all four, lo*hi for all four, hi*hi for all four: a dependent MFMA is four
  // instructions away).  The fragments of group i+1 are fetched from LDS before the MFMAs of group i are issued
  // (sched_barrier keeps the compiler from sinking the reads back to their use, where each would expose its full latency
  // to this SIMD's only wave).
  bf16x8 fh[2][MB], fl[2][MB];
  auto load_group = [&](int buf, int mat, int g, int kt) __attribute__((always_inline)) {
#pragma unroll
    for (int m = 0; m < MB; ++m) fh[buf][m] = wfrag(mat, g, kt, m, 0), fl[buf][m] = wfrag(mat, g, kt, m, 1);
  };
  auto mma_group = [&](int buf, f32x4 (&acc)[MB], const bf16x8 &dh, const bf16x8 &dl) __attribute__((always_inline)) {
#pragma unroll
    for (int m = 0; m < MB; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh[buf][m], dl, acc[m], 0, 0, 0);
#pragma unroll
    for (int m = 0; m < MB; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fl[buf][m], dh, acc[m], 0, 0, 0);
#pragma unroll
    for (int m = 0; m < MB; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh[buf][m], dh, acc[m], 0, 0, 0);
  };
  f32x4 h[MB], c[MB];
#pragma unroll
  for (int m = 0; m < MB; ++m) h[m] = c[m] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 hh[KT], hl[KT];                               // the state as B-operand fragments (zero at t = 0)
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) hh[kt] = hl[kt] = __builtin_bit_cast(bf16x8, make_uint4(0u, 0u, 0u, 0u));

  // The x part of a step does not depend on the state: it is computed one step AHEAD, issued between the state product and
  // the gate arithmetic of the current step, so the matrix pipe works on it while the vector ALU does the gates.
  constexpr int NGH = 2 * G, NGX = KTX * G;            // groups of the state product / of the x part: gate g = i / k-steps, k-step i % k-steps
  f32x4 pre[G][MB];                                    // biases + x[t] W of the step about to run
  float4 xn[NX];                                       // input of the step after that: pieces 4 qd and 16 + 4 qd of every k-step
  auto load_x = [&](int t) __attribute__((always_inline)) {
    const float *nx = xl + (int64_t)t * x_stride;
#pragma unroll
    for (int i = 0; i < NX; ++i) xn[i] = *reinterpret_cast<const float4 *>(nx + 16 * i);
  };
  auto x_part = [&]() __attribute__((always_inline)) { // pre = biases + xn W   (KTX = 0: pre = the given projection)
    if constexpr (KTX == 0) {
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int m = 0; m < MB; ++m) pre[g][m] = f32x4{xn[g * MB + m].x, xn[g * MB + m].y, xn[g * MB + m].z, xn[g * MB + m].w};
    } else {
      bf16x8 xh[KTX ? KTX : 1], xlo[KTX ? KTX : 1];
#pragma unroll
      for (int k = 0; k < KTX; ++k) split8(xn[2 * k], xn[2 * k + 1], xh[k], xlo[k]);
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int m = 0; m < MB; ++m) pre[g][m] = bi[g][m];
      load_group(0, 0, 0, 0);
      static_for<NGX>([&](auto i_) {
        constexpr int I = decltype(i_)::value, g = I / (KTX ? KTX : 1), kt = I % (KTX ? KTX : 1), buf = I % 2;
        if (I + 1 < NGX) load_group(buf ^ 1, 0, (I + 1) / (KTX ? KTX : 1), (I + 1) % (KTX ? KTX : 1));
        __builtin_amdgcn_sched_barrier(0);
        mma_group(buf, pre[g], xh[kt], xlo[kt]);
        __builtin_amdgcn_sched_barrier(0);
      });
    }
  };
  load_x(0);
  x_part();
  if (a.T > 1) load_x(1);
  for (int t = 0; t < a.T; ++t) {
    asm volatile("" : "+v"(wl_lane));
    f32x4 acc[G][MB], ah[MB];                          // ah: the candidate gate's recurrent part (GRU: multiplied by r)
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int m = 0; m < MB; ++m) acc[g][m] = pre[g][m];
#pragma unroll
    for (int m = 0; m < MB; ++m) ah[m] = br[m];
    load_group(0, 1, 0, 0);
    static_for<NGH>([&](auto i_) {
      constexpr int I = decltype(i_)::value, g = I / 2, kt = I % 2, buf = I % 2;
      if (I + 1 < NGH) load_group(buf ^ 1, 1, (I + 1) / 2, (I + 1) % 2);
      __builtin_amdgcn_sched_barrier(0);
      mma_group(buf, (G == 3 && g == 2) ? ah : acc[g], hh[kt], hl[kt]);
      __builtin_amdgcn_sched_barrier(0);
    });
    if (t + 1 < a.T) {
      x_part();                                        // next step's x W: in the matrix pipe during the gates below
      if (t + 2 < a.T) load_x(t + 2);
    }
    f32x4 hn[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (G == 3) {
          const float z = rc_sigmoid(acc[0][m][q]), r = rc_sigmoid(acc[1][m][q]);
          const float cand = rc_tanh(acc[2][m][q] + r * ah[m][q]);
          hn[m][q] = z * h[m][q] + (1.0f - z) * cand;
        } else {
          c[m][q] = rc_sigmoid(acc[1][m][q]) * c[m][q] + rc_sigmoid(acc[0][m][q]) * rc_tanh(acc[2][m][q]);
          hn[m][q] = rc_sigmoid(acc[G - 1][m][q]) * rc_tanh(c[m][q]);
        }
      }
#pragma unroll
    for (int m = 0; m < MB; ++m) h[m] = hn[m];
    // the new state in B-operand shape: blocks (0, 1) are k-step 0, (2, 3) k-step 1
    split8(*reinterpret_cast<const float4 *>(&h[0]), *reinterpret_cast<const float4 *>(&h[1]), hh[0], hl[0]);
    split8(*reinterpret_cast<const float4 *>(&h[2]), *reinterpret_cast<const float4 *>(&h[3]), hh[1], hl[1]);
    if (live) {
      float *o = ol + (int64_t)t * t_stride;
#pragma unroll
      for (int m = 0; m < MB; ++m) *reinterpret_cast<f32x4 *>(o + 16 * m) = h[m];
    }
  }
}

// LDS bytes of the (G, KTX) instantiation
inline size_t recurrent_mfma_lds(int G, int ktx) { return (size_t)G * (ktx + 2) * (4 * 2 * 64) * sizeof(uint4); }

template <int G, int KTX>
inline hipError_t launch_recurrent_mfma_t(const RecurrentMfmaArgs &a, hipStream_t st) {
  static unsigned long long attr_done = 0;                        // once per kernel (never inside a stream capture after the warm-up call)
  const size_t lds = recurrent_mfma_lds(G, KTX);
  if (hipError_t e = set_max_lds_once(reinterpret_cast<const void *>(&k_recurrent_mfma<G, KTX>), (int)lds, attr_done); e != hipSuccess) return e;
  const int units = a.B * a.n_blocks;
  hipLaunchKernelGGL((k_recurrent_mfma<G, KTX>), dim3((unsigned)((units + RC_WAVES - 1) / RC_WAVES)), dim3(RC_WAVES * 64), lds, st, a);
  return hipGetLastError();
}

// which input forms the one-launch kernel takes: F = 64 / 128 input rows when W + U fit the 160 KiB LDS, F = 0 = given projection
inline bool recurrent_mfma_supported(int G, int F) { return (F == 0 || F == 64 || F == 128) && recurrent_mfma_lds(G, F / 32) <= 160 * 1024; }

inline hipError_t launch_recurrent_mfma(const RecurrentMfmaArgs &a, int G, int F, hipStream_t st) {
  if (G == 3) return F == 0 ? launch_recurrent_mfma_t<3, 0>(a, st) : F == 64 ? launch_recurrent_mfma_t<3, 2>(a, st) : launch_recurrent_mfma_t<3, 4>(a, st);
  return F == 0 ? launch_recurrent_mfma_t<4, 0>(a, st) : launch_recurrent_mfma_t<4, 2>(a, st);
}

}  // namespace uds
