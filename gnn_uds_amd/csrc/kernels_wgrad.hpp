// Weight gradient of a row GEMM on the matrix cores (gfx950): dW[f, n] = sum_r A[r, f] * G[r, n] over ALL rows r
// (rows = S * N is 1e5..1e7, f <= 128, n <= 64) -- the `X^T dZ` products of the training step (emulator.py:457-484
// via GradientTape), plus the bias gradient (column sums of G) as one extra output row.  A library GEMM handles this
// tall-skinny reduction poorly (no split over the reduction dimension: ~480 us per call at 250k rows where the HBM
// floor is ~30 us), hence this kernel.
//
// HBM-bound: every wave streams its own contiguous range of rows once, 32 rows (one MFMA k-step) at a time, and keeps
// the whole (MT*16) x (NT*16) partial in accumulators.  Operand fragments are read straight from HBM in the transposed
// shape MFMA wants (lane (m, q) holds A[32 k0 + 8q + j][f0 + m], j = 0..7: sixteen lanes read 64 contiguous bytes of a
// row), split into bf16 hi + lo (three products, fp32 accumulate: the numerics of the forward kernels).  The waves of
// a workgroup add their partials in LDS in a fixed order, the workgroup writes ONE partial, and k_wgrad_reduce sums
// the workgroups' partials in index order: no atomics, bitwise reproducible.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels_fused.hpp"

namespace uds {

struct WgradArgs {
  const float *a, *g;        // A (rows, F) possibly time-shifted, G (rows, H)
  float *partial;            // (grid, MT*16, NT*16)
  int64_t rows;
  int F, H, ones_row;        // ones_row = F when the bias gradient is wanted (A[., F] := 1), else -1
  int shift, T, t_rows;      // causal tap: A row of r is r - shift * t_rows, zero where the time index of r is < shift
  int rows_per_wave;         // multiple of 32
};

template <int MT, int NT>
__global__ __launch_bounds__(256, 2) void k_wgrad_mfma(WgradArgs a) {
  __shared__ float red[MT * 16 * NT * 16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, qd = lane >> 4;
  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int64_t r_begin = ((int64_t)blockIdx.x * 4 + wave) * a.rows_per_wave;
  const int64_t r_end = min(a.rows, r_begin + a.rows_per_wave);

  for (int64_t r0 = r_begin; r0 < r_end; r0 += 32) {
    // this lane's 8 rows of the k-step, their validity and (for a causal tap) the shifted source row
    int64_t ra[8], rg[8];
    bool va[8], vg[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int64_t r = r0 + 8 * qd + j;
      vg[j] = r < r_end;
      rg[j] = vg[j] ? r : r_begin;
      va[j] = vg[j];
      ra[j] = rg[j];
      if (a.shift) {
        const int t = (int)((rg[j] / a.t_rows) % a.T);
        va[j] = vg[j] && t >= a.shift;
        ra[j] = va[j] ? rg[j] - (int64_t)a.shift * a.t_rows : r_begin;
      }
    }
    bf16x8 gh[NT], gl[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int c = 16 * n + r16;
      const bool cv = c < a.H;
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (cv && vg[j]) ? a.g[rg[j] * a.H + c] : 0.f;
      split8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), gh[n], gl[n]);
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int f = 16 * m + r16;
      float v[8];
      if (f < a.F) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = va[j] ? a.a[ra[j] * a.F + f] : 0.f;
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (f == a.ones_row && vg[j]) ? 1.f : 0.f;
      }
      bf16x8 ah, al;
      split8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), ah, al);
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[m][n] = mfma3(ah, al, gh[n], gl[n], acc[m][n]);
    }
  }

  // waves add their partials in wave order (fixed summation order), then the workgroup stores one partial
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float *p = red + (16 * m + 4 * qd + j) * (NT * 16) + 16 * n + r16;     // acc[m][n][j] = dW[16m + 4qd + j][16n + r16]
            *p = w == 0 ? acc[m][n][j] : *p + acc[m][n][j];
          }
    }
    __syncthreads();
  }
  float *out = a.partial + (int64_t)blockIdx.x * (MT * 16 * NT * 16);
  for (int i = tid; i < MT * 16 * NT * 16; i += 256) out[i] = red[i];
}

// out[f, n] = sum_b partial[b, f, n] for f < f_rows, n < H (partials are (MT*16) x (NT*16) padded).  A block sums 16
// outputs: 16 lanes per output take every 16th partial (independent loads in flight), then the 16 lane sums are added
// in lane order -- a fixed order, so the result is reproducible.
__global__ __launch_bounds__(256) void k_wgrad_reduce(const float *partial, int n_part, int ld_f, int ld_n, int f_rows, int H, float *out) {
  __shared__ float red[256];
  const int tid = threadIdx.x, o = tid & 15, bl = tid >> 4;
  const int i = blockIdx.x * 16 + o;
  const bool ok = i < f_rows * H;
  const int f = ok ? i / H : 0, n = ok ? i - f * H : 0;
  const float *p = partial + (int64_t)f * ld_n + n;
  const int64_t stride = (int64_t)ld_f * ld_n;
  float s = 0.f;
#pragma unroll 8
  for (int b = bl; b < n_part; b += 16) s += p[b * stride];
  red[tid] = s;
  __syncthreads();
  if (tid < 16 && ok) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k * 16 + tid];
    out[i] = t;
  }
}

template <int MT, int NT>
inline hipError_t launch_wgrad_t(const WgradArgs &a, int grid, hipStream_t st) {
  hipLaunchKernelGGL((k_wgrad_mfma<MT, NT>), dim3((unsigned)grid), dim3(256), 0, st, a);
  return hipGetLastError();
}

inline int wgrad_mt(int f_rows) {           // instantiated row-tile counts
  const int need = (f_rows + 15) / 16;
  const int have[6] = {1, 2, 3, 5, 7, 8};
  for (int k = 0; k < 6; ++k)
    if (have[k] >= need) return have[k];
  return 0;
}
inline int wgrad_nt(int H) { return H <= 16 ? 1 : (H <= 32 ? 2 : (H <= 64 ? 4 : 0)); }
inline int wgrad_grid(int64_t rows) { return (int)std::min<int64_t>(512, (rows + 127) / 128); }

inline hipError_t launch_wgrad(const WgradArgs &a, int mt, int nt, int grid, hipStream_t st) {
#define UDS_WG(M, N) if (mt == M && nt == N) return launch_wgrad_t<M, N>(a, grid, st);
  UDS_WG(1, 1) UDS_WG(1, 2) UDS_WG(1, 4) UDS_WG(2, 1) UDS_WG(2, 2) UDS_WG(2, 4) UDS_WG(3, 1) UDS_WG(3, 2) UDS_WG(3, 4)
  UDS_WG(5, 1) UDS_WG(5, 2) UDS_WG(5, 4) UDS_WG(7, 1) UDS_WG(7, 2) UDS_WG(7, 4) UDS_WG(8, 1) UDS_WG(8, 2) UDS_WG(8, 4)
#undef UDS_WG
  return hipErrorInvalidValue;
}

}  // namespace uds
