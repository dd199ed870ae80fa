// Row-local dense kernels (gfx950): out = act([xa | xb] @ W + bias), optional GAT attention scalars.
// Replaces keras Dense (emulator.py:225-226 ...) and the node-update einsum of Spektral GATConv.
//
// v1: exact-fp32 FMA tile kernel.  Each 4x4 register tile accumulates in ascending-k order with
// one fmaf per product, i.e. the same arithmetic as a k-ordered fp32 dot product.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>
#include <stdint.h>

#include "../../include/uds_hip.h"

namespace uds {

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute: `done` (one static word per kernel instantiation) keeps a
// bit per device ordinal, so a process that drives several GPUs raises the limit on each of them once.
inline hipError_t set_max_lds_once(const void *fn, int bytes, unsigned long long &done) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const unsigned long long bit = 1ull << (dev & 63);
  if (done & bit) return hipSuccess;
  e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) done |= bit;
  return e;
}


struct DenseArgs {
  const float *xa, *xb, *W, *bias, *a_self, *a_nbr;
  float *out, *s_self, *s_nbr;
  int fa, fb, fo, act;
  int64_t rows;
  // causal dilated Conv1D over time as a GEMM on time-shifted rows (taps = 0: plain dense).  Rows are (b, t, r) with
  // r fastest: tap j reads the row `(taps-1-j)*dil` time steps earlier (t_rows rows back) or zero outside [0, T)
  // (dil < 0 looks ahead instead: the input-gradient of the causal convolution).
  int taps = 0, dil = 1, T = 1, t_rows = 1;
};

__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case UDS_ACT_RELU: return fmaxf(v, 0.0f);
    case UDS_ACT_TANH: return tanhf(v);
    case UDS_ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
    case UDS_ACT_HARD_SIGMOID: return fminf(fmaxf(0.2f * v + 0.5f, 0.0f), 1.0f);
    default: return v;
  }
}

// The activation code is wave-uniform.  A `switch` per output ELEMENT compiles to chains of scalar branches around the
// inlined tanh / exp bodies (it made the streaming Conv1D 1.7x slower), so kernels call their epilogue through
// with_act(): one branch per epilogue, the common cases (relu, linear) compiled in.
template <int ACT>
__device__ __forceinline__ float act_ct(float v, int act_rt) {
  if constexpr (ACT == UDS_ACT_RELU) return fmaxf(v, 0.0f);
  else if constexpr (ACT == UDS_ACT_LINEAR) return v;
  else return apply_act(v, act_rt);
}
template <class F>
__device__ __forceinline__ void with_act(int act, F &&f) {
  if (act == UDS_ACT_RELU) f(std::integral_constant<int, UDS_ACT_RELU>{});
  else if (act == UDS_ACT_LINEAR) f(std::integral_constant<int, UDS_ACT_LINEAR>{});
  else f(std::integral_constant<int, -1>{});
}

// CG = number of 4-wide column groups (power of two, <= 64).  RT row-threads x TM rows each.
template <int CG>
__global__ __launch_bounds__(256) void k_dense_act(DenseArgs a) {
  constexpr int RT = (256 / CG) < 32 ? (256 / CG) : 32;
  constexpr int TM = 4;
  constexpr int ROWS = RT * TM;
  constexpr int NT = RT * CG;
  constexpr int KC = 32;
  __shared__ float As[ROWS][KC + 1];
  __shared__ __attribute__((aligned(16))) float Ws[KC][CG * 4];

  const int tid = threadIdx.x;
  const int cg = tid % CG;
  const int rt = tid / CG;
  const int64_t row0 = (int64_t)blockIdx.x * ROWS;
  const int F = a.taps ? a.taps * a.fa : a.fa + a.fb;

  float acc[TM][4];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.0f;

  for (int k0 = 0; k0 < F; k0 += KC) {
    for (int idx = tid; idx < ROWS * KC; idx += NT) {
      const int r = idx / KC, k = idx % KC;
      const int64_t grow = row0 + r;
      const int kk = k0 + k;
      float v = 0.0f;
      if (grow < a.rows && kk < F) {
        if (a.taps) {
          const int j = kk / a.fa, f = kk - j * a.fa;
          const int shift = (a.taps - 1 - j) * a.dil;
          const int t = (int)((grow / a.t_rows) % a.T);
          if (t - shift >= 0 && t - shift < a.T) v = a.xa[(grow - (int64_t)shift * a.t_rows) * a.fa + f];
        } else if (kk < a.fa) {
          v = a.xa[grow * a.fa + kk];
        } else {
          v = a.xb[grow * a.fb + (kk - a.fa)];
        }
      }
      As[r][k] = v;
    }
    for (int idx = tid; idx < KC * CG * 4; idx += NT) {
      const int k = idx / (CG * 4), c = idx % (CG * 4);
      Ws[k][c] = (k0 + k < F && c < a.fo) ? a.W[(int64_t)(k0 + k) * a.fo + c] : 0.0f;
    }
    __syncthreads();
#pragma unroll 8
    for (int k = 0; k < KC; ++k) {
      const float4 w = *reinterpret_cast<const float4 *>(&Ws[k][cg * 4]);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const float av = As[rt * TM + i][k];
        acc[i][0] = fmaf(av, w.x, acc[i][0]);
        acc[i][1] = fmaf(av, w.y, acc[i][1]);
        acc[i][2] = fmaf(av, w.z, acc[i][2]);
        acc[i][3] = fmaf(av, w.w, acc[i][3]);
      }
    }
    __syncthreads();
  }

  const int c0 = cg * 4;
  float as4[4] = {0, 0, 0, 0}, an4[4] = {0, 0, 0, 0}, b4[4] = {0, 0, 0, 0};
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (c0 + j < a.fo) {
      if (a.bias) b4[j] = a.bias[c0 + j];
      if (a.a_self) {
        as4[j] = a.a_self[c0 + j];
        an4[j] = a.a_nbr[c0 + j];
      }
    }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int64_t grow = row0 + rt * TM + i;
    if (a.a_self) {  // attention scalars on the pre-activation row (GAT linear has no bias / act)
      float ps = 0.0f, pn = 0.0f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        ps = fmaf(acc[i][j], as4[j], ps);
        pn = fmaf(acc[i][j], an4[j], pn);
      }
#pragma unroll
      for (int off = CG / 2; off > 0; off >>= 1) {  // the CG lanes of a row are contiguous in the wave
        ps += __shfl_xor(ps, off);
        pn += __shfl_xor(pn, off);
      }
      if (cg == 0 && grow < a.rows) {
        a.s_self[grow] = ps;
        a.s_nbr[grow] = pn;
      }
    }
    if (grow < a.rows) {
      float o[4];
      with_act(a.act, [&](auto act_) {
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = act_ct<decltype(act_)::value>(acc[i][j] + b4[j], a.act);
      });
      if ((a.fo & 3) == 0) {
        if (c0 < a.fo) *reinterpret_cast<float4 *>(&a.out[grow * a.fo + c0]) = make_float4(o[0], o[1], o[2], o[3]);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (c0 + j < a.fo) a.out[grow * a.fo + c0 + j] = o[j];
      }
    }
  }
}

template <int CG>
inline hipError_t launch_dense_cg(const DenseArgs &a, hipStream_t st) {
  constexpr int RT = (256 / CG) < 32 ? (256 / CG) : 32;
  constexpr int ROWS = RT * 4;
  const int64_t blocks = (a.rows + ROWS - 1) / ROWS;
  hipLaunchKernelGGL(k_dense_act<CG>, dim3((unsigned)blocks), dim3(RT * CG), 0, st, a);
  return hipGetLastError();
}

// Narrow inputs (the embeddings: 1..8 features -> d, emulator.py:198-212): a pure output stream.  A lane owns one float4
// column chunk and keeps its F x 4 weights in registers; the row's F inputs are scalar loads shared by the row's lanes;
// rows are walked with a grid stride so workgroups live long (the tiled kernel above spends its time launching
// 64-row workgroups: 2.2 TB/s of writes against 4.6 here).  Same fp32 fmaf order over k as the tiled kernel.
template <int F>
__global__ __launch_bounds__(256) void k_embed_act(DenseArgs a) {
  const int f4 = a.fo / 4;                       // lanes per row (fo % 4 == 0)
  const int64_t total = a.rows * f4;
  const int64_t stride = (int64_t)gridDim.x * 256;
  const int64_t t0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int c = (int)(t0 % f4);                  // stride % f4 == 0 (launcher): the chunk of a lane never changes
  float4 w[F];
#pragma unroll
  for (int k = 0; k < F; ++k) w[k] = *reinterpret_cast<const float4 *>(a.W + (int64_t)k * a.fo + 4 * c);
  float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
  if (a.bias) b = *reinterpret_cast<const float4 *>(a.bias + 4 * c);
  with_act(a.act, [&](auto act_) {
    constexpr int A = decltype(act_)::value;
    for (int64_t t = t0; t < total; t += stride) {
      const int64_t r = t / f4;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int k = 0; k < F; ++k) {
        const float x = a.xa[r * F + k];
        acc.x = fmaf(x, w[k].x, acc.x);
        acc.y = fmaf(x, w[k].y, acc.y);
        acc.z = fmaf(x, w[k].z, acc.z);
        acc.w = fmaf(x, w[k].w, acc.w);
      }
      float4 o;
      o.x = act_ct<A>(acc.x + b.x, a.act);
      o.y = act_ct<A>(acc.y + b.y, a.act);
      o.z = act_ct<A>(acc.z + b.z, a.act);
      o.w = act_ct<A>(acc.w + b.w, a.act);
      *reinterpret_cast<float4 *>(a.out + r * a.fo + 4 * c) = o;
    }
  });
}

template <int F>
inline hipError_t launch_embed_f(const DenseArgs &a, hipStream_t st) {
  const int f4 = a.fo / 4;
  const int64_t total = a.rows * f4;
  int64_t blocks = std::min<int64_t>((total + 255) / 256, 256 * 16);      // 16 workgroups of 4 waves per CU
  // grid stride a multiple of f4 so a lane keeps its column chunk: 256 * blocks % f4 == 0 needs blocks % (f4 / gcd(256, f4)) == 0
  int g = f4;
  for (int x = 256; x % 2 == 0 && g % 2 == 0; x /= 2) g /= 2;
  blocks = std::max<int64_t>(g, blocks / g * g);
  hipLaunchKernelGGL(k_embed_act<F>, dim3((unsigned)blocks), dim3(256), 0, st, a);
  return hipGetLastError();
}

inline hipError_t launch_dense_act(const DenseArgs &a, hipStream_t st) {
  if (a.fb == 0 && a.taps == 0 && a.a_self == nullptr && a.fa <= 8 && (a.fo & 3) == 0 && a.rows >= 64) {
    switch (a.fa) {
      case 1: return launch_embed_f<1>(a, st);
      case 2: return launch_embed_f<2>(a, st);
      case 3: return launch_embed_f<3>(a, st);
      case 4: return launch_embed_f<4>(a, st);
      case 5: return launch_embed_f<5>(a, st);
      case 6: return launch_embed_f<6>(a, st);
      case 7: return launch_embed_f<7>(a, st);
      default: return launch_embed_f<8>(a, st);
    }
  }
  const int groups = (a.fo + 3) / 4;
  if (groups <= 1) return launch_dense_cg<1>(a, st);
  if (groups <= 2) return launch_dense_cg<2>(a, st);
  if (groups <= 4) return launch_dense_cg<4>(a, st);
  if (groups <= 8) return launch_dense_cg<8>(a, st);
  if (groups <= 16) return launch_dense_cg<16>(a, st);
  if (groups <= 32) return launch_dense_cg<32>(a, st);
  return launch_dense_cg<64>(a, st);
}

// keras GRU / LSTM(units, return_sequences=True) along the time axis (emulator.py:158-161: the `recurrent` alternatives to
// the causal Conv1D; 6 of the reference's 86 shipped model configurations use LSTM, 5 fall back to the GRU default).  The input projections xp = x @ kernel + input bias for ALL time steps
// come from the Dense kernels; this kernel is the time recurrence, exact fp32.  A workgroup owns `rows` independent rows
// (node or link series) and walks t = 0..T-1: thread (row, f) forms the G gate pre-activations of feature f,
//     a_g = xp[b, t, n, g H + f] + sum_k h[row][k] U[k][g H + f] (+ recurrent bias),
// from the recurrent kernel U (H x G H, staged once in LDS) and the previous state (LDS), then
//   GRU (G = 3, gates z, r, h; TF2 default reset_after=True, recurrent_activation = sigmoid):
//       z = sig(a_z), r = sig(a_r): here a_r, a_z add the two projections; cand = tanh(xp_h + r * (h U_h + rb_h)); h' = z h + (1 - z) cand
//   LSTM (G = 4, gates i, f, c, o): c' = sig(a_f) c + sig(a_i) tanh(a_c); h' = sig(a_o) tanh(c')
// x is indexed as (B, T, R, .) directly: no (B*R, T, .) transpose as in the reference (emulator.py:244).
struct RecurrentArgs {
  const float *xp, *U, *rb;
  float *out;
  float *c_out;                 // LSTM cell state after every step (B, T, R, H), or NULL: what the backward kernel needs besides `out`
  int B, T, R, H, G, rows;      // rows per workgroup; blockDim = rows * H
};

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }

template <int G>
__global__ void k_recurrent(RecurrentArgs a) {
  extern __shared__ float smem_rc[];
  const int H = a.H, GH = G * H;
  float *U = smem_rc;                       // H x GH
  float *hs = smem_rc + (size_t)H * GH;     // rows x H
  const int tid = threadIdx.x, row = tid / H, f = tid - row * H;
  for (int i = tid; i < H * GH; i += blockDim.x) U[i] = a.U[i];
  const int64_t grow = (int64_t)blockIdx.x * a.rows + row;      // global row = b * R + n
  const bool live = grow < (int64_t)a.B * a.R;
  const int b = live ? (int)(grow / a.R) : 0, n = live ? (int)(grow - (int64_t)b * a.R) : 0;
  hs[row * H + f] = 0.0f;
  float rb[G];
#pragma unroll
  for (int g = 0; g < G; ++g) rb[g] = a.rb ? a.rb[g * H + f] : 0.0f;
  float c = 0.0f, h = 0.0f;
  __syncthreads();
  for (int t = 0; t < a.T; ++t) {
    float acc[G];
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = rb[g];
    const float *hr = hs + row * H;
    for (int k = 0; k < H; ++k) {
      const float hk = hr[k];
#pragma unroll
      for (int g = 0; g < G; ++g) acc[g] = fmaf(hk, U[k * GH + g * H + f], acc[g]);
    }
    float xg[G];
    const int64_t base = (((int64_t)b * a.T + t) * a.R + n);
#pragma unroll
    for (int g = 0; g < G; ++g) xg[g] = live ? a.xp[base * GH + g * H + f] : 0.0f;
    if (G == 3) {
      const float z = sigmoidf_(xg[0] + acc[0]), r = sigmoidf_(xg[1] + acc[1]);
      const float cand = tanhf(xg[2] + r * acc[2]);
      h = z * h + (1.0f - z) * cand;
    } else {
      const float ig = sigmoidf_(xg[0] + acc[0]), fg = sigmoidf_(xg[1] + acc[1]);
      const float cg = tanhf(xg[2] + acc[2]), og = sigmoidf_(xg[G - 1] + acc[G - 1]);
      c = fg * c + ig * cg;
      h = og * tanhf(c);
    }
    __syncthreads();                 // every thread has read the previous state
    hs[row * H + f] = h;
    if (live) a.out[base * H + f] = h;
    if (G == 4 && live && a.c_out) a.c_out[base * H + f] = c;
    __syncthreads();
  }
}

inline hipError_t launch_recurrent(const RecurrentArgs &a, hipStream_t st) {
  const int64_t total = (int64_t)a.B * a.R;
  const unsigned grid = (unsigned)((total + a.rows - 1) / a.rows);
  const size_t lds = ((size_t)a.H * a.G * a.H + (size_t)a.rows * a.H) * sizeof(float);
  static size_t attr_lds[2] = {0, 0};                 // raise the dynamic-LDS limit once per size (not inside a stream capture)
  if (lds > attr_lds[a.G - 3]) {
    hipError_t e = hipFuncSetAttribute(a.G == 3 ? reinterpret_cast<const void *>(k_recurrent<3>) : reinterpret_cast<const void *>(k_recurrent<4>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_lds[a.G - 3] = lds;
  }
  if (a.G == 3) hipLaunchKernelGGL(k_recurrent<3>, dim3(grid), dim3(a.rows * a.H), lds, st, a);
  else hipLaunchKernelGGL(k_recurrent<4>, dim3(grid), dim3(a.rows * a.H), lds, st, a);
  return hipGetLastError();
}

}  // namespace uds
