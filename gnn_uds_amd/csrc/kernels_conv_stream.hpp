// Causal dilated Conv1D(kernel 3) as a TIME-STREAMING kernel on the matrix cores (gfx950): every input row is read
// from HBM once, instead of once per tap (keras Conv1D(H, 3, padding='causal', dilation_rate=D) on (B,T,R,F) rows,
// emulator.py:155-157,244-257,299-310).
//
// A wave owns 16 consecutive rows r of one batch element and walks the T time steps in order.  Step t DMAs the block
// x[t] (4 KB, fragment order) into its LDS ring, splits it once (bf16 hi + lo) and multiplies it by all three taps:
//   out[t] += x[t] W_2,   out[t+D] += x[t] W_1,   out[t+2D] += x[t] W_0
// (a causal window seen from the input side: zero padding needs no code, contributions past T are dropped).  The
// 2D+1 output blocks in progress live in a register ring of accumulators; out[t] is complete after step t: bias,
// activation, transposed through a small LDS tile, stored as whole 128-B lines.  The time loop is unrolled 2D+1 times
// so every ring index is a compile-time register.  dir = -1 walks time backwards: with transposed taps that is the
// input gradient of the causal layer.  Weights (48 KB pre-split) are staged once per workgroup in LDS and read per
// k-step; the ring keeps PREF time steps of DMA in flight; the slot a step has just consumed doubles as its store tile.
// The activation is a template parameter: a run-time switch per output element tripled the kernel's time.
//
// HBM-bound: 4 (F + H) bytes per row.  Numerics: the split-bf16 3-product scheme of the other MFMA kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels_rowgemm.hpp"

namespace uds {

struct ConvStreamArgs {
  const float *x, *bias;
  const uint4 *packed;        // k_pack_weight_frags layout of the (3*64, 64) kernel: k-step kt = 2*tap + half
  float *out;
  int B, T, R, act, dir;      // dir = +1 causal (forward), -1 look-ahead (input gradient)
  int n_blocks;               // ceil(R / 16)
  int n_seg, seg_len;         // the T steps are cut into n_seg segments of seg_len steps (each re-reads a 2D-step halo)
};

#ifndef UDS_CS_WAVES
#define UDS_CS_WAVES 8
#endif
#ifndef UDS_CS_PREF
#define UDS_CS_PREF 2
#endif
constexpr int CS_WAVES = UDS_CS_WAVES, CS_PREF = UDS_CS_PREF, CS_RING = CS_PREF + 1;      // one workgroup per CU (two waves per SIMD): 48 KB weights + waves x ring

template <int D, int ACT>
__global__ __launch_bounds__(CS_WAVES * 64, 2) void k_conv3_stream(ConvStreamArgs a) {
  constexpr int F = 64, MB = 4, KT = 6, NSET = 2 * D + 1, LD = 36;
  extern __shared__ __attribute__((aligned(16))) float smem_cs[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, qd = lane >> 4;
  uint4 *wlds = reinterpret_cast<uint4 *>(smem_cs);                       // KT * MB * 2 fragments x 64 lanes
  float *bias_s = smem_cs + KT * MB * 2 * 64 * 4;
  float *ring = bias_s + 64 + wave * (CS_RING * 1024);
  for (int i = tid; i < KT * MB * 2 * 64; i += CS_WAVES * 64) wlds[i] = a.packed[i];
  if (tid < 64) bias_s[tid] = a.bias ? a.bias[tid] : 0.f;
  __syncthreads();

  const int unit = blockIdx.x * CS_WAVES + wave;          // (time segment, batch element, 16-row block)
  if (unit >= a.n_seg * a.B * a.n_blocks) return;
  const int seg = unit / (a.B * a.n_blocks), bn = unit - seg * (a.B * a.n_blocks);
  const int b = bn / a.n_blocks, nb = bn - b * a.n_blocks;
  const int s0 = seg * a.seg_len, s1 = min(a.T, s0 + a.seg_len);       // logical steps whose outputs this wave writes
  const int t_first = max(0, s0 - 2 * D);                                // ... after the halo steps that feed them
  const int n_valid = min(16, a.R - nb * 16);
  const int64_t row0 = (int64_t)b * a.T * a.R + nb * 16;   // row of time step 0
  const float *src_lane = a.x + (row0 + min(r16, n_valid - 1)) * F + 4 * qd;
  const int64_t t_stride = (int64_t)a.R * F;
  const unsigned my_lds = __builtin_amdgcn_readfirstlane(lds_addr(ring));
  auto tmem = [&](int t) { return a.dir > 0 ? t : a.T - 1 - t; };         // memory time index of logical step t
  auto issue = [&](int t, int slot) {                                     // past the end: dummy re-fetch of the last step
    const float *s = src_lane + (int64_t)tmem(min(t, s1 - 1)) * t_stride;
    const float *pc[4] = {s, s + 16, s + 32, s + 48};
    glds16_run<4>(pc, my_lds + (unsigned)slot * 4096);
  };
#pragma unroll
  for (int q = 0; q < CS_PREF; ++q) issue(t_first + q, q);

  f32x4 acc[NSET][MB];
#pragma unroll
  for (int s = 0; s < NSET; ++s)
#pragma unroll
    for (int m = 0; m < MB; ++m) acc[s][m] = f32x4{0.f, 0.f, 0.f, 0.f};
  int slot = 0;            // ring slot of the step consumed next

  auto mult = [&](int tap, f32x4 (&dst)[MB], const bf16x8 (&dh)[2], const bf16x8 (&dl)[2]) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int kt = 2 * tap + h;
#pragma unroll
      for (int m = 0; m < MB; ++m) {
        const bf16x8 wh = __builtin_bit_cast(bf16x8, wlds[((kt * MB + m) * 2 + 0) * 64 + lane]);
        const bf16x8 wl = __builtin_bit_cast(bf16x8, wlds[((kt * MB + m) * 2 + 1) * 64 + lane]);
        dst[m] = mfma3(wh, wl, dh[h], dl[h], dst[m]);
      }
    }
  };

  for (int t0 = t_first; t0 < s1; t0 += NSET) {
    static_for<NSET>([&](auto u_) {
      constexpr int U = decltype(u_)::value;
      constexpr int CUR = U, MID = (U + D) % NSET, FAR = (U + 2 * D) % NSET;
      const int t = t0 + U;
      if (t < s1) {
        int rs = slot + CS_PREF;
        rs = rs >= CS_RING ? rs - CS_RING : rs;
        issue(t + CS_PREF, rs);
        // x[t] landed once everything older than its 4 pieces is done: younger = 4 PREF pieces + 4 stores per finished step
        const int st = min(max(t - s0, 0), CS_PREF);
        bool waited = false;
        static_for<CS_PREF>([&](auto k_) {
          constexpr int K = decltype(k_)::value;
          if (st == K) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * CS_PREF + 4 * K) : "memory");
            waited = true;
          }
        });
        if (!waited) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * CS_PREF) : "memory");
        const float4 *sl = reinterpret_cast<const float4 *>(ring + slot * 1024) + lane;
        const float4 v0 = sl[0], v1 = sl[64], v2 = sl[128], v3 = sl[192];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // slot consumed before a later trip refills it
        float *tile = ring + slot * 1024;                      // ... and free until the next step's DMA: it doubles as the store tile
        slot = slot + 1 == CS_RING ? 0 : slot + 1;
        bf16x8 dh[2], dl[2];
        split8(v0, v1, dh[0], dl[0]);
        split8(v2, v3, dh[1], dl[1]);
        mult(2, acc[CUR], dh, dl);
        mult(1, acc[MID], dh, dl);
        mult(0, acc[FAR], dh, dl);
        // out[t] is complete: bias, activation, 16 x 64 block through the tile in two 32-column halves -> 4 stores
        // (halo steps only feed later outputs: their own block is dropped)
        const int64_t orow = row0 + (int64_t)tmem(t) * a.R;
        if (t < s0) {
#pragma unroll
          for (int m = 0; m < MB; ++m) acc[CUR][m] = f32x4{0.f, 0.f, 0.f, 0.f};
        } else
#pragma unroll
        for (int g = 0; g < 2; ++g) {
#pragma unroll
          for (int mm = 0; mm < 2; ++mm) {
            const int m = 2 * g + mm;
            const f32x4 bb = *reinterpret_cast<const f32x4 *>(bias_s + 16 * m + 4 * qd);
            f32x4 o = acc[CUR][m];
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = fused_act<ACT>(o[j] + bb[j], a.act);      // relu compiled in, others decided at run time
            *reinterpret_cast<f32x4 *>(tile + r16 * LD + 16 * mm + 4 * qd) = o;
            acc[CUR][m] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int rr = min(i * 8 + (lane >> 3), n_valid - 1), cc = 4 * (lane & 7);    // rows past the block repeat the last valid row
            const f32x4 v = *reinterpret_cast<const f32x4 *>(tile + rr * LD + cc);
            *reinterpret_cast<f32x4 *>(a.out + (orow + rr) * 64 + 32 * g + cc) = v;
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
      }
    });
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the dummy pieces land before the LDS is released
}

inline int64_t conv_stream_lds_bytes() {
  return (int64_t)6 * 4 * 2 * 1024 + 256 + CS_WAVES * CS_RING * 4096;
}

template <int D, int ACT>
inline hipError_t launch_conv_stream_a(const ConvStreamArgs &a, hipStream_t st) {
  static unsigned long long attr_done = 0;
  if (hipError_t e = set_max_lds_once(reinterpret_cast<const void *>(&k_conv3_stream<D, ACT>), 160 * 1024, attr_done); e != hipSuccess) return e;
  const int units = a.n_seg * a.B * a.n_blocks;
  hipLaunchKernelGGL((k_conv3_stream<D, ACT>), dim3((unsigned)((units + CS_WAVES - 1) / CS_WAVES)), dim3(CS_WAVES * 64),
                     (size_t)conv_stream_lds_bytes(), st, a);
  return hipGetLastError();
}

template <int D>
inline hipError_t launch_conv_stream_t(const ConvStreamArgs &a, hipStream_t st) {
  if (a.act == 1) return launch_conv_stream_a<D, 1>(a, st);     // relu
  if (a.act == 0) return launch_conv_stream_a<D, 0>(a, st);     // linear (the input-gradient pass)
  return launch_conv_stream_a<D, -1>(a, st);
}

// taps = 3, F = 64, f_out = 64, |dil| in {1, 2, 4}: the shapes of every temporal layer of a d = H = 64 emulator
inline bool conv_stream_supported(int taps, int F, int fo, int dil) {
  const int d = dil < 0 ? -dil : dil;
  return taps == 3 && F == 64 && fo == 64 && (d == 1 || d == 2 || d == 4);
}

inline hipError_t launch_conv_stream(ConvStreamArgs a, int dil, hipStream_t st) {
  const int d = dil < 0 ? -dil : dil;
  // enough independent streams for ONE resident round (256 CUs x CS_WAVES waves), but segments long enough that the
  // 2D-step halo each one re-reads stays small
  const int streams = a.B * a.n_blocks;
  int n_seg = (256 * CS_WAVES) / streams;                 // round down: everything resident in ONE round
  n_seg = std::max(1, std::min(n_seg, a.T / (4 * d)));      // a segment re-reads 2d halo steps: keep that <= 50 %
  a.seg_len = (a.T + n_seg - 1) / n_seg;
  a.n_seg = (a.T + a.seg_len - 1) / a.seg_len;
  switch (d) {
    case 1: return launch_conv_stream_t<1>(a, st);
    case 2: return launch_conv_stream_t<2>(a, st);
    default: return launch_conv_stream_t<4>(a, st);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Dense(64 -> 64) followed by the prefix sum over time, the residual and the activation in ONE pass:
//   out[b,t,r,:] = act( sum_{t' <= t} (x[b,t',r,:] W + bias) + res[b,0,r,:] )
// (`Dense` named dense_resx, `cumsum(x_out, axis=1) + res`, activation: emulator.py:313-320).  Same streaming scheme as
// k_conv3_stream: a wave owns 16 rows of one batch element and walks the T steps; the running sum never leaves the MFMA
// accumulators (x W accumulates across steps), the bias enters as (t+1) * bias, the residual block sits in registers.
// Replaces a row GEMM + a prefix-sum kernel (2 reads + 2 writes of the tensor) by 1 read + 1 write.
// ---------------------------------------------------------------------------------------------------------------
struct DenseCumsumArgs {
  const float *x, *bias, *res;      // res (B, 1, R, 64) or nullptr
  const uint4 *packed;              // k_pack_weight_frags layout of the (64, 64) kernel
  float *out;
  int B, T, R, act, n_blocks;
};

constexpr int DC_WAVES = 2, DC_PREF = 4, DC_RING = DC_PREF + 1;

template <int ACT>
__global__ __launch_bounds__(DC_WAVES * 64, 2) void k_dense_cumsum_stream(DenseCumsumArgs a) {
  constexpr int F = 64, MB = 4, KT = 2, LD = 36;
  extern __shared__ __attribute__((aligned(16))) float smem_dc[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, qd = lane >> 4;
  uint4 *wlds = reinterpret_cast<uint4 *>(smem_dc);
  float *ring = smem_dc + KT * MB * 2 * 64 * 4 + wave * (DC_RING * 1024);
  for (int i = tid; i < KT * MB * 2 * 64; i += DC_WAVES * 64) wlds[i] = a.packed[i];
  __syncthreads();
  const int unit = blockIdx.x * DC_WAVES + wave;
  if (unit >= a.B * a.n_blocks) return;
  const int b = unit / a.n_blocks, nb = unit - b * a.n_blocks;
  const int n_valid = min(16, a.R - nb * 16);
  const int64_t row0 = (int64_t)b * a.T * a.R + nb * 16;
  const int my_row = min(r16, n_valid - 1);
  const float *src_lane = a.x + (row0 + my_row) * F + 4 * qd;
  const int64_t t_stride = (int64_t)a.R * F;
  const unsigned my_lds = __builtin_amdgcn_readfirstlane(lds_addr(ring));
  auto issue = [&](int t, int slot) {
    const float *s = src_lane + (int64_t)min(t, a.T - 1) * t_stride;
    const float *pc[4] = {s, s + 16, s + 32, s + 48};
    glds16_run<4>(pc, my_lds + (unsigned)slot * 4096);
  };
#pragma unroll
  for (int q = 0; q < DC_PREF; ++q) issue(q, q);
  // weights, bias and the residual block in accumulator layout: lane (r16, qd) <-> out[row r16][16 m + 4 qd + j]
  bf16x8 wh[KT][MB], wl[KT][MB];
  f32x4 bb[MB], rr[MB], run[MB];
#pragma unroll
  for (int m = 0; m < MB; ++m) {
#pragma unroll
    for (int h = 0; h < KT; ++h) {
      wh[h][m] = __builtin_bit_cast(bf16x8, wlds[((h * MB + m) * 2 + 0) * 64 + lane]);
      wl[h][m] = __builtin_bit_cast(bf16x8, wlds[((h * MB + m) * 2 + 1) * 64 + lane]);
    }
    bb[m] = a.bias ? *reinterpret_cast<const f32x4 *>(a.bias + 16 * m + 4 * qd) : f32x4{0.f, 0.f, 0.f, 0.f};
    rr[m] = a.res ? *reinterpret_cast<const f32x4 *>(a.res + ((int64_t)b * a.R + nb * 16 + my_row) * F + 16 * m + 4 * qd)
                  : f32x4{0.f, 0.f, 0.f, 0.f};
    run[m] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  int slot = 0;
  for (int t = 0; t < a.T; ++t) {
    int rs = slot + DC_PREF;
    rs = rs >= DC_RING ? rs - DC_RING : rs;
    issue(t + DC_PREF, rs);
    const int st = min(t, DC_PREF);
    bool waited = false;
    static_for<DC_PREF>([&](auto k_) {
      constexpr int K = decltype(k_)::value;
      if (st == K) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * DC_PREF + 4 * K) : "memory");
        waited = true;
      }
    });
    if (!waited) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * DC_PREF) : "memory");
    const float4 *sl = reinterpret_cast<const float4 *>(ring + slot * 1024) + lane;
    const float4 v0 = sl[0], v1 = sl[64], v2 = sl[128], v3 = sl[192];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float *tile = ring + slot * 1024;          // the consumed slot doubles as the store tile until the next step's DMA
    slot = slot + 1 == DC_RING ? 0 : slot + 1;
    bf16x8 dh[2], dl[2];
    split8(v0, v1, dh[0], dl[0]);
    split8(v2, v3, dh[1], dl[1]);
#pragma unroll
    for (int h = 0; h < KT; ++h)
#pragma unroll
      for (int m = 0; m < MB; ++m) run[m] = mfma3(wh[h][m], wl[h][m], dh[h], dl[h], run[m]);
    const float tb = (float)(t + 1);
    const int64_t orow = row0 + (int64_t)t * a.R;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
#pragma unroll
      for (int mm = 0; mm < 2; ++mm) {
        const int m = 2 * g + mm;
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = fused_act<ACT>(run[m][j] + fmaf(tb, bb[m][j], rr[m][j]), a.act);
        *reinterpret_cast<f32x4 *>(tile + r16 * LD + 16 * mm + 4 * qd) = o;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int r_ = min(i * 8 + (lane >> 3), n_valid - 1), cc = 4 * (lane & 7);
        const f32x4 v = *reinterpret_cast<const f32x4 *>(tile + r_ * LD + cc);
        *reinterpret_cast<f32x4 *>(a.out + (orow + r_) * 64 + 32 * g + cc) = v;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

inline int64_t dense_cumsum_lds_bytes() { return (int64_t)2 * 4 * 2 * 1024 + DC_WAVES * DC_RING * 4096; }

template <int ACT>
inline hipError_t launch_dense_cumsum_a(const DenseCumsumArgs &a, hipStream_t st) {
  const int units = a.B * a.n_blocks;
  hipLaunchKernelGGL((k_dense_cumsum_stream<ACT>), dim3((unsigned)((units + DC_WAVES - 1) / DC_WAVES)), dim3(DC_WAVES * 64),
                     (size_t)dense_cumsum_lds_bytes(), st, a);
  return hipGetLastError();
}

inline hipError_t launch_dense_cumsum(const DenseCumsumArgs &a, hipStream_t st) {
  if (a.act == 1) return launch_dense_cumsum_a<1>(a, st);
  if (a.act == 0) return launch_dense_cumsum_a<0>(a, st);
  return launch_dense_cumsum_a<-1>(a, st);
}

// ---------------------------------------------------------------------------------------------------------------------
// k_dense_cumsum_stream with the emulator's output heads as its epilogue (emulator.py:313-338): the 64-wide resnet output
//     y[t] = act(cumsum_t(x W + b) + res)
// is consumed where it is produced -- by `out` = Dense(n_a <= 4, act_a)(y) and, optionally, the flood chain
// Dense(32, act) x n_hidden (64 -> 32 -> 32 -> 32) + Dense(1, act_f) -- and only the n_a (+ 1) head outputs per row are written:
// y never reaches HBM (one 256-byte row write and three row reads fewer per row and step, and five to eight launches fewer).
// The accumulator layout of one layer (lane (r16, qd): features 16 m + 4 qd + q of row r16) is the B-operand fragment
// layout of the next (frag_k), so the chain runs from registers: split8 of two accumulator blocks = one k-step.
struct HeadsArgs {
  const uint4 *a_packed;            // (64, n_a) padded to 16 outputs: 2 k-steps x 1 block
  const float *a_bias;              // n_a floats
  const uint4 *h_packed[5];         // hidden layers: (64, 32), then up to four (32, 32)
  const float *h_bias[5];
  const uint4 *f_packed;            // (32, 1) padded to 16 outputs
  const float *f_bias;
  int n_a, act_a, n_hidden, act_h, act_f;
};

// two waves per SIMD (the head chain is a serial dependency per step: a second wave fills its gaps): 4 waves per workgroup with a
// ring of 3 x 4 KiB each + 16 KiB of weights + 16 KiB for the 32 x 32 hidden layers = 80 KiB, two workgroups per CU
constexpr int DH_WAVES = 4, DH_PREF = 2, DH_RING = DH_PREF + 1;

template <int ACT>
__global__ __launch_bounds__(DH_WAVES * 64, 2) void k_dense_cumsum_heads(DenseCumsumArgs a, HeadsArgs hd) {
  constexpr int F = 64, MB = 4, KT = 2;
  extern __shared__ __attribute__((aligned(16))) float smem_dc[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, qd = lane >> 4;
  uint4 *wlds = reinterpret_cast<uint4 *>(smem_dc);
  uint4 *hlds = wlds + KT * MB * 2 * 64;                          // hidden layers 2..5 of the second head: 4 fragments x 64 lanes each
  float *hbl = reinterpret_cast<float *>(hlds + 4 * 4 * 64);        // their biases (4 x 32 floats): no global loads inside the time loop,
                                                                    // whose compiler-placed vmcnt waits would drain the DMA prefetch
  float *ring = smem_dc + (KT * MB * 2 * 64 + 4 * 4 * 64) * 4 + 128 + wave * (DH_RING * 1024);
  for (int i = tid; i < KT * MB * 2 * 64; i += DH_WAVES * 64) wlds[i] = a.packed[i];
  for (int l = 1; l < hd.n_hidden; ++l) {
    for (int i = tid; i < 4 * 64; i += DH_WAVES * 64) hlds[(l - 1) * 256 + i] = hd.h_packed[l][i];
    if (tid < 32) hbl[(l - 1) * 32 + tid] = hd.h_bias[l] ? hd.h_bias[l][tid] : 0.0f;
  }
  __syncthreads();
  const int unit = blockIdx.x * DH_WAVES + wave;
  if (unit >= a.B * a.n_blocks) return;
  const int b = unit / a.n_blocks, nb = unit - b * a.n_blocks;
  const int n_valid = min(16, a.R - nb * 16);
  const int64_t row0 = (int64_t)b * a.T * a.R + nb * 16;
  const int my_row = min(r16, n_valid - 1);
  const float *src_lane = a.x + (row0 + my_row) * F + 4 * qd;
  const int64_t t_stride = (int64_t)a.R * F;
  const unsigned my_lds = __builtin_amdgcn_readfirstlane(lds_addr(ring));
  auto issue = [&](int t, int slot) {
    const float *s = src_lane + (int64_t)min(t, a.T - 1) * t_stride;
    const float *pc[4] = {s, s + 16, s + 32, s + 48};
    glds16_run<4>(pc, my_lds + (unsigned)slot * 4096);
  };
#pragma unroll
  for (int q = 0; q < DH_PREF; ++q) issue(q, q);
  // the 64 x 64 kernel stays in LDS and is re-read every step (the head fragments take the registers it had in
  // k_dense_cumsum_stream); the laundered lane index keeps the reads inside the time loop
  int wl_lane = lane;
  f32x4 bb[MB], rr[MB], run[MB];
#pragma unroll
  for (int m = 0; m < MB; ++m) {
    bb[m] = a.bias ? *reinterpret_cast<const f32x4 *>(a.bias + 16 * m + 4 * qd) : f32x4{0.f, 0.f, 0.f, 0.f};
    rr[m] = a.res ? *reinterpret_cast<const f32x4 *>(a.res + ((int64_t)b * a.R + nb * 16 + my_row) * F + 16 * m + 4 * qd)
                  : f32x4{0.f, 0.f, 0.f, 0.f};
    run[m] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // head weights as register-resident fragments (22 of them), biases in accumulator layout
  auto frag = [&](const uint4 *p, int kt, int mb, int m, int hl) __attribute__((always_inline)) {
    return __builtin_bit_cast(bf16x8, p[((kt * mb + m) * 2 + hl) * 64 + lane]);
  };
  auto bias4 = [&](const float *p, int n, int m) __attribute__((always_inline)) {      // features 16 m + 4 qd + q, zero past n
    f32x4 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = (p && 16 * m + 4 * qd + q < n) ? p[16 * m + 4 * qd + q] : 0.0f;
    return v;
  };
  bf16x8 ah[KT], al[KT], h1h[KT][2], h1l[KT][2], fh, fl;
  f32x4 ab = bias4(hd.a_bias, hd.n_a, 0), h1b[2], fb = bias4(hd.f_bias, 1, 0);
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) ah[kt] = frag(hd.a_packed, kt, 1, 0, 0), al[kt] = frag(hd.a_packed, kt, 1, 0, 1);
  if (hd.n_hidden > 0) {
#pragma unroll
    for (int m = 0; m < 2; ++m) {
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) h1h[kt][m] = frag(hd.h_packed[0], kt, 2, m, 0), h1l[kt][m] = frag(hd.h_packed[0], kt, 2, m, 1);
      h1b[m] = bias4(hd.h_bias[0], 32, m);
    }
    fh = frag(hd.f_packed, 0, 1, 0, 0), fl = frag(hd.f_packed, 0, 1, 0, 1);
  }
  const int n_out = hd.n_a + (hd.n_hidden > 0 ? 1 : 0);
  int slot = 0;
  for (int t = 0; t < a.T; ++t) {
    int rs = slot + DH_PREF;
    rs = rs >= DH_RING ? rs - DH_RING : rs;
    issue(t + DH_PREF, rs);
    // x[t] landed once everything older than its 4 pieces is done: younger = 4 pieces per prefetched step (the head outputs are
    // plain stores of a few bytes per row: they count as vector-memory operations too, one per step, hence the + K below)
    const int st = min(t, DH_PREF);
    bool waited = false;
    static_for<DH_PREF>([&](auto k_) {
      constexpr int K = decltype(k_)::value;
      if (st == K) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * DH_PREF) : "memory");
        waited = true;
      }
    });
    if (!waited) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * DH_PREF) : "memory");
    const float4 *sl = reinterpret_cast<const float4 *>(ring + slot * 1024) + lane;
    const float4 v0 = sl[0], v1 = sl[64], v2 = sl[128], v3 = sl[192];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    slot = slot + 1 == DH_RING ? 0 : slot + 1;
    bf16x8 dh[2], dl[2];
    split8(v0, v1, dh[0], dl[0]);
    split8(v2, v3, dh[1], dl[1]);
    asm volatile("" : "+v"(wl_lane));
#pragma unroll
    for (int h = 0; h < KT; ++h)
#pragma unroll
      for (int m = 0; m < MB; ++m)
        run[m] = mfma3(__builtin_bit_cast(bf16x8, wlds[((h * MB + m) * 2 + 0) * 64 + wl_lane]),
                       __builtin_bit_cast(bf16x8, wlds[((h * MB + m) * 2 + 1) * 64 + wl_lane]), dh[h], dl[h], run[m]);
    const float tb = (float)(t + 1);
    f32x4 y[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int j = 0; j < 4; ++j) y[m][j] = fused_act<ACT>(run[m][j] + fmaf(tb, bb[m][j], rr[m][j]), a.act);
    bf16x8 yh[KT], yl[KT];
    split8(*reinterpret_cast<const float4 *>(&y[0]), *reinterpret_cast<const float4 *>(&y[1]), yh[0], yl[0]);
    split8(*reinterpret_cast<const float4 *>(&y[2]), *reinterpret_cast<const float4 *>(&y[3]), yh[1], yl[1]);
    f32x4 oa = ab;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) oa = mfma3(ah[kt], al[kt], yh[kt], yl[kt], oa);
    float of = 0.0f;
    if (hd.n_hidden > 0) {
      f32x4 c1[2] = {h1b[0], h1b[1]};
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int m = 0; m < 2; ++m) c1[m] = mfma3(h1h[kt][m], h1l[kt][m], yh[kt], yl[kt], c1[m]);
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) c1[m][j] = apply_act(c1[m][j], hd.act_h);
      bf16x8 ch, cl;
      split8(*reinterpret_cast<const float4 *>(&c1[0]), *reinterpret_cast<const float4 *>(&c1[1]), ch, cl);
      for (int l = 1; l < hd.n_hidden; ++l) {      // 32 -> 32 layers: fragments from LDS (index [(m * 2 + hl) * 64 + lane]), bias from memory
        f32x4 cn[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          cn[m] = *reinterpret_cast<const f32x4 *>(hbl + (l - 1) * 32 + 16 * m + 4 * qd);
          cn[m] = mfma3(__builtin_bit_cast(bf16x8, hlds[(l - 1) * 256 + (m * 2 + 0) * 64 + wl_lane]),
                        __builtin_bit_cast(bf16x8, hlds[(l - 1) * 256 + (m * 2 + 1) * 64 + wl_lane]), ch, cl, cn[m]);
        }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int j = 0; j < 4; ++j) cn[m][j] = apply_act(cn[m][j], hd.act_h);
        split8(*reinterpret_cast<const float4 *>(&cn[0]), *reinterpret_cast<const float4 *>(&cn[1]), ch, cl);
      }
      f32x4 cf = mfma3(fh, fl, ch, cl, fb);
      of = apply_act(cf[0], hd.act_f);
    }
    if (qd == 0 && r16 < n_valid) {       // lane (r16, 0) holds outputs 0..3 of its row
      float *o = a.out + (row0 + (int64_t)t * a.R + r16) * n_out;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (j < hd.n_a) o[j] = apply_act(oa[j], hd.act_a);
      if (hd.n_hidden > 0) o[hd.n_a] = of;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

inline hipError_t launch_dense_cumsum_heads(const DenseCumsumArgs &a, const HeadsArgs &hd, hipStream_t st) {
  const int units = a.B * a.n_blocks;
  const dim3 grid((unsigned)((units + DH_WAVES - 1) / DH_WAVES)), block(DH_WAVES * 64);
  const size_t lds = (size_t)2 * 4 * 2 * 1024 + (size_t)4 * 4 * 1024 + 512 + (size_t)DH_WAVES * DH_RING * 4096;
  if (a.act == 1) hipLaunchKernelGGL((k_dense_cumsum_heads<1>), grid, block, lds, st, a, hd);
  else if (a.act == 0) hipLaunchKernelGGL((k_dense_cumsum_heads<0>), grid, block, lds, st, a, hd);
  else hipLaunchKernelGGL((k_dense_cumsum_heads<-1>), grid, block, lds, st, a, hd);
  return hipGetLastError();
}

}  // namespace uds
