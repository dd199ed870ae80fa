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

#include <algorithm>

#include "kernels_fused.hpp"

namespace uds {

struct RowGemmArgs {
  const float *x, *bias;
  const uint4 *packed;      // k_pack_weight_frags layout, F_out padded to 16*MB
  float *out;
  int64_t rows;
  int F, taps, dil, T, t_rows, fo, act;   // A row = taps x F floats, K = taps * F (multiple of 32)
  int seg;                  // seg > 0: XCD-aware mapping (see k_rowgemm_mfma), else consecutive wave-tiles
  // Dense only (taps = 1): the row may be the concatenation [x (F1 floats) | x2 (F - F1 floats)] of two tensors, and the
  // output may be a column block of a wider matrix (row stride ldo floats, first column col0)
  const float *x2 = nullptr;
  int F1 = 0, ldo = 0, col0 = 0;
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

// Two 1-KiB LDS-DMA pieces in one statement (one k-step of one 16-row block): src0 -> lds_byte, src1 -> +1 KiB.
__device__ __forceinline__ void glds16_pair(const float *src0, const float *src1, unsigned lds_byte) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
               "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(src0), "v"(src1), "s"(lds_byte)
               : "memory", "scc");
}

// Epilogue of a wave-tile: bias + activation, whole-row stores.  `tile` is the wave's 16 x LD staging tile in LDS.
template <int MB, int NB, bool STAGE = true>
__device__ __forceinline__ void rowgemm_epilogue(const RowGemmArgs &a, f32x4 (&acc)[NB][MB], int base, int nv, float *tile,
                                                 const float *bias_s, int lane) {
  constexpr int CG = MB >= 2 ? 2 : 1;
  constexpr int LD = 16 * CG + 4;
  const int r16 = lane & 15, qd = lane >> 4;
  with_act(a.act, [&](auto act_) {
    constexpr int ACT_ = decltype(act_)::value;
  if (a.fo == 16 * MB && MB >= 2) {
    // turn each 16 x 32 accumulator pair through the wave's tile so that one store instruction writes 8 rows x 128 B
    // (whole cache lines; 1 KiB contiguous when fo = 32) instead of sixteen 64-byte pieces
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int g = 0; g < MB / CG; ++g) {
#pragma unroll
        for (int mm = 0; mm < CG; ++mm) {
          const int m = g * CG + mm;
          const f32x4 bb = *reinterpret_cast<const f32x4 *>(bias_s + 16 * m + 4 * qd);
          f32x4 o = acc[b][m];
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = act_ct<ACT_>(o[j] + bb[j], a.act);
          *reinterpret_cast<f32x4 *>(tile + r16 * LD + 16 * mm + 4 * qd) = o;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int rr = i * 8 + (lane >> 3), cc = 4 * (lane & 7);
          const f32x4 v = *reinterpret_cast<const f32x4 *>(tile + rr * LD + cc);
          if (b * 16 + rr < nv) *reinterpret_cast<f32x4 *>(a.out + (int64_t)(base + b * 16 + rr) * a.ldo + a.col0 + 32 * g + cc) = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
  } else {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int64_t r = (int64_t)base + b * 16 + r16;
      if (b * 16 + r16 >= nv) continue;
#pragma unroll
      for (int m = 0; m < MB; ++m) {
        const int c0 = 16 * m + 4 * qd;
        f32x4 o = acc[b][m];
        if ((a.fo & 3) == 0) {
          if (c0 < a.fo) {
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = act_ct<ACT_>(o[j] + bias_s[c0 + j], a.act);
            *reinterpret_cast<f32x4 *>(a.out + r * a.ldo + a.col0 + c0) = o;
          }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (c0 + j < a.fo) a.out[r * a.ldo + a.col0 + c0 + j] = act_ct<ACT_>(o[j] + bias_s[c0 + j], a.act);
        }
      }
    }
  }
  });
}

// Persistent 8-wave workgroups, one per CU.  LDS: the packed weights of every k-step and the bias (staged once per
// workgroup) | per wave a ring of RING 2-KiB slots filled by LDS-DMA in fragment order (the lane that issues a 16-B
// piece reads it back, so the wave's own counted vmcnt is the only synchronisation) | per wave a 16 x 32 tile that
// turns accumulators into whole output rows.  A wave owns a stream of 64-row wave-tiles; a "piece" is one k-step (32
// of the K values) of one 16-row block, and the wave keeps RING-1 future pieces in flight -- across k-steps, blocks
// AND wave-tiles -- while it splits and multiplies the current one, so the prologue / epilogue of a tile overlaps
// the fetch of the next.  No VGPRs are spent on prefetch and the loop has no compiler-visible vector-memory loads
// (weights and bias come from LDS), so nothing but the counted waits below touches the DMA queue.
//
// vmcnt bookkeeping: every trip issues exactly one piece (2 DMA instructions; past the end of the stream it re-fetches
// the last tile, so the count never changes) and then waits for vmcnt <= 2 (RING-1).  The output stores of a finished
// tile are younger than the pieces already in flight, so this wait is at worst conservative (it may also wait for some
// stores), never too weak.
// STAGE = false drops the per-wave output tile (direct 64-byte stores from the accumulators): 18 KiB of LDS that buy a
// deeper ring when the weights are large (K = 384, the first Conv1D of a d = 128 emulator: ring 3 instead of 2).
template <int MB, int NB, int RING, bool STAGE = true>
__global__ __launch_bounds__(512, 2) void k_rowgemm_mfma(RowGemmArgs a) {
  static_assert(RING - 1 <= NB && RING >= 2, "the ring may reach at most one k-step ahead");
  constexpr int CG = MB >= 2 ? 2 : 1;            // accumulator fragments per transposed store group (32 columns)
  constexpr int LD = 16 * CG + 4;                // padded tile row (floats)
  extern __shared__ __attribute__((aligned(16))) float smem_rg[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, qd = lane >> 4;
  const int KT = a.taps * a.F / 32;
  const int n_w = KT * MB * 2 * 64;             // uint4 entries of the packed weights
  uint4 *wlds = reinterpret_cast<uint4 *>(smem_rg);
  float *bias_s = smem_rg + n_w * 4;
  float *ring = bias_s + 64 + wave * (RING * 512);
  float *tile = bias_s + 64 + 8 * (RING * 512) + wave * (16 * LD);
  for (int i = tid; i < n_w; i += 512) wlds[i] = a.packed[i];
  if (tid < 64) bias_s[tid] = (a.bias && tid < a.fo) ? a.bias[tid] : 0.f;
  __syncthreads();

  // The wave's stream of 64-row wave-tiles.  Linear: tile w = rows [64 w, 64 (w+1)).  XCD-aware (causal conv, seg > 0):
  // workgroups go round-robin to the 8 XCDs, so XCD x = blockIdx % 8 takes the row range [x seg, (x+1) seg) of EVERY
  // time slab and walks the slabs in time order -- the shifted rows of the earlier taps were fetched by the same XCD a
  // few slabs before and are still in its L2, instead of crossing the fabric once per tap.
  int w, w_stride, w_total, wt_x = 1, n_lo = 0, n_hi = 0;
  if (a.seg > 0) {
    const int xcd = blockIdx.x & 7;
    n_lo = xcd * a.seg;
    n_hi = min(a.t_rows, n_lo + a.seg);
    wt_x = max(0, (n_hi - n_lo + NB * 16 - 1) / (NB * 16));
    w_total = (int)(a.rows / a.t_rows) * wt_x;
    w = (blockIdx.x >> 3) * 8 + wave;
    w_stride = (gridDim.x >> 3) * 8;
  } else {
    // wave-major: a problem of fewer than 8 x 256 wave-tiles spreads over all the CUs (one busy wave each) before a
    // second wave of any workgroup gets work -- the rollout's T = seq_in rows would otherwise sit on two dozen CUs
    w_total = (int)((a.rows + NB * 16 - 1) / (NB * 16));
    w = wave * gridDim.x + blockIdx.x;
    w_stride = gridDim.x * 8;
  }
  if (w >= w_total) return;
  auto locate = [&](int wt, int &base, int &nv) {       // first row and number of existing rows of wave-tile wt
    if (a.seg > 0) {
      const int slab = wt / wt_x, k = wt - slab * wt_x;
      const int n0 = n_lo + k * (NB * 16);
      base = slab * a.t_rows + n0;
      nv = min(NB * 16, n_hi - n0);
    } else {
      base = wt * (NB * 16);
      nv = (int)min<int64_t>(NB * 16, a.rows - base);
    }
  };
  auto lane_rows = [&](int base, int nv, int (&row)[NB], int (&tix)[NB]) {   // rows < 2^31 (checked by the launcher)
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      row[b] = base + min(b * 16 + r16, nv - 1);
      tix[b] = (row[b] / a.t_rows) % a.T;
    }
  };
  const unsigned my_lds = __builtin_amdgcn_readfirstlane(lds_addr(ring));
  // piece (k-step t, row rw at time index tx): two 1-KiB halves.  A row outside [0, T) of the (causal, or for
  // dil < 0 look-ahead) window is fetched from the row itself (always in bounds) and zeroed at use.
  auto live_of = [&](int t, int tx) {
    const int ts = tx - (a.taps - 1 - (32 * t) / a.F) * a.dil;
    return ts >= 0 && ts < a.T;
  };
  auto issue = [&](int t, int rw, int tx, int slot) {
    const int k0 = 32 * t;
    const int j = k0 / a.F, f0 = k0 - j * a.F;
    const int shift = (a.taps - 1 - j) * a.dil;
    const bool live = tx - shift >= 0 && tx - shift < a.T;
    const float *src = a.x + (int64_t)(rw - (live ? shift * a.t_rows : 0)) * a.F + f0 + 4 * qd;
    if (a.x2) src = k0 < a.F1 ? a.x + (int64_t)rw * a.F1 + k0 + 4 * qd : a.x2 + (int64_t)rw * (a.F - a.F1) + (k0 - a.F1) + 4 * qd;
    glds16_pair(src, src + 16, my_lds + (unsigned)slot * 2048);
  };

  int base, nv, row[NB], tix[NB];
  locate(w, base, nv);
  lane_rows(base, nv, row, tix);
#pragma unroll
  for (int q = 0; q < RING - 1; ++q) issue(0, row[q], tix[q], q);
  int cs = 0;                                     // ring slot of the piece consumed next

  for (; w < w_total; w += w_stride) {
    const int wn = w + w_stride < w_total ? w + w_stride : w;     // past the end: dummy re-fetch of this tile
    int base_n, nv_n, row_n[NB], tix_n[NB];
    locate(wn, base_n, nv_n);
    lane_rows(base_n, nv_n, row_n, tix_n);
    f32x4 acc[NB][MB];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int m = 0; m < MB; ++m) acc[b][m] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int t = 0; t < KT; ++t) {
      bf16x8 wh[MB], wl[MB];
#pragma unroll
      for (int m = 0; m < MB; ++m) {
        wh[m] = __builtin_bit_cast(bf16x8, wlds[((t * MB + m) * 2 + 0) * 64 + lane]);
        wl[m] = __builtin_bit_cast(bf16x8, wlds[((t * MB + m) * 2 + 1) * 64 + lane]);
      }
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        // refill the slot the previous trip left: piece RING-1 ahead (this tile's next k-step, or the next tile's first)
        constexpr int dummy_fb = 0;
        (void)dummy_fb;
        const int fb = (b + RING - 1) % NB, ft = t + (b + RING - 1) / NB;
        int rs = cs + (RING - 1);
        rs = rs >= RING ? rs - RING : rs;
        if (ft < KT) issue(ft, row[fb], tix[fb], rs);
        else issue(ft - KT, row_n[fb], tix_n[fb], rs);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (RING - 1)) : "memory");
        const float4 *st = reinterpret_cast<const float4 *>(ring + cs * 512) + lane;
        float4 v0 = st[0], v1 = st[64];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // slot consumed before the next trip refills it
        cs = cs + 1 == RING ? 0 : cs + 1;
        if (!live_of(t, tix[b])) v0 = v1 = make_float4(0.f, 0.f, 0.f, 0.f);
        bf16x8 dh, dl;
        split8(v0, v1, dh, dl);
#pragma unroll
        for (int m = 0; m < MB; ++m) acc[b][m] = mfma3(wh[m], wl[m], dh, dl, acc[b][m]);
      }
    }

    rowgemm_epilogue<MB, NB, STAGE>(a, acc, base, nv, tile, bias_s, lane);
    base = base_n;
    nv = nv_n;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      row[b] = row_n[b];
      tix[b] = tix_n[b];
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the dummy pieces land before the LDS is released
}

// Small problems (the rollout's T = seq_in windows: a few thousand rows).  The persistent kernel above is built for
// streams: a wave keeps RING-1 pieces in flight and pays one memory latency per RING-1 pieces, so a problem of one
// wave-tile per wave is a chain of K/32 * NB / (RING-1) latencies (14 us for 12 k rows x 192).  Here a wave owns ONE
// 16-row block and loads its whole A row (KT k-steps, 8 floats per lane each) into registers up front -- one latency --
// while the workgroup stages the weights; blocks are dealt wave-major so every CU gets one before any gets two.
// Same fragment layout, split and MFMA order as k_rowgemm_mfma: bit-identical results.
template <int MB, int KT>
__device__ __forceinline__ void rowgemm_small_body(const RowGemmArgs &a) {
  constexpr int CG = MB >= 2 ? 2 : 1;
  constexpr int LD = 16 * CG + 4;
  extern __shared__ __attribute__((aligned(16))) float smem_rg[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, qd = lane >> 4;
  constexpr int n_w = KT * MB * 2 * 64;
  uint4 *wlds = reinterpret_cast<uint4 *>(smem_rg);
  float *bias_s = smem_rg + n_w * 4;
  float *tile = bias_s + 64 + wave * (16 * LD);
  const int n_blocks = (int)((a.rows + 15) / 16);
  const int blk = wave * gridDim.x + blockIdx.x;
  const bool active = blk < n_blocks;
  const int base = blk * 16;
  const int nv = active ? (int)min<int64_t>(16, a.rows - base) : 0;
  float4 v[KT][2];
  int tx = 0;
  if (active) {
    const int rw = base + min(r16, nv - 1);
    tx = (rw / a.t_rows) % a.T;
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      const int k0 = 32 * t;
      const int j = k0 / a.F, f0 = k0 - j * a.F;
      const int shift = (a.taps - 1 - j) * a.dil;
      const bool live = tx - shift >= 0 && tx - shift < a.T;
      const float *src = a.x + (int64_t)(rw - (live ? shift * a.t_rows : 0)) * a.F + f0 + 4 * qd;
      if (a.x2) src = k0 < a.F1 ? a.x + (int64_t)rw * a.F1 + k0 + 4 * qd : a.x2 + (int64_t)rw * (a.F - a.F1) + (k0 - a.F1) + 4 * qd;
      v[t][0] = *reinterpret_cast<const float4 *>(src);
      v[t][1] = *reinterpret_cast<const float4 *>(src + 16);
    }
  }
  for (int i = tid; i < n_w; i += 512) wlds[i] = a.packed[i];
  if (tid < 64) bias_s[tid] = (a.bias && tid < a.fo) ? a.bias[tid] : 0.f;
  __syncthreads();
  if (!active) return;
  f32x4 acc[1][MB];
#pragma unroll
  for (int m = 0; m < MB; ++m) acc[0][m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < KT; ++t) {
    const int ts = tx - (a.taps - 1 - (32 * t) / a.F) * a.dil;
    float4 v0 = v[t][0], v1 = v[t][1];
    if (!(ts >= 0 && ts < a.T)) v0 = v1 = make_float4(0.f, 0.f, 0.f, 0.f);
    bf16x8 dh, dl;
    split8(v0, v1, dh, dl);
#pragma unroll
    for (int m = 0; m < MB; ++m) {
      const bf16x8 wh = __builtin_bit_cast(bf16x8, wlds[((t * MB + m) * 2 + 0) * 64 + lane]);
      const bf16x8 wl = __builtin_bit_cast(bf16x8, wlds[((t * MB + m) * 2 + 1) * 64 + lane]);
      acc[0][m] = mfma3(wh, wl, dh, dl, acc[0][m]);
    }
  }
  rowgemm_epilogue<MB, 1>(a, acc, base, nv, tile, bias_s, lane);
}

template <int MB, int KT>
__global__ __launch_bounds__(512) void k_rowgemm_small(RowGemmArgs a) {
  rowgemm_small_body<MB, KT>(a);
}

// TWO independent small problems of the same kernel shape in one launch (blockIdx.y picks the problem): the node-side and the
// link-side temporal layer of an autoregressive step, each a few microseconds on its own -- launch latency, not work, is what
// such a step pays for.  The body is instantiated once per problem (no run-time selection of the argument block).
template <int MB, int KT>
__global__ __launch_bounds__(512) void k_rowgemm_small_pair(RowGemmArgs a0, RowGemmArgs a1) {
  if (blockIdx.y == 0) rowgemm_small_body<MB, KT>(a0);
  else rowgemm_small_body<MB, KT>(a1);
}

constexpr int64_t ROWGEMM_SMALL_ROWS = 16 * 8 * 256;      // one 16-row block per wave of one workgroup per CU

template <int MB, int KT>
inline hipError_t launch_rowgemm_small_k(const RowGemmArgs &a, hipStream_t st) {
  constexpr int cg = MB >= 2 ? 2 : 1;
  const size_t lds = (size_t)KT * MB * 2 * 1024 + 256 + 8 * 16 * (16 * cg + 4) * 4;
  static unsigned long long attr_done = 0;
  if (hipError_t e = set_max_lds_once(reinterpret_cast<const void *>(&k_rowgemm_small<MB, KT>), 160 * 1024, attr_done); e != hipSuccess) return e;
  const int64_t n_blocks = (a.rows + 15) / 16;
  const int64_t grid = std::min<int64_t>(256, n_blocks);
  RowGemmArgs b = a;
  b.seg = 0;
  hipLaunchKernelGGL((k_rowgemm_small<MB, KT>), dim3((unsigned)grid), dim3(512), lds, st, b);
  return hipGetLastError();
}

template <int MB, int KT>
inline hipError_t launch_rowgemm_small_pair_k(const RowGemmArgs &a0, const RowGemmArgs &a1, hipStream_t st) {
  constexpr int cg = MB >= 2 ? 2 : 1;
  const size_t lds = (size_t)KT * MB * 2 * 1024 + 256 + 8 * 16 * (16 * cg + 4) * 4;
  static unsigned long long attr_done = 0;
  if (hipError_t e = set_max_lds_once(reinterpret_cast<const void *>(&k_rowgemm_small_pair<MB, KT>), 160 * 1024, attr_done); e != hipSuccess) return e;
  const int64_t n_blocks = std::max((a0.rows + 15) / 16, (a1.rows + 15) / 16);
  const int64_t grid = std::max<int64_t>(1, std::min<int64_t>(128, n_blocks));      // 128 workgroups per problem: the pair fills the 256 CUs
  RowGemmArgs b0 = a0, b1 = a1;
  b0.seg = b1.seg = 0;
  hipLaunchKernelGGL((k_rowgemm_small_pair<MB, KT>), dim3((unsigned)grid, 2), dim3(512), lds, st, b0, b1);
  return hipGetLastError();
}

// both problems small enough for one 16-row block per wave of 128 workgroups each (the two grids share the 256 CUs)
constexpr int64_t ROWGEMM_PAIR_ROWS = 16 * 8 * 128;

template <int MB>
inline bool launch_rowgemm_small_pair(const RowGemmArgs &a0, const RowGemmArgs &a1, hipStream_t st, hipError_t &e) {
  if (a0.rows > ROWGEMM_PAIR_ROWS || a1.rows > ROWGEMM_PAIR_ROWS) return false;
  switch (a0.taps * a0.F / 32) {
    case 2: e = launch_rowgemm_small_pair_k<MB, 2>(a0, a1, st); return true;
    case 6: e = launch_rowgemm_small_pair_k<MB, 6>(a0, a1, st); return true;
    default: return false;
  }
}

// true (and launched) when the problem is small and K / 32 is one of the instantiated depths
template <int MB>
inline bool launch_rowgemm_small(const RowGemmArgs &a, hipStream_t st, hipError_t &e) {
  if (a.rows > ROWGEMM_SMALL_ROWS) return false;
  switch (a.taps * a.F / 32) {
    case 2: e = launch_rowgemm_small_k<MB, 2>(a, st); return true;
    case 3: e = launch_rowgemm_small_k<MB, 3>(a, st); return true;
    case 4: e = launch_rowgemm_small_k<MB, 4>(a, st); return true;
    case 6: e = launch_rowgemm_small_k<MB, 6>(a, st); return true;
    case 9: e = launch_rowgemm_small_k<MB, 9>(a, st); return true;
    case 12: e = launch_rowgemm_small_k<MB, 12>(a, st); return true;
    default: return false;
  }
}

inline int64_t rowgemm_lds_bytes(int K, int MB, int ring, bool stage = true) {
  const int cg = MB >= 2 ? 2 : 1;
  return (int64_t)(K / 32) * MB * 2 * 1024 + 256 + 8 * ring * 2048 + (stage ? 8 * 16 * (16 * cg + 4) * 4 : 0);
}
// deepest ring (5, 3, 3 without the output tile = -3, else 2) that fits the LDS next to the weights; 0 = the weights alone are too large
inline int rowgemm_ring(int K, int MB) {
  if (rowgemm_lds_bytes(K, MB, 5) <= 160 * 1024) return 5;
  if (rowgemm_lds_bytes(K, MB, 3) <= 160 * 1024) return 3;
  if (rowgemm_lds_bytes(K, MB, 3, false) <= 160 * 1024) return -3;
  if (rowgemm_lds_bytes(K, MB, 2) <= 160 * 1024) return 2;
  return 0;
}

template <int MB, int NB, int RING, bool STAGE = true>
inline hipError_t launch_rowgemm_r(const RowGemmArgs &a, hipStream_t st) {
  static unsigned long long attr_done = 0;
  if (hipError_t e = set_max_lds_once(reinterpret_cast<const void *>(&k_rowgemm_mfma<MB, NB, RING, STAGE>), 160 * 1024, attr_done); e != hipSuccess) return e;
  constexpr int64_t WT = NB * 16;                          // rows of a wave-tile
  const int64_t lds = rowgemm_lds_bytes(a.taps * a.F, MB, RING, STAGE);
  RowGemmArgs b = a;
  b.seg = 0;
  int64_t grid = std::min<int64_t>(256, (a.rows + WT - 1) / WT);                // one persistent workgroup per CU
  if (a.taps > 1 && a.t_rows >= 64 * WT) {                 // a slab wide enough to give every XCD several wave-tiles
    b.seg = (int)(((a.t_rows + 7) / 8 + 15) / 16 * 16);
    grid = 256;
  }
  hipLaunchKernelGGL((k_rowgemm_mfma<MB, NB, RING, STAGE>), dim3((unsigned)grid), dim3(512), (size_t)lds, st, b);
  return hipGetLastError();
}

template <int MB, int NB>
inline hipError_t launch_rowgemm_t(const RowGemmArgs &a, hipStream_t st) {
  const int ring = rowgemm_ring(a.taps * a.F, MB);
  if (ring == 5) return launch_rowgemm_r<MB, NB, 5>(a, st);
  if (ring == 3) return launch_rowgemm_r<MB, NB, 3>(a, st);
  if (ring == -3) return launch_rowgemm_r<MB, NB, 3, false>(a, st);
  return launch_rowgemm_r<MB, NB, 2>(a, st);      // e.g. the 3 x 128 -> 64 Conv1D of a d = 128 emulator: 96 KB of weights
}

inline int rowgemm_mb(int fo) { return fo <= 16 ? 1 : (fo <= 32 ? 2 : 4); }

inline hipError_t launch_rowgemm(const RowGemmArgs &a, hipStream_t st) {
  hipError_t e = hipSuccess;
  switch (rowgemm_mb(a.fo)) {
    case 1: if (launch_rowgemm_small<1>(a, st, e)) return e; break;
    case 2: if (launch_rowgemm_small<2>(a, st, e)) return e; break;
    default: if (launch_rowgemm_small<4>(a, st, e)) return e; break;
  }
  switch (rowgemm_mb(a.fo)) {
    case 1: return launch_rowgemm_t<1, 4>(a, st);
    case 2: return launch_rowgemm_t<2, 4>(a, st);
    default: return launch_rowgemm_t<4, 4>(a, st);
  }
}

}  // namespace uds
