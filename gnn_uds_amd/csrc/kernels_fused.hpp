// Fused spatial-layer kernel (gfx950): one workgroup (8 waves, one per CU) = one tile of one side of the
// network, looping over a chunk of snapshots.  Replaces, per side, the whole chain
//   Dense(d/2,relu) on the secondary rows -> NodeEdge aggregation -> concat -> GAT linear ->
//   attention logits / segmented softmax / neighbour sum -> bias -> activation
// (emulator.py:225-230) with ONE pass over HBM: inputs are read, the only global writes are the
// layer outputs; x_e / e_x / agg / hx / attention scalars live in LDS or registers.
//
// Per snapshot, per tile:
//   P1  secondary MLP   sec[q]  = relu(in_sec[q] @ Wsmall + b)                      stage -> MFMA -> LDS
//   P2  primary linear  hx[p]   = [in_prim[p] | sum_q w_pq sec[q]] @ Wbig ; s_self, s_nbr -> LDS
//   P3  GAT aggregate   out[i]  = act(sum_j softmax_j(leaky(s_self_i+s_nbr_j)) hx[j] + bias)  LDS -> HBM
//
// HBM -> LDS: every input row of snapshot s+1 is fetched by LDS-DMA (global_load_lds_dwordx4, no VGPRs)
// while snapshot s is being computed, into a staging image laid out in MFMA-fragment order: the lane that
// issues a 16-B piece is the lane that later reads it back (ds_read_b128, conflict-free, no cross-wave
// hand-off: the issuing wave's own counted vmcnt is the only synchronisation the stage needs).
//
// GEMMs: v_mfma_f32_16x16x32_bf16 with the WEIGHTS as the A operand and 16 data rows as the B operand
// (result lane layout = 4 consecutive features of one row -> 16-B LDS / HBM accesses).  fp32 operands
// are split into bf16 hi + lo and three products are accumulated in fp32 (hi*hi + lo*hi + hi*lo,
// error ~2^-16 relative per product: MORE mantissa than the TF32 path TensorFlow uses by default on
// the reference's GPUs, cheaper than the 157 TF fp32 MFMA that would cap the layer below the HBM
// roofline -- SURVEY.md section 7).  Weight fragments (32 + 96 VGPRs per lane at F=64) are loaded once
// per workgroup and stay in registers across the snapshot loop.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "kernels_dense.hpp"
#include "kernels_sparse.hpp"
#include "tile_plan.hpp"

namespace uds {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int FUSED_H = 32;             // d/2
constexpr int FUSED_D = 64;             // d
constexpr int FUSED_WAVES = 8;          // 512 threads: one workgroup per CU owns the whole 160 KiB LDS
constexpr int FUSED_U = 4;              // P3: row groups (of 4 rows) a wave keeps in flight (the trip code assumes 4)
constexpr int SEC_STRIDE = FUSED_H + 4; // floats; 144-B rows make the 16-B fragment writes conflict-free

struct FusedSide {
  const float *prim_in, *sec_in;
  // 96-wide rows may come in two pieces, [64 floats | 32 floats] from two tensors (the reference concatenates the
  // boundary embedding to x before block 2, emulator.py:260): *_in then holds the first 64 columns (row stride 64) and
  // *_in2 the last 32 (row stride 32); nullptr = one tensor with row stride 96.
  const float *prim_in2, *sec_in2;
  float *out;
  const uint4 *w_small, *w_big;   // packed bf16 hi/lo fragments (k_pack_weight_frags)
  const uint4 *w_big32, *w_small32;      // both kernels as 32x32x16 A-operand fragments (k_pack_weight_frags32): k_fused_ws only, else NULL
  const float *b_small, *a_self, *a_nbr, *b_out, *ne_val;
  int n_prim_glob, n_sec_glob;
  // k_fused_cs only, else NULL: (S, n_prim_glob, h) added to the NodeEdge aggregate of the primary rows -- the dense part of a
  // trained NodeEdge, (bias off the incidence support) @ Dense(secondary rows), computed by uds_remainder_forward (emulator.py:36-45)
  const float *rem;
};

struct FusedArgs {
  FusedSide side[2];
  const int32_t *hdr, *pool;
  const int32_t *blocks;     // k_fused_tile: per tile of the list one block [header 8 | fixed-width index lists] of meta_cap ints (tile_plan.hpp)
  int n_tiles, S, chunk, p_cap, q_cap, meta_cap, act, side_mask;
  unsigned long long *dbg;   // diagnostic builds only (UDS_PHASE_TIMING): 8 cycle sums per wave
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

// The same for v_mfma_f32_32x32x16_bf16 (lane l: r = l & 31, hf = l >> 5 holds A[row r][k = 8 hf + j]): W (K x F_out, K % 16 == 0,
// F_out % 32 == 0) -> out[((t*MB + m)*2 + hl)*64 + lane] = 8 bf16 {W[16 t + 8 hf + jj][32 m + r]}, MB = F_out / 32.
__global__ void k_pack_weight_frags32(const float *__restrict__ W, int K, int F_out, uint4 *__restrict__ out) {
  const int MB = F_out / 32, KT = K / 16;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= KT * MB * 64) return;
  const int lane = idx & 63, m = (idx >> 6) % MB, t = (idx >> 6) / MB;
  const int hf = lane >> 5, f = 32 * m + (lane & 31);
  bf16x8 hi, lo;
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) {
    const float w = W[(int64_t)(16 * t + 8 * hf + jj) * F_out + f];
    const __bf16 h = (__bf16)w;
    hi[jj] = h;
    lo[jj] = (__bf16)(w - (float)h);
  }
  out[((t * MB + m) * 2 + 0) * 64 + lane] = __builtin_bit_cast(uint4, hi);
  out[((t * MB + m) * 2 + 1) * 64 + lane] = __builtin_bit_cast(uint4, lo);
}

// hi = bf16(v) (round to nearest even), lo = bf16(v - hi) for 8 floats, written pair-wise so that every pair takes the
// five-instruction form: v_cvt_pk_bf16_f32 (hi pair), shift / mask (the two hi values back as floats), v_pk_add_f32 with
// negated operand (both differences), v_cvt_pk_bf16_f32 (lo pair).  The element-wise formulation gives the same values
// but the compiler converts one pair of every eight separately (8 instructions for it).
__device__ __forceinline__ void split8(const float4 &a, const float4 &b, bf16x8 &hi, bf16x8 &lo) {
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  const f32x2_t v[4] = {{a.x, a.y}, {a.z, a.w}, {b.x, b.y}, {b.z, b.w}};
  unsigned hw[4], lw[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const bf16x2_t h = __builtin_convertvector(v[p], bf16x2_t);
    const unsigned hb = __builtin_bit_cast(unsigned, h);
    const f32x2_t hf = {__uint_as_float(hb << 16), __uint_as_float(hb & 0xffff0000u)};
    const bf16x2_t l = __builtin_convertvector(v[p] - hf, bf16x2_t);
    hw[p] = hb;
    lw[p] = __builtin_bit_cast(unsigned, l);
  }
  hi = __builtin_bit_cast(bf16x8, make_uint4(hw[0], hw[1], hw[2], hw[3]));
  lo = __builtin_bit_cast(bf16x8, make_uint4(lw[0], lw[1], lw[2], lw[3]));
}

__device__ __forceinline__ f32x4 mfma3(const bf16x8 &wh, const bf16x8 &wl, const bf16x8 &dh, const bf16x8 &dl, f32x4 acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, dl, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, dh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, dh, acc, 0, 0, 0);
  return acc;
}

// 16-lane all-reduce (max / sum) with DPP row operations: no LDS traffic, VALU latency only.  quad_perm / row_ror read
// a valid lane everywhere, so bound_ctrl changes nothing -- but with it (and full masks) the backend folds the move into
// the max / add that consumes it (v_max_f32_dpp): 4 instructions per reduction instead of 8.
template <int CTRL>
__device__ __forceinline__ float row_dpp(float v) {
  const int x = __builtin_bit_cast(int, v);
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(x, x, CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, row_dpp<0xB1>(v));    // quad_perm [1,0,3,2]
  v = fmaxf(v, row_dpp<0x4E>(v));    // quad_perm [2,3,0,1]
  v = fmaxf(v, row_dpp<0x124>(v));   // row_ror:4
  v = fmaxf(v, row_dpp<0x128>(v));   // row_ror:8
  return v;
}
__device__ __forceinline__ float row16_sum(float v) {
  v += row_dpp<0xB1>(v);
  v += row_dpp<0x4E>(v);
  v += row_dpp<0x124>(v);
  v += row_dpp<0x128>(v);
  return v;
}

// The same all-reduces for FOUR values at once with the DPP move folded into the max / add (v_max_f32_dpp: hipcc keeps a
// separate v_mov_b32_dpp per step, 8 instructions per reduction instead of 4).  Written as one asm block because the
// "VALU write -> DPP read of the same VGPR" hazard (2 wait states) is invisible to the compiler inside inline asm: the
// four values are interleaved so that dependent steps are 3 instructions apart, and one s_nop covers the producer of
// the inputs.  In-place is safe: quad_perm and row_ror read inside the 16-lane row that the same pass writes.
#define UDS_DPP4(op, ctrl)                                                      \
  op " %0, %0, %0 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"       \
  op " %1, %1, %1 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"       \
  op " %2, %2, %2 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"       \
  op " %3, %3, %3 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
// first step: out = op(dpp(in), in) into fresh registers (the inputs stay live, no copies), the rest in place
#define UDS_DPP4_FIRST(op, ctrl)                                                \
  op " %0, %4, %4 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"       \
  op " %1, %5, %5 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"       \
  op " %2, %6, %6 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"       \
  op " %3, %7, %7 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
__device__ __forceinline__ void row16_max4(const float (&in)[4], float (&out)[4]) {
  asm("s_nop 1\n\t" UDS_DPP4_FIRST("v_max_f32_dpp", "quad_perm:[1,0,3,2]") UDS_DPP4("v_max_f32_dpp", "quad_perm:[2,3,0,1]")
      UDS_DPP4("v_max_f32_dpp", "row_ror:4") UDS_DPP4("v_max_f32_dpp", "row_ror:8")
      : "=&v"(out[0]), "=&v"(out[1]), "=&v"(out[2]), "=&v"(out[3])
      : "v"(in[0]), "v"(in[1]), "v"(in[2]), "v"(in[3]));
}
__device__ __forceinline__ void row16_sum4(const float (&in)[4], float (&out)[4]) {
  asm("s_nop 1\n\t" UDS_DPP4_FIRST("v_add_f32_dpp", "quad_perm:[1,0,3,2]") UDS_DPP4("v_add_f32_dpp", "quad_perm:[2,3,0,1]")
      UDS_DPP4("v_add_f32_dpp", "row_ror:4") UDS_DPP4("v_add_f32_dpp", "row_ror:8")
      : "=&v"(out[0]), "=&v"(out[1]), "=&v"(out[2]), "=&v"(out[3])
      : "v"(in[0]), "v"(in[1]), "v"(in[2]), "v"(in[3]));
}

// sum over the four lanes l, l^16, l^32, l^48 with v_permlane16_swap / v_permlane32_swap (VALU, no LDS round trip):
// swap16(A=B=v) gives A' = [r0,r0,r2,r2], B' = [r1,r1,r3,r3] (16-lane rows r0..r3), swap32 gives [lo,lo] and [hi,hi].
__device__ __forceinline__ float quarters_sum(float v) {
  const unsigned u = __float_as_uint(v);
  const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const float s = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  const unsigned w = __float_as_uint(s);
  const auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// ACT: UDS_ACT_* known at compile time (relu / linear fast paths), or -1 = decide at run time from a.act.
template <int ACT>
__device__ __forceinline__ float fused_act(float v, int act_rt) {
  // relu as a signed-integer max on the bit pattern: negative floats (sign bit set) are negative integers, so max(bits, 0)
  // is +0 for them and the value itself otherwise -- ONE instruction (v_max_i32).  fmaxf(v, 0) costs two: the IEEE rules
  // make the compiler canonicalise the MFMA result first (v_max_f32 v, v, v), in every epilogue.  Same result for every
  // finite input and +-inf; a NaN stays a NaN (as tf.nn.relu) where fmaxf would return 0.
  if constexpr (ACT == UDS_ACT_RELU) return __int_as_float(max(__float_as_int(v), 0));
  else if constexpr (ACT == UDS_ACT_LINEAR) return v;
  else return apply_act(v, act_rt);
}

// workgroup barrier that does NOT drain the vector-memory counter: LDS-DMA prefetches stay in flight across it
// (a __syncthreads() would wait vmcnt(0) while an LDS-DMA is outstanding).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// wait until all but the n youngest vector-memory operations of this wave are done (n = the output stores it
// issued last: CDNA4 counts stores in vmcnt too, and the DMA pieces are older than they are)
// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>)
template <int N, int I = 0, class F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<N, I + 1>(f);
  }
}

#ifdef UDS_SMALL_IN_LDS
constexpr bool SMALL_IN_LDS = true;     // the fusion-MLP weights (32 VGPRs) live in LDS, read per 16-row block
#else
constexpr bool SMALL_IN_LDS = false;
#endif

#ifndef UDS_P3_SLOTS
#define UDS_P3_SLOTS 2
#endif
constexpr int P3_SLOTS = UDS_P3_SLOTS;      // neighbour slots per P3 step (reads in flight per row group)

// value of lane K of this lane's 16-lane row (v_mov_b32_dpp row_newbcast:K), K compile-time
template <int K>
__device__ __forceinline__ int row16_bcast(int v) {
  static_assert(K >= 0 && K < 16, "lane inside a 16-lane row");
  return __builtin_amdgcn_update_dpp(v, v, 0x150 + K, 0xf, 0xf, true);      // every lane is written: no `old` value to set up
}

__device__ __forceinline__ void wait_all_but(int n) {
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
  }
}
// Counted wait with a wave-uniform run-time count: all but the n youngest vector-memory operations of this wave are done.
// gfx950 has no register form of s_waitcnt, so the count is rounded DOWN to an even number <= 24 (waiting for more than
// asked is always safe) and picked by a four-level tree of scalar compares.
__device__ __forceinline__ void wait_vm(int n) {
#define UDS_VMC(k) asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory")
  if (n >= 16) {
    if (n >= 20) { if (n >= 24) UDS_VMC(24); else if (n >= 22) UDS_VMC(22); else UDS_VMC(20); }
    else { if (n >= 18) UDS_VMC(18); else UDS_VMC(16); }
  } else if (n >= 8) {
    if (n >= 12) { if (n >= 14) UDS_VMC(14); else UDS_VMC(12); }
    else { if (n >= 10) UDS_VMC(10); else UDS_VMC(8); }
  } else {
    if (n >= 4) { if (n >= 6) UDS_VMC(6); else UDS_VMC(4); }
    else { if (n >= 2) UDS_VMC(2); else UDS_VMC(0); }
  }
#undef UDS_VMC
}

// One 16-B-per-lane LDS-DMA piece (1 KiB per wave): global `src` (per lane) -> LDS `lds_byte` + lane * 16 (wave-
// uniform base in M0).  Written as inline asm on purpose: hipcc orders every later LDS read behind a DMA it can see
// (it inserted `s_waitcnt vmcnt(0)` in front of the first ds_read after each __builtin_amdgcn_global_load_lds, which
// serialised the whole prefetch); an asm DMA is invisible to that bookkeeping, so completion is tracked by hand with
// wait_all_but() -- the loop below issues no compiler-visible vector-memory LOADS, only stores.  M0 is saved and
// restored inside the statement (the compiler owns it).
__device__ __forceinline__ void glds16(const float *src, unsigned lds_byte) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(src), "s"(lds_byte)
               : "memory");
}
// NP consecutive 1-KiB pieces in ONE statement (one M0 save / restore): piece i goes to lds_byte + i * 1024.
template <int NP>
__device__ __forceinline__ void glds16_run(const float *const (&src)[NP], unsigned lds_byte) {
  static_assert(NP == 4 || NP == 6, "pieces per 16-row block: 4 (64 floats) or 6 (96 floats)");
  unsigned keep;
  if constexpr (NP == 4) {
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, off\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src[0]), "v"(src[1]), "v"(src[2]), "v"(src[3]), "s"(lds_byte)
                 : "memory", "scc");
  } else {
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, off\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, off\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %5, off\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %6, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src[0]), "v"(src[1]), "v"(src[2]), "v"(src[3]), "v"(src[4]), "v"(src[5]), "s"(lds_byte)
                 : "memory", "scc");
  }
}
// The same run with ONE 32-bit per-lane byte offset and a wave-uniform 64-bit base in SGPRs (saddr form): piece i reads
// base + voff + 64 i.  The instruction offset is added to the memory address AND to the LDS address, so M0 advances by
// 1 KiB - 64 B per piece to land piece i at lds_byte + i KiB.  No 64-bit vector address arithmetic per piece and
// snapshot (it was ~8 VALU instructions per 16-row block): the snapshot only moves the scalar base.
template <int NP>
__device__ __forceinline__ void glds16_run_s(const float *base, unsigned voff, unsigned lds_byte) {
  static_assert(NP == 2 || NP == 4 || NP == 6, "pieces per run");
  unsigned keep;
  if constexpr (NP == 2) {
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                 "s_add_u32 m0, m0, 0x3c0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:64\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(base), "s"(lds_byte)
                 : "memory", "scc");
  } else if constexpr (NP == 4) {
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                 "s_add_u32 m0, m0, 0x3c0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:64\n\t"
                 "s_add_u32 m0, m0, 0x3c0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:128\n\t"
                 "s_add_u32 m0, m0, 0x3c0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:192\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(base), "s"(lds_byte)
                 : "memory", "scc");
  } else {
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                 "s_add_u32 m0, m0, 0x3c0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:64\n\t"
                 "s_add_u32 m0, m0, 0x3c0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:128\n\t"
                 "s_add_u32 m0, m0, 0x3c0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:192\n\t"
                 "s_add_u32 m0, m0, 0x3c0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:256\n\t"
                 "s_add_u32 m0, m0, 0x3c0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:320\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(base), "s"(lds_byte)
                 : "memory", "scc");
  }
}
__device__ __forceinline__ unsigned lds_addr(const void *p) {
  return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void *)p;
}

template <int FP, int FS, int ACT>
__global__ __launch_bounds__(FUSED_WAVES * 64, 2) void k_fused_tile(FusedArgs a) {
  constexpr int NW = FUSED_WAVES, NT = FUSED_WAVES * 64;
  constexpr int KT_S = FS / 32, MB_S = FUSED_H / 16;                    // small GEMM: FS -> 32
  constexpr int KT_X = FP / 32, KT_B = KT_X + 1, MB_B = FUSED_D / 16;   // big GEMM: FP + 32 -> 64
  constexpr int U = FUSED_U;
  extern __shared__ __attribute__((aligned(16))) int32_t smem[];
#ifdef UDS_PHASE_TIMING
  unsigned long long tm_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tl_ = clock64();
  const unsigned long long rt0_ = __builtin_amdgcn_s_memrealtime(), mt0_ = tl_;      // 100 MHz wall clock beside the shader-cycle counter: the clock the chip holds
#define UDS_STAMP(k) do { const unsigned long long n_ = clock64(); tm_[k] += n_ - tl_; tl_ = n_; } while (0)
#else
#define UDS_STAMP(k) do { } while (0)
#endif

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, qd = lane >> 4;
  const int c16 = r16, rs = qd;                  // P3 mapping: 16 lanes x float4 per output row, 4 rows per group
#ifdef UDS_PRIO_YOUNG
  if (wave >= 4) __builtin_amdgcn_s_setprio(1);  // experiment: static priority for the second-dispatched half (MI355X_MICROARCH.md item 4)
#endif

  // XCD-aware bijective remap: workgroups b, b+8, b+16.. share an XCD (round-robin dispatch), give each XCD
  // a contiguous range of work items so neighbouring tiles of one snapshot chunk meet in one L2.
  const int W = gridDim.x, b = blockIdx.x;
  const int q8 = W / 8, r8 = W % 8, xcd = b % 8;
  const int w = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + b / 8;
  const int tile = w % a.n_tiles;
  const int chunk_id = w / a.n_tiles;
  const int s_begin = chunk_id * a.chunk;
  const int s_end = min(a.S, (chunk_id + 1) * a.chunk);
  // Set-up is two dependent round trips: (1) everything that needs no metadata -- the side's weight fragments, attention
  // vectors, biases -- together with the tile's block (header + index lists at a fixed stride, so nothing has to be known
  // before the fetch); (2) the first snapshot's rows and the NodeEdge values the lists point at.  The side comes from the
  // (scalar-cached) header array.
  const int sd = __builtin_amdgcn_readfirstlane(a.hdr[tile * TILE_HDR_INTS + 6]);
  if (!((a.side_mask >> sd) & 1)) return;
  const FusedSide &S_ = a.side[sd];
  constexpr bool COLD_X = FP > 64, COLD_S = FS > 64 || SMALL_IN_LDS;
  constexpr int T_COLD_X = KT_X - 1, T_COLD_S = SMALL_IN_LDS ? 0 : KT_S - 1;     // first k-step that lives in LDS
  constexpr int N_COLD_S = (KT_S - T_COLD_S) * MB_S * 2 * 64;
  float *s_self = reinterpret_cast<float *>(smem + a.meta_cap);
  float *s_nbr = s_self + a.p_cap;
  float *attn = s_nbr + a.p_cap;                 // a_self[64] | a_nbr[64] | b_small[32]
  // 96-wide inputs: the weight fragments of the third 32-wide k-step stay in LDS ("cold": read per 16-row block) so
  // that the register-resident set is the same as for 64-wide inputs
  uint4 *cold_b = reinterpret_cast<uint4 *>(attn + 2 * FUSED_D + FUSED_H);     // MB_B x 2 fragments x 64 lanes
  uint4 *cold_s = cold_b + (COLD_X ? MB_B * 2 * 64 : 0);                        // (KT_S - T_COLD_S) x MB_S x 2 fragments x 64 lanes
  // weight fragments straight from global memory into registers (every wave reads the same 22-45 KB: L2 / L1 hits after the first)
  bf16x8 wsh[KT_S][MB_S], wsl[KT_S][MB_S], wbh[KT_B][MB_B], wbl[KT_B][MB_B];
#pragma unroll
  for (int t = 0; t < KT_S; ++t)
#pragma unroll
    for (int m = 0; m < MB_S; ++m) {
      if (COLD_S && t >= T_COLD_S) continue;
      wsh[t][m] = __builtin_bit_cast(bf16x8, S_.w_small[((t * MB_S + m) * 2 + 0) * 64 + lane]);
      wsl[t][m] = __builtin_bit_cast(bf16x8, S_.w_small[((t * MB_S + m) * 2 + 1) * 64 + lane]);
    }
#pragma unroll
  for (int t = 0; t < KT_B; ++t)
#pragma unroll
    for (int m = 0; m < MB_B; ++m) {
      if (COLD_X && t == T_COLD_X) continue;
      wbh[t][m] = __builtin_bit_cast(bf16x8, S_.w_big[((t * MB_B + m) * 2 + 0) * 64 + lane]);
      wbl[t][m] = __builtin_bit_cast(bf16x8, S_.w_big[((t * MB_B + m) * 2 + 1) * 64 + lane]);
    }
  if (COLD_X)
    for (int i = tid; i < MB_B * 2 * 64; i += NT) cold_b[i] = S_.w_big[T_COLD_X * MB_B * 2 * 64 + i];
  if (COLD_S)
    for (int i = tid; i < N_COLD_S; i += NT) cold_s[i] = S_.w_small[T_COLD_S * MB_S * 2 * 64 + i];
  f32x4 bo = f32x4{0.f, 0.f, 0.f, 0.f};
  if (S_.b_out) bo = *reinterpret_cast<const f32x4 *>(S_.b_out + 4 * c16);
  {
    const uint4 *src = reinterpret_cast<const uint4 *>(a.blocks + (int64_t)tile * a.meta_cap);
    uint4 *dst = reinterpret_cast<uint4 *>(smem);
    for (int i = tid; i < a.meta_cap / 4; i += NT) dst[i] = src[i];
  }
  if (tid < FUSED_D) {   // attention vectors and the small GEMM's bias live in LDS (read once per 16-row block)
    // scaled by log2(e): the scores s_self, s_nbr then are logits in base-2 units (leaky_relu commutes with a positive
    // scale, softmax = exp2 of base-2 logits), and P3 needs no multiply in front of its exp2
    attn[tid] = S_.a_self[tid] * 1.44269504088896340736f;
    attn[FUSED_D + tid] = S_.a_nbr[tid] * 1.44269504088896340736f;
    if (tid < FUSED_H) attn[2 * FUSED_D + tid] = S_.b_small ? S_.b_small[tid] : 0.f;
  }
  __syncthreads();
  UDS_STAMP(7);   // tile block (header + index lists) fetched
  const int n_own = __builtin_amdgcn_readfirstlane(smem[0]), n_prim = __builtin_amdgcn_readfirstlane(smem[1]),
            n_sec = __builtin_amdgcn_readfirstlane(smem[2]), flags = __builtin_amdgcn_readfirstlane(smem[3]),
            n_ovf = __builtin_amdgcn_readfirstlane(smem[4]), n_adj = __builtin_amdgcn_readfirstlane(smem[5]),
            inc_width = __builtin_amdgcn_readfirstlane(smem[7]);
  const EllOffsets off = ell_offsets(n_own, n_prim, n_sec, flags, n_ovf, n_adj);      // section offsets inside the block (tile_plan.hpp)

  float *sec = reinterpret_cast<float *>(cold_s + (COLD_S ? N_COLD_S : 0));
  float *hx = sec + a.q_cap * SEC_STRIDE;
  float *stage_s = hx + a.p_cap * FUSED_D;       // (q_cap/16) blocks x KT_S x 2 pieces x 1 KiB, fragment order
  float *stage_p = stage_s + a.q_cap * FS;       // (p_cap/16) blocks x KT_X x 2 pieces x 1 KiB

  const int32_t *prim_ids = smem + off.prim;
  const int32_t *sec_ids = smem + off.sec;
  const uint32_t *inc_loc = reinterpret_cast<const uint32_t *>(smem + off.inc_loc);      // 4 x u8 local secondary rows per primary row
  int32_t *inc_w = smem + off.inc_w;                                                      // 4 value positions per primary row -> 4 values
  const f32x4 *inc_val4 = reinterpret_cast<const f32x4 *>(inc_w);
  const unsigned char *adj_b = reinterpret_cast<const unsigned char *>(smem + off.adj);   // 16 x u8 neighbours per own row (0xFF = none)
  const int32_t *ovf_ptr = smem + off.ovf_ptr, *ovf_loc = smem + off.ovf_loc;            // flags & ELL_FLAG_INC_OVF
  int32_t *ovf_w = smem + off.ovf_w;
  const int32_t *adj_ptr = smem + off.adj_ptr, *adj_loc = smem + off.adj_loc;            // flags & ELL_FLAG_LONG_ROWS
  // LDS-DMA of one 16-row block in fragment order: piece (t, i) of lane (r16, qd) = floats 32t + 16i + 4qd .. +3
  // of row r16 of the block.  The destination is wave-uniform (base + lane * 16 B is implicit).  Row offsets do not
  // change from snapshot to snapshot, so a wave keeps those of its first two P1 blocks and first P2 block in registers;
  // the snapshot only moves a scalar base (sec_next / prim_next: the rows of the NEXT snapshot to fetch).
  auto sec_row = [&](int blk) { return sec_ids[min(blk * 16 + r16, n_sec - 1)]; };      // row * width < 2^31 floats
  auto prim_row = [&](int blk) { return prim_ids[min(blk * 16 + r16, n_prim - 1)]; };
  const int srow0 = wave * 16 < n_sec ? sec_row(wave) : 0, srow1 = (wave + NW) * 16 < n_sec ? sec_row(wave + NW) : 0;
  const int prow0 = wave * 16 < n_prim ? prim_row(wave) : 0;
  const bool sec_split = FS == 96 && S_.sec_in2 != nullptr, prim_split = FP == 96 && S_.prim_in2 != nullptr;
  const int64_t sec_stride = (int64_t)S_.n_sec_glob * (sec_split ? 64 : FS), sec_stride2 = (int64_t)S_.n_sec_glob * 32;       // floats per snapshot
  const int64_t prim_stride = (int64_t)S_.n_prim_glob * (prim_split ? 64 : FP), prim_stride2 = (int64_t)S_.n_prim_glob * 32;
  const float *sec_next = S_.sec_in + s_begin * sec_stride, *sec_next2 = sec_split ? S_.sec_in2 + s_begin * sec_stride2 : nullptr;
  const float *prim_next = S_.prim_in + s_begin * prim_stride, *prim_next2 = prim_split ? S_.prim_in2 + s_begin * prim_stride2 : nullptr;
  auto dma_sec = [&](int blk) {      // secondary block blk of the snapshot sec_next points at
    const unsigned row = (unsigned)(blk == wave ? srow0 : (blk == wave + NW ? srow1 : sec_row(blk)));
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_addr(stage_s) + (unsigned)blk * (KT_S * 2 * 1024));
    if (sec_split) {      // 96 floats from two tensors: 64 (four pieces) + 32 (two pieces)
      glds16_run_s<4>(sec_next, row * 256u + 16u * qd, dst);
      glds16_run_s<2>(sec_next2, row * 128u + 16u * qd, dst + 4096u);
    } else {              // piece (t, i) = floats 32t + 16i = 16 * (2t + i) of the row
      glds16_run_s<2 * KT_S>(sec_next, row * (unsigned)(FS * 4) + 16u * qd, dst);
    }
  };
  auto dma_prim = [&](int blk) {
    const unsigned row = (unsigned)(blk == wave ? prow0 : prim_row(blk));
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_addr(stage_p) + (unsigned)blk * (KT_X * 2 * 1024));
    if (prim_split) {
      glds16_run_s<4>(prim_next, row * 256u + 16u * qd, dst);
      glds16_run_s<2>(prim_next2, row * 128u + 16u * qd, dst + 4096u);
    } else {
      glds16_run_s<2 * KT_X>(prim_next, row * (unsigned)(FP * 4) + 16u * qd, dst);
    }
  };
  auto sec_advance = [&]() { sec_next += sec_stride; if (sec_split) sec_next2 += sec_stride2; };
  auto prim_advance = [&]() { prim_next += prim_stride; if (prim_split) prim_next2 += prim_stride2; };

  // P3 (the planner keeps n_own <= 4*NW*U, so one trip covers the tile: row = wave*4 + 4*NW*u + rs): the (wave-
  // uniform) largest degree of each of this wave's row groups never changes, keep it in SGPRs.  Degrees come from the
  // fixed-width lists (valid bytes of a row), for tiles with long rows from the full lists.
  int p3_dmax[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int i = wave * 4 + 4 * NW * u + rs;
    const int ic = min(i, n_own - 1);
    int dmx;
    if (flags & ELL_FLAG_LONG_ROWS) {
      dmx = i < n_own ? adj_ptr[ic + 1] - adj_ptr[ic] : 0;
    } else {
      const unsigned long long m = __ballot(i < n_own && adj_b[ic * ELL_ADJ + c16] != 0xFF);
      dmx = __builtin_popcountll((m >> (16 * rs)) & 0xffffull);
    }
    dmx = max(dmx, __shfl_xor(dmx, 16));
    dmx = max(dmx, __shfl_xor(dmx, 32));
    p3_dmax[u] = __builtin_amdgcn_readfirstlane(dmx);
  }

  // Counted waits: vmcnt retires vector-memory operations in issue order (DMA pieces and output stores alike), so a
  // phase waits for "all but the operations this wave issued after the batch it needs".  Two running counts (wave-
  // uniform): operations issued since the last secondary-row batch / since the last primary-row batch.
  const int np_sec = sec_split ? 6 : 2 * KT_S, np_prim = prim_split ? 6 : 2 * KT_X;      // DMA pieces per 16-row block
  int vm_since_sec = 0, vm_since_prim = 0;
  if (s_begin < s_end) {
    for (int blk = wave; blk * 16 < n_sec; blk += NW) dma_sec(blk);
    for (int blk = wave; blk * 16 < n_prim; blk += NW) {
      dma_prim(blk);
      vm_since_sec += np_prim;
    }
    sec_advance();
    prim_advance();
  }
  // everything below overlaps with the first snapshot's DMA
#ifdef UDS_PHASE_TIMING
  const unsigned long long t_dma_issued_ = clock64();
#endif
  for (int i = tid; i < ELL_INC * n_prim; i += NT) {      // NodeEdge values of the fixed-width lists (0 for padding)
    const int k = inc_w[i];
    reinterpret_cast<float *>(inc_w)[i] = k >= 0 ? S_.ne_val[k] : 0.f;
  }
  if (flags & ELL_FLAG_INC_OVF)
    for (int i = tid; i < n_ovf; i += NT) reinterpret_cast<float *>(ovf_w)[i] = S_.ne_val[ovf_w[i]];
  const float *ovf_val = reinterpret_cast<const float *>(ovf_w);
#ifdef UDS_PHASE_TIMING
  const unsigned long long t_loads_issued_ = clock64();
#endif
  __syncthreads();   // NodeEdge values, attention vectors and bias are in LDS
  UDS_STAMP(0);   // setup: metadata, first DMA issue, weights

  // ---------------- P1: secondary MLP -> LDS ----------------
  auto phase1 = [&](int s) {
    if (wave * 16 < n_sec) wait_vm(vm_since_sec);      // this wave's secondary-row slots for snapshot s have landed
    UDS_STAMP(1);
    const bool more = s + 1 < s_end;
    for (int blk = wave; blk * 16 < n_sec; blk += NW) {
      const float4 *st = reinterpret_cast<const float4 *>(stage_s + blk * (KT_S * 2 * 256)) + lane;
      bf16x8 dh[KT_S], dl[KT_S];
#pragma unroll
      for (int t = 0; t < KT_S; ++t) split8(st[(2 * t) * 64], st[(2 * t + 1) * 64], dh[t], dl[t]);
      const int lrow = blk * 16 + r16;
      f32x4 acc[MB_S];
#pragma unroll
      for (int m = 0; m < MB_S; ++m) acc[m] = *reinterpret_cast<const f32x4 *>(attn + 2 * FUSED_D + 16 * m + 4 * qd);
#pragma unroll
      for (int t = 0; t < KT_S; ++t)
#pragma unroll
        for (int m = 0; m < MB_S; ++m) {
          if (COLD_S && t >= T_COLD_S)
            acc[m] = mfma3(__builtin_bit_cast(bf16x8, cold_s[(((t - T_COLD_S) * MB_S + m) * 2 + 0) * 64 + lane]),
                           __builtin_bit_cast(bf16x8, cold_s[(((t - T_COLD_S) * MB_S + m) * 2 + 1) * 64 + lane]), dh[t], dl[t], acc[m]);
          else
            acc[m] = mfma3(wsh[t][m], wsl[t][m], dh[t], dl[t], acc[m]);
        }
      // The slot is consumed (its values went through the split): refill it with the next snapshot's rows.  Issued BEHIND
      // the MFMAs on purpose: a DMA piece holds the wave at issue for 100+ cycles under load, and here the matrix pipe
      // works through the chain meanwhile (issued ahead of them, as at first, the pipe idled through the stall).
      if (more) {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        dma_sec(blk);
        vm_since_prim += np_sec;
        vm_since_sec = 0;
        __builtin_amdgcn_sched_barrier(0);
      }
      if (lrow < n_sec) {
#pragma unroll
        for (int m = 0; m < MB_S; ++m) {
          f32x4 o;
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = fused_act<ACT>(acc[m][j], a.act);
          *reinterpret_cast<f32x4 *>(sec + lrow * SEC_STRIDE + 16 * m + 4 * qd) = o;
        }
      }
    }
    if (more) sec_advance();
    UDS_STAMP(2);
  };

  unsigned p3_jb[U];      // P3: this lane's neighbour byte of each row group (static per tile; fetched at the end of P2)
  // ---------------- P2: [prim | agg] @ Wbig -> hx, attention scalars -> LDS ----------------
  auto phase2 = [&](int s, unsigned ag_locs, f32x4 ag_vals) {
    if (wave * 16 < n_prim) wait_vm(vm_since_prim);     // this wave's primary-row slots for snapshot s have landed
    const bool more = s + 1 < s_end;
    for (int blk = wave; blk * 16 < n_prim; blk += NW) {
      const float4 *st = reinterpret_cast<const float4 *>(stage_p + blk * (KT_X * 2 * 256)) + lane;
      bf16x8 dh[KT_B], dl[KT_B];
#pragma unroll
      for (int t = 0; t < KT_X; ++t) split8(st[(2 * t) * 64], st[(2 * t + 1) * 64], dh[t], dl[t]);
      UDS_STAMP(8);    // P2a: stage read + split
      const int lrow = blk * 16 + r16;
      const bool valid = lrow < n_prim;
      const int lr = min(lrow, n_prim - 1);
      // the k-steps over the row's own features do not need the aggregate: issue their MFMAs first, the matrix pipe
      // works on them while this wave refills the stage slot and gathers the incident secondary rows below
      f32x4 acc[MB_B];
#pragma unroll
      for (int m = 0; m < MB_B; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < KT_X; ++t)
#pragma unroll
        for (int m = 0; m < MB_B; ++m) {
          if (COLD_X && t == T_COLD_X)
            acc[m] = mfma3(__builtin_bit_cast(bf16x8, cold_b[(m * 2 + 0) * 64 + lane]), __builtin_bit_cast(bf16x8, cold_b[(m * 2 + 1) * 64 + lane]),
                           dh[t], dl[t], acc[m]);
          else
            acc[m] = mfma3(wbh[t][m], wbl[t][m], dh[t], dl[t], acc[m]);
        }
      // aggregation operands of this row: up to four (local secondary row, NodeEdge value) pairs at fixed width, read
      // with two independent LDS reads (the wave's first block has them since before the barrier)
      unsigned locs = ag_locs;
      f32x4 vals = ag_vals;
      if (blk != wave) {
        locs = inc_loc[lr];
        vals = inc_val4[lr];
      }
      if (more) {      // refill the consumed stage slot (behind the MFMAs: see P1)
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        dma_prim(blk);
        vm_since_sec += np_prim;
        vm_since_prim = 0;
        __builtin_amdgcn_sched_barrier(0);
      }
      UDS_STAMP(9);    // P2b: x-part MFMA issue + DMA issue
      float4 g0 = make_float4(0.f, 0.f, 0.f, 0.f), g1 = g0;   // this lane's 8 aggregate features (fragment shape)
      auto fma_row = [&](float wv, const float4 &u0, const float4 &u1) {
        g0.x = fmaf(wv, u0.x, g0.x); g0.y = fmaf(wv, u0.y, g0.y); g0.z = fmaf(wv, u0.z, g0.z); g0.w = fmaf(wv, u0.w, g0.w);
        g1.x = fmaf(wv, u1.x, g1.x); g1.y = fmaf(wv, u1.y, g1.y); g1.z = fmaf(wv, u1.z, g1.z); g1.w = fmaf(wv, u1.w, g1.w);
      };
      auto add2 = [&](unsigned la, float wa, unsigned lb, float wb) {      // two rows: four reads in flight, then the FMAs
        const float *ra = sec + la * SEC_STRIDE + 4 * qd, *rb = sec + lb * SEC_STRIDE + 4 * qd;
        const float4 a0 = *reinterpret_cast<const float4 *>(ra), a1 = *reinterpret_cast<const float4 *>(ra + 16);
        const float4 b0 = *reinterpret_cast<const float4 *>(rb), b1 = *reinterpret_cast<const float4 *>(rb + 16);
        fma_row(wa, a0, a1);
        fma_row(wb, b0, b1);
      };
#ifndef UDS_ABL_NO_AGG
      add2(locs & 0xffu, vals[0], (locs >> 8) & 0xffu, vals[1]);
      if (inc_width > 2) add2((locs >> 16) & 0xffu, vals[2], locs >> 24, vals[3]);
      if (flags & ELL_FLAG_INC_OVF)      // rows with more than four incident rows (a junction of five or more conduits)
        for (int p = ovf_ptr[lr]; p < ovf_ptr[lr + 1]; ++p) {
          const float *ra = sec + ovf_loc[p] * SEC_STRIDE + 4 * qd;
          fma_row(ovf_val[p], *reinterpret_cast<const float4 *>(ra), *reinterpret_cast<const float4 *>(ra + 16));
        }
#endif
      split8(g0, g1, dh[KT_X], dl[KT_X]);
      UDS_STAMP(10);   // P2c: aggregation
#pragma unroll
      for (int m = 0; m < MB_B; ++m) acc[m] = mfma3(wbh[KT_X][m], wbl[KT_X][m], dh[KT_X], dl[KT_X], acc[m]);
      UDS_STAMP(11);   // P2d: MFMA chain of the aggregate
      // <hx row, a_self>, <hx row, a_nbr>: even and odd columns in the two halves of packed FMAs (16 v_pk_fma_f32 instead of
      // 32 dependent v_fmac_f32), halves added, then the four lanes that share the row
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      f32x2 ps2 = {0.f, 0.f}, pn2 = {0.f, 0.f};
#pragma unroll
      for (int m = 0; m < MB_B; ++m) {
        const f32x4 as4 = *reinterpret_cast<const f32x4 *>(attn + 16 * m + 4 * qd);
        const f32x4 an4 = *reinterpret_cast<const f32x4 *>(attn + FUSED_D + 16 * m + 4 * qd);
        ps2 = __builtin_elementwise_fma(acc[m].xy, as4.xy, ps2);
        pn2 = __builtin_elementwise_fma(acc[m].xy, an4.xy, pn2);
        ps2 = __builtin_elementwise_fma(acc[m].zw, as4.zw, ps2);
        pn2 = __builtin_elementwise_fma(acc[m].zw, an4.zw, pn2);
      }
      float ps = quarters_sum(ps2.x + ps2.y);
      float pn = quarters_sum(pn2.x + pn2.y);
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
    if (more) prim_advance();
    // P3's neighbour bytes: static, one independent read per row group, back before the barrier releases
#pragma unroll
    for (int u = 0; u < U; ++u) p3_jb[u] = adj_b[min(wave * 4 + 4 * NW * u + rs, n_own - 1) * ELL_ADJ + c16];
    UDS_STAMP(4);
  };

  // ---------------- P3: segmented softmax + neighbour sum -> HBM ----------------
  // 16 lanes per output row, U row groups (4*U rows) per wave in flight at once so the dependent LDS reads of
  // one group hide behind the others.  Lane c scores neighbour c (one exp per neighbour, not per lane); the row
  // max / sum are DPP all-reduces inside the 16-lane group; weights and neighbour indices are then broadcast
  // lane by lane while every lane accumulates its own float4 feature chunk.
  auto phase3 = [&](int s) {
#ifndef UDS_ABL_NO_P3
    int n_st = 0;      // output-store instructions of this phase
    // rows are degree-sorted inside the tile; 4-row groups are dealt round-robin to the waves (group g -> wave g % NW),
    // so every wave gets the same mix of degrees and its groups come in descending degree
    {   // the planner keeps n_own <= 4*NW*U (= p_limit 128), so one trip covers the tile: row = wave*4 + 4*NW*u + rs
      int jn[U], dmax[U], orow[U];
      float ss[U], lg[U], wgt[U], den[U];
      bool ok[U], has[U];
      int dm = 0;
#pragma unroll
      for (int u = 0; u < U; ++u) {   // unconditional loads on clamped indices: the four groups' reads overlap
        const int i = wave * 4 + 4 * NW * u + rs;
        const int ic = min(i, n_own - 1);
        ok[u] = i < n_own;
        has[u] = ok[u] && p3_jb[u] != 0xFFu;      // neighbour slot c16 of this row exists (slots fill from 0)
        jn[u] = has[u] ? (int)p3_jb[u] : 0;
        dmax[u] = p3_dmax[u];
        ss[u] = s_self[ic];
        orow[u] = prim_ids[ic] * FUSED_D + 4 * c16;
        dm = max(dm, dmax[u]);
      }
      f32x4 acc[U];
#pragma unroll
      for (int u = 0; u < U; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (dm <= 16) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const float sv = ss[u] + s_nbr[jn[u]];
          const float sc = fmaxf(sv, 0.2f * sv);      // leaky_relu(0.2): the larger of v and 0.2 v, for either sign (2 ops, not 3)
          lg[u] = has[u] ? sc : -INFINITY;
        }
        int joff[U];      // byte offset of the neighbour's hx row with its swizzle key in bits 4-6: j*256 + (j&7)*16
        static_assert(U == 4, "the four-at-once reductions below");
        float mx[U];
        row16_max4(lg, mx);
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const float ex = __builtin_amdgcn_exp2f(lg[u] - mx[u]);      // logits are in base-2 units (see the set-up)
          wgt[u] = has[u] ? ex : 0.f;     // rows past the tile have no slots: every weight 0, the NaN of -inf - -inf dropped
          joff[u] = jn[u] * (FUSED_D * 4) + ((jn[u] & 7) << 4);
        }
        row16_sum4(wgt, den);
#ifndef UDS_ABL_NO_P3_STEPS
        // Lane c of a 16-lane row group holds (weight, offset) of neighbour c.  Neighbour K reaches the row's other
        // lanes by a DPP row broadcast (v_mov_b32_dpp row_newbcast:K): no LDS round trip for the pairs.  The XOR with
        // the lane's own chunk offset (c16 << 4) applies the hx swizzle: bits 4-6 key ^ chunk, bits >= 8 the row.
        // Two neighbour slots per step and row group, all reads of a step in flight together; groups whose longest
        // list is exhausted drop out (wave-uniform).  Slots beyond a row's degree hold weight 0 and row 0.
        const char *hxb = reinterpret_cast<const char *>(hx);
        const int cx = c16 << 4;
        auto step = [&](auto K_, auto A_) {
          constexpr int K = decltype(K_)::value, A = decltype(A_)::value, W = P3_SLOTS;
          f32x4 hh[A][W];
#pragma unroll
          for (int u = 0; u < A; ++u)
            static_for<W>([&](auto i_) {
              constexpr int i = decltype(i_)::value;
              hh[u][i] = *reinterpret_cast<const f32x4 *>(hxb + (row16_bcast<K + i>(joff[u]) ^ cx));
            });
#pragma unroll
          for (int u = 0; u < A; ++u)
            static_for<W>([&](auto i_) {
              constexpr int i = decltype(i_)::value;
              const float w = __int_as_float(row16_bcast<K + i>(__float_as_int(wgt[u])));
#pragma unroll
              for (int q = 0; q < 4; ++q) acc[u][q] = fmaf(w, hh[u][i][q], acc[u][q]);
            });
        };
        // The degree bounds never change from snapshot to snapshot, so the compiler hoists all 32 "slot K < bound" tests out
        // of the snapshot loop as 64-bit masks, spills them to VGPR lanes and reads them back with v_readlane (vector-ALU
        // issue slots) at every branch.  Laundering the four scalars keeps the tests where they are: s_cmp on the scalar unit.
        int d0 = dmax[0], d1 = dmax[1], d2 = dmax[2], d3 = dmax[3];
        asm volatile("" : "+s"(d0), "+s"(d1), "+s"(d2), "+s"(d3));
        const int e3 = d3, e2 = max(e3, d2), e1 = max(e2, d1), e0 = max(e1, d0);
        auto steps = [&](auto K_) {      // slots K .. K+W-1: as many groups as still have neighbours there
          constexpr int K = decltype(K_)::value;
          if (K < e3) step(K_, std::integral_constant<int, 4>{});
          else if (K < e2) step(K_, std::integral_constant<int, 3>{});
          else if (K < e1) step(K_, std::integral_constant<int, 2>{});
          else step(K_, std::integral_constant<int, 1>{});
        };
        bool more = e0 > 0;
        static_for<16 / P3_SLOTS>([&](auto t_) {
          constexpr int K = decltype(t_)::value * P3_SLOTS;
          if (more) {
            steps(std::integral_constant<int, K>{});
            more = e0 > K + P3_SLOTS;
          }
        });
#else
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u] = f32x4{wgt[u], __int_as_float(joff[u]), wgt[u], wgt[u]};
#endif
      } else {   // some row has more than 16 neighbours: every lane walks its row's whole list (tiles with ELL_FLAG_LONG_ROWS)
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int i = wave * 4 + 4 * NW * u + rs;
          const int b0 = i < n_own ? adj_ptr[i] : 0;
          const int dg = i < n_own ? adj_ptr[i + 1] - b0 : 0;
          float mx = -INFINITY;
          for (int p = b0; p < b0 + dg; ++p) mx = fmaxf(mx, leaky02(ss[u] + s_nbr[adj_loc[p]]));
          den[u] = 0.f;
          for (int p = b0; p < b0 + dg; ++p) {
            const int jj = adj_loc[p];
            const float wv = __builtin_amdgcn_exp2f(leaky02(ss[u] + s_nbr[jj]) - mx);
            const f32x4 hv = *reinterpret_cast<const f32x4 *>(hx + jj * FUSED_D + ((c16 ^ (jj & 7)) << 2));
            den[u] += wv;
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[u][q] = fmaf(wv, hv[q], acc[u][q]);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (dmax[u] > 0) ++n_st;       // wave-uniform: one store instruction per row group that has a valid row
        if (ok[u]) {
          const float inv = __builtin_amdgcn_rcpf(den[u]);
          f32x4 o;
#pragma unroll
          for (int q = 0; q < 4; ++q) o[q] = fused_act<ACT>(fmaf(acc[u][q], inv, bo[q]), a.act);
#ifndef UDS_ABL_NO_STORE
          *reinterpret_cast<f32x4 *>(S_.out + ((int64_t)s * S_.n_prim_glob * FUSED_D + orow[u])) = o;
#else
          asm volatile("" ::"v"(o[0]), "v"(o[1]), "v"(o[2]), "v"(o[3]), "v"(orow[u]));
#endif
        }
      }
    }
#ifndef UDS_ABL_NO_STORE
    vm_since_sec += n_st;
    vm_since_prim += n_st;
#endif
#endif
    UDS_STAMP(6);
  };

#ifdef UDS_ABL_NO_BAR
#define UDS_BAR() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#else
#define UDS_BAR() lds_barrier()
#endif
  if (s_begin < s_end) phase1(s_begin);
  for (int s = s_begin; s < s_end; ++s) {
    // index lists do not depend on the data: fetch the aggregation operands of this wave's first P2 block now, so the
    // reads are back by the time the barrier releases (loads on clamped indices, no branches)
    const int lr0 = min(wave * 16 + r16, n_prim - 1);
    const unsigned ag_locs = inc_loc[lr0];
    const f32x4 ag_vals = inc_val4[lr0];
    UDS_BAR();      // every secondary row of snapshot s is in LDS
    UDS_STAMP(3);
    phase2(s, ag_locs, ag_vals);
    UDS_BAR();      // hx and the attention scalars of snapshot s are in LDS; nobody reads `sec` any more
    UDS_STAMP(5);
    // P3 of this snapshot and P1 of the next touch disjoint LDS (hx / scores vs the stage and `sec`).  No barrier after
    // them: the next P2 writes hx / scores only behind the next barrier, which every wave reaches after its own P3.
    phase3(s);
    if (s + 1 < s_end) phase1(s + 1);
  }
#undef UDS_BAR
#ifdef UDS_PHASE_TIMING
  if (a.dbg && lane == 0) {
    unsigned long long *o = a.dbg + ((size_t)blockIdx.x * NW + wave) * 16;
    o[12] = wave;
    o[15] = ((__builtin_amdgcn_s_memrealtime() - rt0_) << 32) | ((clock64() - mt0_) & 0xffffffffull);
    o[13] = tm_[7];
    o[14] = t_loads_issued_ - t_dma_issued_;
    for (int k = 0; k < 7; ++k) o[k] = tm_[k];
    o[8] = tm_[8]; o[9] = tm_[9]; o[10] = tm_[10]; o[11] = tm_[11];
    o[7] = 1ull | ((unsigned long long)sd << 8) | ((unsigned long long)n_own << 16) | ((unsigned long long)n_prim << 32) |
           ((unsigned long long)n_sec << 48);
  }
#endif
}

}  // namespace uds
