// Wave-specialised form of the fused spatial-layer kernel for 64-wide rows (every layer of a d = 64 block but the
// 96-wide first layer of block 2): same tiles, same tile blocks, same arithmetic and the same results, bit for bit,
// as k_fused_tile<64, 64> (kernels_fused.hpp) -- what changes is WHO does what and WHEN.
//
// k_fused_tile runs its eight waves through the phases in lock-step (all multiply, then all gather), every wave keeps
// all the weights (128 VGPRs), every input row passes through an LDS stage, and there are two workgroup barriers per
// snapshot.  Here the workgroup is two teams, software-pipelined over the snapshots, ONE barrier per snapshot:
//
//   interval k     "Y" team (waves 0 .. NY-1): the big weights (96 VGPRs)      "X" team (the rest): the small weights (32 VGPRs)
//                  P2(s):  [prim | agg] @ Wbig -> hx[k & 1], scores[k & 1]      P1(s+1): secondary MLP -> sec[(k+1) & 1]
//                          prim rows of s+1 -> registers                                 secondary rows of s+2 -> registers
//                                                                               P3(s-1): softmax + neighbour sum from hx[(k-1) & 1] -> HBM
//   ---- barrier ----
//
// * No input row touches LDS: each Y wave loads the primary blocks it multiplies straight into the registers its split
//   reads, each X wave the secondary blocks it feeds to the small GEMM, both one interval ahead.  LDS holds only what is
//   shared: sec (x2), hx (x2), the attention scalars (x2) and the tile block: 135 KB instead of 157, and ~30 % less LDS
//   traffic per snapshot (the stage's writes and reads are gone).
// * Registers follow the roles: Y keeps Wbig and no P3 state, X keeps Wsmall, the rows in flight and the P3 state.
// * One barrier per interval orders everything: P1(s+1) writes the sec buffer P2(s-1) read an interval ago, P2(s) writes
//   the hx buffer P3(s-2) read an interval ago.
// * Global accesses: Y's row loads are ordinary loads (it stores nothing: the compiler's vmcnt counting is exact).  X
//   stores its outputs behind its row loads, and the compiler -- which merges its pending-operation state over the
//   exec-masked store branches -- would wait vmcnt(0) for those fresh stores before the next P1; X's row loads are
//   therefore inline asm and their one wait is counted by hand: vmcnt(number of stores issued since).
#pragma once
#include "kernels_fused.hpp"

namespace uds {

#ifndef UDS_WS_NY
#define UDS_WS_NY 4
#endif
#ifndef UDS_WS_SLOTS
#define UDS_WS_SLOTS 4      // measured: 4 slots 257 us, 2 slots 265 us per launch at the headline size (X has the registers: no weights)
#endif
constexpr int WS_NY = UDS_WS_NY, WS_NX = FUSED_WAVES - WS_NY;      // team sizes
#ifdef UDS_WS_YG
constexpr int WS_YG = UDS_WS_NY;        // experiment: one P3 row group (the NY highest-degree ones) per Y wave, in the time Y waits at the barrier
#else                                   // (measured: 263 us against 259 us without -- the Y waves' P3 costs the X partner what it saves it)
constexpr int WS_YG = 0;
#endif
constexpr int WS_SLOTS = UDS_WS_SLOTS;                              // neighbour slots per P3 step (reads in flight per row group)
#ifndef UDS_WS_PRE
#define UDS_WS_PRE 0
#endif
constexpr int WS_PRE = UDS_WS_PRE;      // P3: neighbour slots whose hx rows are fetched BEFORE the softmax chain (their addresses are static)
static_assert(WS_PRE % WS_SLOTS == 0, "WS_PRE must be a multiple of WS_SLOTS");
#ifndef UDS_WS_XPRIO
#define UDS_WS_XPRIO 1      // s_setprio of the X team's waves (the Y team stays at 0)
#endif
static_assert(WS_NY >= 2 && WS_NY <= 4, "2 .. 4 waves multiply the primary rows");

// LDS bytes: tile block + 2 x (s_self, s_nbr) + attention vectors / bias + 2 x sec rows + 2 x hx rows
inline int64_t fused_ws_lds_bytes(int p_cap, int q_cap, int meta_cap) {
  return 4 * ((int64_t)meta_cap + 4 * p_cap + 3 * FUSED_D + FUSED_H + 2 * (int64_t)q_cap * SEC_STRIDE + 2 * (int64_t)p_cap * FUSED_D);
}

// 16 bytes per lane from base (wave-uniform, SGPR pair) + voff (per-lane byte offset) + IMM, invisible to the compiler's
// vmcnt bookkeeping (see the header): the value is NOT there when the statement ends -- ws_vmcnt() makes it so.
template <int IMM>
__device__ __forceinline__ void ws_gld16(f32x4 &dst, const float *base, unsigned voff) {
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(base), "n"(IMM) : "memory");
}
// tie four row registers to a point of the program: nothing that reads them is scheduled above it, nothing that writes them below
__device__ __forceinline__ void ws_pin4(f32x4 &a0, f32x4 &a1, f32x4 &a2, f32x4 &a3) {
  asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
}
// 8-lane all-reduce (max / sum) of FOUR values at once with the DPP move folded into the operation, as row16_max4 / row16_sum4
// (kernels_fused.hpp) but for the octet layout of P3 (8 lanes per output row): quad_perm x2, then row_half_mirror (lane i <-> 7 - i
// inside every 8 lanes) joins the two quads.
__device__ __forceinline__ void row8_max4(const float (&in)[4], float (&out)[4]) {
  asm("s_nop 1\n\t" UDS_DPP4_FIRST("v_max_f32_dpp", "quad_perm:[1,0,3,2]") UDS_DPP4("v_max_f32_dpp", "quad_perm:[2,3,0,1]")
      UDS_DPP4("v_max_f32_dpp", "row_half_mirror")
      : "=&v"(out[0]), "=&v"(out[1]), "=&v"(out[2]), "=&v"(out[3])
      : "v"(in[0]), "v"(in[1]), "v"(in[2]), "v"(in[3]));
}
__device__ __forceinline__ void row8_sum4(const float (&in)[4], float (&out)[4]) {
  asm("s_nop 1\n\t" UDS_DPP4_FIRST("v_add_f32_dpp", "quad_perm:[1,0,3,2]") UDS_DPP4("v_add_f32_dpp", "quad_perm:[2,3,0,1]")
      UDS_DPP4("v_add_f32_dpp", "row_half_mirror")
      : "=&v"(out[0]), "=&v"(out[1]), "=&v"(out[2]), "=&v"(out[3])
      : "v"(in[0]), "v"(in[1]), "v"(in[2]), "v"(in[3]));
}
// 16-bit half HI of pk (zero-extended) + base, one SDWA instruction.  Written as C the compiler unpacks every half it will
// need into a register of its own ahead of the snapshot loop: 32 registers the octet P3 does not have.
template <int HI>
__device__ __forceinline__ unsigned ws_add_half(unsigned pk, unsigned base) {
  unsigned r;
  if (HI) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "=v"(r) : "v"(pk), "v"(base));
  else asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD" : "=v"(r) : "v"(pk), "v"(base));
  return r;
}
// wait until all but the n youngest vector-memory operations of this wave are done (n <= 8, exact)
__device__ __forceinline__ void ws_vmcnt(int n) {
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
  }
}

template <int ACT>
__global__ __launch_bounds__(FUSED_WAVES * 64, 2) void k_fused_ws(FusedArgs a) {
  constexpr int NT = FUSED_WAVES * 64, NY = WS_NY, NX = WS_NX;
  constexpr int FP = 64, FS = 64;
  constexpr int KT_S = FS / 32, MB_S = FUSED_H / 16;                    // small GEMM: 64 -> 32
  constexpr int KT_X = FP / 32, KT_B = KT_X + 1, MB_B = FUSED_D / 16;   // big GEMM: 64 + 32 -> 64
  constexpr int U = FUSED_U;
  constexpr int PJ = (8 + NY - 1) / NY;        // primary 16-row blocks per Y wave (p_cap <= 128)
  constexpr int SJ = (16 + NX - 1) / NX;       // secondary 16-row blocks per X wave (q_cap <= 256)
  extern __shared__ __attribute__((aligned(16))) int32_t smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef UDS_WS_PAIR_SAME
  // waves w and w + 4 share a SIMD: put each team on SIMDs of its own (NY = 4: Y on the SIMDs of waves 0/4 and 1/5, X on 2/6 and 3/7)
  static_assert(NY == 4, "UDS_WS_PAIR_SAME assumes 4 + 4");
  const bool team_y = (wave & 3) < 2;
  const int yw = (wave & 1) + 2 * (wave >> 2), xw = ((wave & 3) - 2) + 2 * (wave >> 2);
#else
  const bool team_y = wave < NY;                 // wave-uniform role
  const int yw = wave, xw = wave - NY;           // index inside the team
#endif
  const int r16 = lane & 15, qd = lane >> 4;
  const int c16 = r16, rs = qd;                  // P3 mapping: 16 lanes x float4 per output row, 4 rows per group

  // XCD-aware bijective remap (as k_fused_tile): workgroups b, b+8, .. share an XCD and get a contiguous range of items
  const int W = gridDim.x, b = blockIdx.x;
  const int q8 = W / 8, r8 = W % 8, xcd = b % 8;
  const int w = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + b / 8;
  const int tile = w % a.n_tiles;
  const int chunk_id = w / a.n_tiles;
  const int s_begin = chunk_id * a.chunk;
  const int s_end = min(a.S, (chunk_id + 1) * a.chunk);
  const int sd = __builtin_amdgcn_readfirstlane(a.hdr[tile * TILE_HDR_INTS + 6]);
  if (!((a.side_mask >> sd) & 1)) return;
  const FusedSide &S_ = a.side[sd];

  float *s_self = reinterpret_cast<float *>(smem + a.meta_cap);         // [2][p_cap]
  float *s_nbr = s_self + 2 * a.p_cap;                                   // [2][p_cap]
  float *attn = s_nbr + 2 * a.p_cap;             // a_self[64] | a_nbr[64] | b_small[32] | b_out[64]
  float *sec = attn + 3 * FUSED_D + FUSED_H;     // [2][q_cap * SEC_STRIDE]
  float *hx = sec + 2 * a.q_cap * SEC_STRIDE;    // [2][p_cap * 64]
  const int sec_buf = a.q_cap * SEC_STRIDE, hx_buf = a.p_cap * FUSED_D;

  // ---- set-up: tile block, attention vectors (each team loads its own weights inside its branch) ----
  {
    const uint4 *src = reinterpret_cast<const uint4 *>(a.blocks + (int64_t)tile * a.meta_cap);
    uint4 *dst = reinterpret_cast<uint4 *>(smem);
    for (int i = tid; i < a.meta_cap / 4; i += NT) dst[i] = src[i];
  }
  if (tid < FUSED_D) {   // scaled by log2(e): scores are logits in base-2 units, P3 needs no multiply in front of its exp2
    attn[tid] = S_.a_self[tid] * 1.44269504088896340736f;
    attn[FUSED_D + tid] = S_.a_nbr[tid] * 1.44269504088896340736f;
    if (tid < FUSED_H) attn[2 * FUSED_D + tid] = S_.b_small ? S_.b_small[tid] : 0.f;
    attn[2 * FUSED_D + FUSED_H + tid] = S_.b_out ? S_.b_out[tid] : 0.f;
  }
  __syncthreads();
  const int n_own = __builtin_amdgcn_readfirstlane(smem[0]), n_prim = __builtin_amdgcn_readfirstlane(smem[1]),
            n_sec = __builtin_amdgcn_readfirstlane(smem[2]), flags = __builtin_amdgcn_readfirstlane(smem[3]),
            n_ovf = __builtin_amdgcn_readfirstlane(smem[4]), n_adj = __builtin_amdgcn_readfirstlane(smem[5]),
            inc_width = __builtin_amdgcn_readfirstlane(smem[7]);
  const EllOffsets off = ell_offsets(n_own, n_prim, n_sec, flags, n_ovf, n_adj);
  const int32_t *prim_ids = smem + off.prim;
  const int32_t *sec_ids = smem + off.sec;
  const uint32_t *inc_loc = reinterpret_cast<const uint32_t *>(smem + off.inc_loc);
  int32_t *inc_w = smem + off.inc_w;
  const f32x4 *inc_val4 = reinterpret_cast<const f32x4 *>(inc_w);
  const unsigned char *adj_b = reinterpret_cast<const unsigned char *>(smem + off.adj);
  const int32_t *ovf_ptr = smem + off.ovf_ptr, *ovf_loc = smem + off.ovf_loc;
  int32_t *ovf_w = smem + off.ovf_w;
  const int32_t *adj_ptr = smem + off.adj_ptr, *adj_loc = smem + off.adj_loc;

  for (int i = tid; i < ELL_INC * n_prim; i += NT) {      // NodeEdge values of the fixed-width lists (0 for padding)
    const int k = inc_w[i];
    reinterpret_cast<float *>(inc_w)[i] = k >= 0 ? S_.ne_val[k] : 0.f;
  }
  if (flags & ELL_FLAG_INC_OVF)
    for (int i = tid; i < n_ovf; i += NT) reinterpret_cast<float *>(ovf_w)[i] = S_.ne_val[ovf_w[i]];
  const float *ovf_val = reinterpret_cast<const float *>(ovf_w);

#ifdef UDS_PHASE_TIMING
  unsigned long long tm_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tl_ = clock64();
  const unsigned long long rt0_ = __builtin_amdgcn_s_memrealtime(), mt0_ = tl_;
#define WS_STAMP(k) do { const unsigned long long n_ = clock64(); tm_[k] += n_ - tl_; tl_ = n_; } while (0)
#define WS_DUMP() do { if (a.dbg && lane == 0) { unsigned long long *o = a.dbg + ((size_t)blockIdx.x * 8 + wave) * 16; \
    for (int q_ = 0; q_ < 7; ++q_) o[q_] = tm_[q_]; o[12] = wave; \
    o[15] = ((__builtin_amdgcn_s_memrealtime() - rt0_) << 32) | ((clock64() - mt0_) & 0xffffffffull); \
    o[7] = 1ull | ((unsigned long long)sd << 8) | ((unsigned long long)n_own << 16) | ((unsigned long long)n_prim << 32) | ((unsigned long long)n_sec << 48); } } while (0)
#else
#define WS_STAMP(k) do { } while (0)
#define WS_DUMP() do { } while (0)
#endif
  const int64_t sec_stride = (int64_t)S_.n_sec_glob * FS, prim_stride = (int64_t)S_.n_prim_glob * FP;      // floats per snapshot
  const int n_snap = s_end - s_begin;

  __syncthreads();   // NodeEdge values are in LDS
  // The two teams run two separate loops (the same number of barriers in each): what a team keeps in registers across the
  // snapshots -- the big weights here, the small weights, rows in flight and P3 state there -- is live in its own loop only.
  if (team_y) {
#ifndef UDS_WS_MFMA16
    // =========================== Y team: P2 on v_mfma_f32_32x32x16_bf16 ===========================
    // One 32-row block per wave (rows 32 yw ..).  Same flops per cycle as the 16x16x32 form, but an MFMA holds its SIMD's
    // vector issue for 8 of its 32 cycles instead of 8 of 16 (MI355X_MICROARCH.md): the X wave on this SIMD -- the long
    // instruction stream of the kernel -- loses half as many issue slots to this wave's matrix work.  Lane l: n = l & 31 is
    // the row, hf = l >> 5 the k half: B fragment of k-step t = floats 16 t + 8 hf .. + 7 of the row; the accumulators hold
    // features (r & 3) + 8 (r >> 2) + 4 hf (+ 32 m) of row n (cdna_hip_programming.md section 3).
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    constexpr int KT32 = (FP + FUSED_H) / 16, KX32 = FP / 16, MB32 = FUSED_D / 32;      // 6 k-steps (4 of x, 2 of agg), 2 feature blocks
    static_assert(NY == 4, "one 32-row block per Y wave: four waves cover p_cap = 128");
    const int n32 = lane & 31, hf = lane >> 5;
    bf16x8 wh[KT32][MB32], wl[KT32][MB32];       // 96 VGPRs, resident across the snapshot loop
#pragma unroll
    for (int t = 0; t < KT32; ++t)
#pragma unroll
      for (int m = 0; m < MB32; ++m) {
        wh[t][m] = __builtin_bit_cast(bf16x8, S_.w_big32[((t * MB32 + m) * 2 + 0) * 64 + lane]);
        wl[t][m] = __builtin_bit_cast(bf16x8, S_.w_big32[((t * MB32 + m) * 2 + 1) * 64 + lane]);
      }
    const int lrow = 32 * yw + n32;
    const bool valid = lrow < n_prim;
    const int lr = min(lrow, n_prim - 1);
    const unsigned prow = (unsigned)prim_ids[lr] * FP + 8u * hf;      // element offset of this lane's 8-float piece of k-step 0
    f32x4 pp[2 * KX32];           // the primary row of the snapshot P2 multiplies next: piece i = floats 16 (i >> 1) + 8 hf + 4 (i & 1)
    // the dense remainder of a trained NodeEdge (FusedSide::rem, (S, n_prim_glob, 32)): this lane's 16 aggregate features of its
    // row, fetched with the row itself one interval ahead and used as the initial value of the aggregate
    const bool has_rem = S_.rem != nullptr;
    const unsigned rrow = (unsigned)prim_ids[lr] * FUSED_H + 8u * hf;
    f32x4 rm[4] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    const int64_t rem_stride = (int64_t)S_.n_prim_glob * FUSED_H;
    const unsigned ag_locs = inc_loc[lr];
    const f32x4 ag_vals = inc_val4[lr];
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): drain the set-up loads on every path (see the X team's note)
    if (n_snap > 0) {
      const float *base = S_.prim_in + s_begin * prim_stride;
      static_for<2 * KX32>([&](auto i_) {
        constexpr int i = decltype(i_)::value;
        pp[i] = *reinterpret_cast<const f32x4 *>(base + prow + 16 * (i >> 1) + 4 * (i & 1));
      });
      if (has_rem) {
        const float *rb = S_.rem + s_begin * rem_stride + rrow;
        rm[0] = *reinterpret_cast<const f32x4 *>(rb);
        rm[1] = *reinterpret_cast<const f32x4 *>(rb + 4);
        rm[2] = *reinterpret_cast<const f32x4 *>(rb + 16);
        rm[3] = *reinterpret_cast<const f32x4 *>(rb + 20);
      }
    }
    WS_STAMP(0);
    lds_barrier();      // the X team has computed the secondary MLP of the first snapshot

    auto mfma3_32 = [&](const bf16x8 &a_h, const bf16x8 &a_l, const bf16x8 &d_h, const bf16x8 &d_l, f32x16 acc) __attribute__((always_inline)) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_h, d_l, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_l, d_h, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_h, d_h, acc, 0, 0, 0);
      return acc;
    };
    for (int k = 0; k <= n_snap; ++k) {
#ifdef UDS_WS_ABL_NO_P2
      if (false) {
#else
      if (k < n_snap && 32 * yw < n_prim) {
#endif
        const bool more = k + 1 < n_snap;
        const float *next = S_.prim_in + (s_begin + k + 1) * prim_stride;
        const float *secr = sec + (k & 1) * sec_buf;
        float *hxw = hx + (k & 1) * hx_buf;
        float *ssw = s_self + (k & 1) * a.p_cap, *snw = s_nbr + (k & 1) * a.p_cap;
        bf16x8 dh[KT32], dl[KT32];
        static_for<KX32>([&](auto t_) {
          constexpr int t = decltype(t_)::value;
          const f32x4 u0 = pp[2 * t], u1 = pp[2 * t + 1];
          split8(make_float4(u0[0], u0[1], u0[2], u0[3]), make_float4(u1[0], u1[1], u1[2], u1[3]), dh[t], dl[t]);
        });
        if (more)      // the registers are free: the next snapshot's row, a whole interval ahead
          static_for<2 * KX32>([&](auto i_) {
            constexpr int i = decltype(i_)::value;
            pp[i] = *reinterpret_cast<const f32x4 *>(next + prow + 16 * (i >> 1) + 4 * (i & 1));
          });
        f32x16 acc[MB32];
#pragma unroll
        for (int m = 0; m < MB32; ++m)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
#pragma unroll
        for (int t = 0; t < KX32; ++t)
#pragma unroll
          for (int m = 0; m < MB32; ++m) acc[m] = mfma3_32(wh[t][m], wl[t][m], dh[t], dl[t], acc[m]);
        // NodeEdge aggregate of this row in fragment shape: features 8 hf .. + 7 (k-step 4) and 16 + 8 hf .. + 7 (k-step 5)
        float4 g0 = make_float4(rm[0][0], rm[0][1], rm[0][2], rm[0][3]), g1 = make_float4(rm[1][0], rm[1][1], rm[1][2], rm[1][3]),
               g2 = make_float4(rm[2][0], rm[2][1], rm[2][2], rm[2][3]), g3 = make_float4(rm[3][0], rm[3][1], rm[3][2], rm[3][3]);
        if (more && has_rem) {      // consumed: the next snapshot's remainder piece
          const float *rb = S_.rem + (s_begin + k + 1) * rem_stride + rrow;
          rm[0] = *reinterpret_cast<const f32x4 *>(rb);
          rm[1] = *reinterpret_cast<const f32x4 *>(rb + 4);
          rm[2] = *reinterpret_cast<const f32x4 *>(rb + 16);
          rm[3] = *reinterpret_cast<const f32x4 *>(rb + 20);
        }
        auto fma4 = [&](float4 &g, float wv, const float4 &u) { g.x = fmaf(wv, u.x, g.x); g.y = fmaf(wv, u.y, g.y); g.z = fmaf(wv, u.z, g.z); g.w = fmaf(wv, u.w, g.w); };
        auto add_row = [&](unsigned la, float wa) {
          const float *ra = secr + la * SEC_STRIDE + 8 * hf;
          const float4 u0 = *reinterpret_cast<const float4 *>(ra), u1 = *reinterpret_cast<const float4 *>(ra + 4);
          const float4 u2 = *reinterpret_cast<const float4 *>(ra + 16), u3 = *reinterpret_cast<const float4 *>(ra + 20);
          fma4(g0, wa, u0); fma4(g1, wa, u1); fma4(g2, wa, u2); fma4(g3, wa, u3);
        };
        add_row(ag_locs & 0xffu, ag_vals[0]);
        add_row((ag_locs >> 8) & 0xffu, ag_vals[1]);
        if (inc_width > 2) {
          add_row((ag_locs >> 16) & 0xffu, ag_vals[2]);
          add_row(ag_locs >> 24, ag_vals[3]);
        }
        if (flags & ELL_FLAG_INC_OVF)      // rows with more than four incident rows
          for (int p = ovf_ptr[lr]; p < ovf_ptr[lr + 1]; ++p) add_row((unsigned)ovf_loc[p], ovf_val[p]);
        split8(g0, g1, dh[KX32], dl[KX32]);
        split8(g2, g3, dh[KX32 + 1], dl[KX32 + 1]);
#pragma unroll
        for (int t = KX32; t < KT32; ++t)
#pragma unroll
          for (int m = 0; m < MB32; ++m) acc[m] = mfma3_32(wh[t][m], wl[t][m], dh[t], dl[t], acc[m]);
        // attention scalars: this lane holds 32 of the row's 64 features, lane l ^ 32 the others
        float ps = 0.f, pn = 0.f;
#pragma unroll
        for (int m = 0; m < MB32; ++m)
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            const f32x4 as4 = *reinterpret_cast<const f32x4 *>(attn + 32 * m + 8 * gq + 4 * hf);
            const f32x4 an4 = *reinterpret_cast<const f32x4 *>(attn + FUSED_D + 32 * m + 8 * gq + 4 * hf);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              ps = fmaf(acc[m][4 * gq + q], as4[q], ps);
              pn = fmaf(acc[m][4 * gq + q], an4[q], pn);
            }
          }
        ps += __shfl_xor(ps, 32);
        pn += __shfl_xor(pn, 32);
        if (valid) {
          if (hf == 0) {
            ssw[lrow] = ps;
            snw[lrow] = pn;
          }
#pragma unroll
          for (int m = 0; m < MB32; ++m)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {      // 16-byte chunk 8 m + 2 gq + hf of the row, XOR (row & 7): as the 16x16x32 form writes it
              const f32x4 o = {acc[m][4 * gq], acc[m][4 * gq + 1], acc[m][4 * gq + 2], acc[m][4 * gq + 3]};
              *reinterpret_cast<f32x4 *>(hxw + lrow * FUSED_D + (((8 * m + 2 * gq + hf) ^ (lrow & 7)) << 2)) = o;
            }
        }
      }
      WS_STAMP(1);
      lds_barrier();      // hx / scores of snapshot k are in LDS (and the X team's sec rows of snapshot k + 1)
      WS_STAMP(2);
    }
    WS_DUMP();
#else
    // =========================== Y team: P2 ===========================
    bf16x8 wbh[KT_B][MB_B], wbl[KT_B][MB_B];      // 96 VGPRs, resident across the snapshot loop
#pragma unroll
    for (int t = 0; t < KT_B; ++t)
#pragma unroll
      for (int m = 0; m < MB_B; ++m) {
        wbh[t][m] = __builtin_bit_cast(bf16x8, S_.w_big[((t * MB_B + m) * 2 + 0) * 64 + lane]);
        wbl[t][m] = __builtin_bit_cast(bf16x8, S_.w_big[((t * MB_B + m) * 2 + 1) * 64 + lane]);
      }
    // element offsets of the rows of this wave's primary blocks (blk = yw + NY j); a lane's piece c of a row = floats 16 c + 4 qd
    unsigned prow[PJ];
#pragma unroll
    for (int j = 0; j < PJ; ++j) prow[j] = (unsigned)prim_ids[min((yw + NY * j) * 16 + r16, n_prim - 1)] * FP + 4u * qd;
    f32x4 pp[PJ][2 * KT_X];       // the primary rows of the snapshot P2 multiplies next (compile-time indices only: registers)
    // aggregation operands of the primary blocks (static per tile)
    unsigned ag_locs[PJ];
    f32x4 ag_vals[PJ];
#pragma unroll
    for (int j = 0; j < PJ; ++j) {
      const int lr = min((yw + NY * j) * 16 + r16, n_prim - 1);
      ag_locs[j] = inc_loc[lr];
      ag_vals[j] = inc_val4[lr];
    }
    // P3 of ONE 4-row group per Y wave (group yw: the highest degrees of the tile), static per tile
    const int yg_i = 4 * yw + rs, yg_ic = min(yg_i, n_own - 1);
    const bool yg_ok = WS_YG > 0 && yg_i < n_own;
    const unsigned yg_jb = adj_b[yg_ic * ELL_ADJ + c16];
    const bool yg_has = yg_ok && yg_jb != 0xFFu;
    const int yg_jn = yg_has ? (int)yg_jb : 0;
    const int yg_joff = yg_jn * (FUSED_D * 4) + ((yg_jn & 7) << 4);
    const int yg_orow = prim_ids[yg_ic] * FUSED_D + 4 * c16;
    int yg_dmax;
    {
      int dmx;
      if (flags & ELL_FLAG_LONG_ROWS) {
        dmx = yg_i < n_own ? adj_ptr[yg_ic + 1] - adj_ptr[yg_ic] : 0;
      } else {
        const unsigned long long m = __ballot(yg_i < n_own && adj_b[yg_ic * ELL_ADJ + c16] != 0xFF);
        dmx = __builtin_popcountll((m >> (16 * rs)) & 0xffffull);
      }
      dmx = max(dmx, __shfl_xor(dmx, 16));
      dmx = max(dmx, __shfl_xor(dmx, 32));
      yg_dmax = WS_YG > 0 ? __builtin_amdgcn_readfirstlane(dmx) : 0;
    }
    f32x4 bo = f32x4{0.f, 0.f, 0.f, 0.f};
    if (S_.b_out) bo = *reinterpret_cast<const f32x4 *>(S_.b_out + 4 * c16);
    auto phase3_y = [&](int s, int buf) __attribute__((always_inline)) {
      if (yg_dmax == 0) return;
      const float *hxr = hx + buf * hx_buf;
      const float *ssr = s_self + buf * a.p_cap, *snr = s_nbr + buf * a.p_cap;
      const float ss = ssr[yg_ic];
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
      float den;
      if (yg_dmax <= 16) {
        const float sv = ss + snr[yg_jn];
        const float lg = yg_has ? fmaxf(sv, 0.2f * sv) : -INFINITY;
        const float mx = row16_max(lg);
        const float ex = __builtin_amdgcn_exp2f(lg - mx);
        const float wgt = yg_has ? ex : 0.f;
        den = row16_sum(wgt);
        const char *hxb = reinterpret_cast<const char *>(hxr);
        const int cx = c16 << 4;
        int dmx = yg_dmax;
        asm volatile("" : "+s"(dmx));
        static_for<16 / WS_SLOTS>([&](auto t_) {
          constexpr int K = decltype(t_)::value * WS_SLOTS;
          if (K < dmx) {
            f32x4 hh[WS_SLOTS];
            static_for<WS_SLOTS>([&](auto i_) {
              constexpr int i = decltype(i_)::value;
              hh[i] = *reinterpret_cast<const f32x4 *>(hxb + (row16_bcast<K + i>(yg_joff) ^ cx));
            });
            static_for<WS_SLOTS>([&](auto i_) {
              constexpr int i = decltype(i_)::value;
              const float wv = __int_as_float(row16_bcast<K + i>(__float_as_int(wgt)));
#pragma unroll
              for (int q = 0; q < 4; ++q) acc[q] = fmaf(wv, hh[i][q], acc[q]);
            });
          }
        });
      } else {      // a row with more than 16 neighbours: walk the whole list
        const int b0 = yg_ok ? adj_ptr[yg_i] : 0;
        const int dg = yg_ok ? adj_ptr[yg_i + 1] - b0 : 0;
        float mx = -INFINITY;
        for (int p = b0; p < b0 + dg; ++p) mx = fmaxf(mx, leaky02(ss + snr[adj_loc[p]]));
        den = 0.f;
        for (int p = b0; p < b0 + dg; ++p) {
          const int jj = adj_loc[p];
          const float wv = __builtin_amdgcn_exp2f(leaky02(ss + snr[jj]) - mx);
          const f32x4 hv = *reinterpret_cast<const f32x4 *>(hxr + jj * FUSED_D + ((c16 ^ (jj & 7)) << 2));
          den += wv;
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[q] = fmaf(wv, hv[q], acc[q]);
        }
      }
      if (yg_ok) {
        const float inv = __builtin_amdgcn_rcpf(den);
        f32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = fused_act<ACT>(fmaf(acc[q], inv, bo[q]), a.act);
        *reinterpret_cast<f32x4 *>(S_.out + ((int64_t)s * S_.n_prim_glob * FUSED_D + yg_orow)) = o;
      }
    };
    // Drain the set-up loads HERE, on every path, with a wait the compiler's counter bookkeeping sees: a value loaded before
    // the loop and first used inside it otherwise gets a conservative `s_waitcnt vmcnt(0)` at that use in EVERY iteration
    // (the pending state survives the merge at the loop header), which would drain the row prefetch each time.
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)
    if (n_snap > 0) {
      const float *base = S_.prim_in + s_begin * prim_stride;
      static_for<PJ>([&](auto j_) {
        constexpr int j = decltype(j_)::value;
        static_for<2 * KT_X>([&](auto c_) {
          constexpr int c = decltype(c_)::value;
          pp[j][c] = *reinterpret_cast<const f32x4 *>(base + prow[j] + 16 * c);
        });
      });
    }
    WS_STAMP(0);
    lds_barrier();      // the X team has computed the secondary MLP of the first snapshot

    for (int k = 0; k <= n_snap; ++k) {
#ifdef UDS_WS_ABL_NO_P2
      if (false) {
#else
      if (k < n_snap) {
#endif
        // ---------------- P2: [prim | agg] @ Wbig -> hx, attention scalars ----------------
        const bool more = k + 1 < n_snap;
        const float *next = S_.prim_in + (s_begin + k + 1) * prim_stride;
        const float *secr = sec + (k & 1) * sec_buf;
        float *hxw = hx + (k & 1) * hx_buf;
        float *ssw = s_self + (k & 1) * a.p_cap, *snw = s_nbr + (k & 1) * a.p_cap;
        static_for<PJ>([&](auto j_) {
          constexpr int j = decltype(j_)::value;
          const int blk = yw + NY * j;
#ifndef UDS_WS_P2_BRANCH
          {     // every block unconditionally (rows clamped, stores predicated): one basic block, the scheduler interleaves the blocks (-2 %)
#else
          if (blk * 16 < n_prim) {
#endif
            bf16x8 dh[KT_B], dl[KT_B];
            static_for<KT_X>([&](auto t_) {
              constexpr int t = decltype(t_)::value;
              const f32x4 u0 = pp[j][2 * t], u1 = pp[j][2 * t + 1];
              split8(make_float4(u0[0], u0[1], u0[2], u0[3]), make_float4(u1[0], u1[1], u1[2], u1[3]), dh[t], dl[t]);
            });
            if (more)      // the registers are free (their values went through the split): the next snapshot's rows, a whole interval ahead
              static_for<2 * KT_X>([&](auto c_) {
                constexpr int c = decltype(c_)::value;
                pp[j][c] = *reinterpret_cast<const f32x4 *>(next + prow[j] + 16 * c);
              });
            const int lrow = blk * 16 + r16;
            const bool valid = lrow < n_prim;
            const int lr = min(lrow, n_prim - 1);
            f32x4 acc[MB_B];
#pragma unroll
            for (int m = 0; m < MB_B; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#ifndef UDS_WS_ABL_Y_NO_MFMA
#pragma unroll
            for (int t = 0; t < KT_X; ++t)
#pragma unroll
              for (int m = 0; m < MB_B; ++m) acc[m] = mfma3(wbh[t][m], wbl[t][m], dh[t], dl[t], acc[m]);
#else
#pragma unroll
            for (int m = 0; m < MB_B; ++m) acc[m] = f32x4{__builtin_bit_cast(float, (unsigned)dh[0][0] << 16), (float)m, __builtin_bit_cast(float, (unsigned)dl[1][0] << 16), 0.f};
#endif
            const unsigned locs = ag_locs[j];
            const f32x4 vals = ag_vals[j];
            float4 g0 = make_float4(0.f, 0.f, 0.f, 0.f), g1 = g0;   // this lane's 8 aggregate features (fragment shape)
            auto fma_row = [&](float wv, const float4 &u0, const float4 &u1) {
              g0.x = fmaf(wv, u0.x, g0.x); g0.y = fmaf(wv, u0.y, g0.y); g0.z = fmaf(wv, u0.z, g0.z); g0.w = fmaf(wv, u0.w, g0.w);
              g1.x = fmaf(wv, u1.x, g1.x); g1.y = fmaf(wv, u1.y, g1.y); g1.z = fmaf(wv, u1.z, g1.z); g1.w = fmaf(wv, u1.w, g1.w);
            };
            auto add2 = [&](unsigned la, float wa, unsigned lb, float wb) {      // two rows: four reads in flight, then the FMAs
              const float *ra = secr + la * SEC_STRIDE + 4 * qd, *rb = secr + lb * SEC_STRIDE + 4 * qd;
              const float4 a0 = *reinterpret_cast<const float4 *>(ra), a1 = *reinterpret_cast<const float4 *>(ra + 16);
              const float4 b0 = *reinterpret_cast<const float4 *>(rb), b1 = *reinterpret_cast<const float4 *>(rb + 16);
              fma_row(wa, a0, a1);
              fma_row(wb, b0, b1);
            };
#ifndef UDS_WS_ABL_Y_NO_AGG
            add2(locs & 0xffu, vals[0], (locs >> 8) & 0xffu, vals[1]);
            if (inc_width > 2) add2((locs >> 16) & 0xffu, vals[2], locs >> 24, vals[3]);
#else
            g0.x = vals[0]; g1.y = vals[1]; g0.z = __uint_as_float(locs);
#endif
            if (flags & ELL_FLAG_INC_OVF)      // rows with more than four incident rows (a junction of five or more conduits)
              for (int p = ovf_ptr[lr]; p < ovf_ptr[lr + 1]; ++p) {
                const float *ra = secr + ovf_loc[p] * SEC_STRIDE + 4 * qd;
                fma_row(ovf_val[p], *reinterpret_cast<const float4 *>(ra), *reinterpret_cast<const float4 *>(ra + 16));
              }
            split8(g0, g1, dh[KT_X], dl[KT_X]);
#ifndef UDS_WS_ABL_Y_NO_MFMA
#pragma unroll
            for (int m = 0; m < MB_B; ++m) acc[m] = mfma3(wbh[KT_X][m], wbl[KT_X][m], dh[KT_X], dl[KT_X], acc[m]);
#else
            acc[0][3] += __builtin_bit_cast(float, (unsigned)dh[KT_X][0] << 16) + __builtin_bit_cast(float, (unsigned)dl[KT_X][0] << 16);
#endif
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
            const float ps = quarters_sum(ps2.x + ps2.y);
            const float pn = quarters_sum(pn2.x + pn2.y);
#ifdef UDS_WS_ABL_Y_NO_HXW
            if (valid && ps == 12345.f) {
#else
            if (valid) {
#endif
              if (qd == 0) {
                ssw[lrow] = ps;
                snw[lrow] = pn;
              }
#pragma unroll
              for (int m = 0; m < MB_B; ++m)   // chunk index XOR (row & 7): the 8 lanes of a write group hit 8 different slots
                *reinterpret_cast<f32x4 *>(hxw + lrow * FUSED_D + (((4 * m + qd) ^ (lrow & 7)) << 2)) = acc[m];
            }
          }
        });
      }
      if (WS_YG > 0 && k >= 1) phase3_y(s_begin + k - 1, (k - 1) & 1);      // this wave's share of P3(s - 1), in the time it would wait at the barrier
      WS_STAMP(1);
      lds_barrier();      // hx / scores of snapshot k are in LDS (and the X team's sec rows of snapshot k + 1)
      WS_STAMP(2);
    }
    WS_DUMP();
#endif
  } else {
    // =========================== X team: P1 and P3 ===========================
#ifndef UDS_WS_NO_PRIO
    __builtin_amdgcn_s_setprio(UDS_WS_XPRIO);      // the longer instruction stream of the two, and the younger half of the workgroup (-2 %)
#endif
#ifndef UDS_WS_MFMA16
    // P1 on v_mfma_f32_32x32x16_bf16 too: half as many matrix instructions in this team's (issue-bound) stream.  32-row blocks
    // blk = xw + NX j; lane l: n = l & 31 the row, hf = l >> 5: piece i of the row = floats 16 (i >> 1) + 8 hf + 4 (i & 1).
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    constexpr int KS32 = FS / 16, SJ32 = (8 + NX - 1) / NX;      // 4 k-steps; 32-row secondary blocks per X wave (q_cap <= 256)
    const int n32 = lane & 31, hf = lane >> 5;
    bf16x8 w1h[KS32], w1l[KS32];      // 32 VGPRs
#pragma unroll
    for (int t = 0; t < KS32; ++t) {
      w1h[t] = __builtin_bit_cast(bf16x8, S_.w_small32[(t * 2 + 0) * 64 + lane]);
      w1l[t] = __builtin_bit_cast(bf16x8, S_.w_small32[(t * 2 + 1) * 64 + lane]);
    }
    unsigned srow[SJ32];              // BYTE offsets of this lane's rows
#pragma unroll
    for (int j = 0; j < SJ32; ++j) srow[j] = ((unsigned)sec_ids[min((xw + NX * j) * 32 + n32, n_sec - 1)] * FS + 8u * hf) * 4u;
    f32x4 sp[SJ32][2 * KS32];         // the secondary rows of the snapshot P1 multiplies next (asm loads: see the header)
    auto load_sec_at = [&](const float *base) __attribute__((always_inline)) {      // unconditional (rows are clamped): a fixed number of loads
      static_for<SJ32>([&](auto j_) {
        constexpr int j = decltype(j_)::value;
        ws_gld16<0>(sp[j][0], base, srow[j]);
        ws_gld16<16>(sp[j][1], base, srow[j]);
        ws_gld16<64>(sp[j][2], base, srow[j]);
        ws_gld16<80>(sp[j][3], base, srow[j]);
        ws_gld16<128>(sp[j][4], base, srow[j]);
        ws_gld16<144>(sp[j][5], base, srow[j]);
        ws_gld16<192>(sp[j][6], base, srow[j]);
        ws_gld16<208>(sp[j][7], base, srow[j]);
      });
    };
    auto load_sec = [&](int s) __attribute__((always_inline)) { load_sec_at(S_.sec_in + s * sec_stride); };
#define UDS_WS_RUNNING_PTRS 1
#else
    bf16x8 wsh[KT_S][MB_S], wsl[KT_S][MB_S];      // 32 VGPRs
#pragma unroll
    for (int t = 0; t < KT_S; ++t)
#pragma unroll
      for (int m = 0; m < MB_S; ++m) {
        wsh[t][m] = __builtin_bit_cast(bf16x8, S_.w_small[((t * MB_S + m) * 2 + 0) * 64 + lane]);
        wsl[t][m] = __builtin_bit_cast(bf16x8, S_.w_small[((t * MB_S + m) * 2 + 1) * 64 + lane]);
      }
    // BYTE offsets of the rows of the secondary blocks this wave multiplies (blk = xw + NX j); piece c = bytes 64 c + 16 qd
    unsigned srow[SJ];
#pragma unroll
    for (int j = 0; j < SJ; ++j) srow[j] = ((unsigned)sec_ids[min((xw + NX * j) * 16 + r16, n_sec - 1)] * FS + 4u * qd) * 4u;
    f32x4 sp[SJ][2 * KT_S];       // the secondary rows of the snapshot P1 multiplies next (asm loads: see the header)
    auto load_sec = [&](int s) __attribute__((always_inline)) {      // unconditional (rows are clamped): a fixed number of loads
      const float *base = S_.sec_in + s * sec_stride;
      static_for<SJ>([&](auto j_) {
        constexpr int j = decltype(j_)::value;
        ws_gld16<0>(sp[j][0], base, srow[j]);
        ws_gld16<64>(sp[j][1], base, srow[j]);
        ws_gld16<128>(sp[j][2], base, srow[j]);
        ws_gld16<192>(sp[j][3], base, srow[j]);
      });
    };
#endif
#ifndef UDS_WS_OCT_WD
#define UDS_WS_OCT_WD 4       // neighbour slots per step of the octet P3 (each step: UDS_WS_OCT_WD x 2 x octets ds_read_b128 in flight)
#endif
#ifndef UDS_WS_OCT_GRP
#define UDS_WS_OCT_GRP 1
#endif
#ifndef UDS_WS_P3_GROUPS
    // P3 in the OCTET layout: 8 lanes x 2 float4 per output row, 8 rows per wave-instruction (ro = lane >> 3, c8 = lane & 7 owns
    // features 4 c8 .. + 3 and 32 + 4 c8 .. + 3: the two 16-byte pieces sit 128 B apart in a swizzled hx row, ONE address).
    // Against 16 lanes x float4 (k_fused_tile, and this kernel under -DUDS_WS_P3_GROUPS) every instruction of the softmax
    // prologue and of the output epilogue covers twice the rows, and a neighbour slot costs 7 vector + 2 LDS instructions per
    // 8 rows instead of 10 + 2.  Lane c8 of a row scores neighbour slots c8 and (rows with more than eight neighbours: 7 %
    // of the line graph at the headline, none of the shipped networks) c8 + 8.  The LDS byte offsets of slots 0..7 are
    // static per tile: every lane keeps its own eight as 16-bit halves of four registers (p3_pk); slot K's weight reaches
    // the row's lanes by two bank-masked DPP row broadcasts.  The second slot set goes through ds_bpermute instead (the
    // LDS crossbar).  Rows are degree-sorted inside the tile and dealt in octets: unit u (0..3) of wave xw -> octet NX u + xw.
    const int ro = lane >> 3, c8 = lane & 7, hf8 = ro & 1;      // hf8: which 128-B half of a row this lane takes first (see phase3o)
    int p3_dmax[U];
    unsigned p3_pk[U][4];      // (the row's output row is re-read from the tile block per snapshot: registers are short)
    unsigned p3_jb4 = 0;          // this lane's own slot byte of the four octets, 8 bits each: the index of the score it gathers
    int n_st = 0;                 // output-store instructions this wave issues per snapshot (two per octet that has a row)
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = 8 * (NX * u + xw) + ro;
      const int ic = min(i, n_own - 1);
      int dmx;
      if (flags & ELL_FLAG_LONG_ROWS) {
        dmx = i < n_own ? adj_ptr[ic + 1] - adj_ptr[ic] : 0;
      } else {
        const unsigned long long m0 = __ballot(i < n_own && adj_b[ic * ELL_ADJ + c8] != 0xFF);
        const unsigned long long m1 = __ballot(i < n_own && adj_b[ic * ELL_ADJ + 8 + c8] != 0xFF);
        dmx = __builtin_popcountll((m0 >> (8 * ro)) & 0xffull) + __builtin_popcountll((m1 >> (8 * ro)) & 0xffull);
      }
#pragma unroll
      for (int o = 8; o < 64; o <<= 1) dmx = max(dmx, __shfl_xor(dmx, o));
      p3_dmax[u] = __builtin_amdgcn_readfirstlane(dmx);
#pragma unroll
      for (int k2 = 0; k2 < 4; ++k2) {
        unsigned pk = 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const unsigned jb = adj_b[ic * ELL_ADJ + 2 * k2 + h];
          const unsigned j = (i < n_own && jb != 0xFFu) ? jb : 0u;
          pk |= (j * (FUSED_D * 4) + (((j & 7) ^ c8) << 4)) << (16 * h);      // < 2^16: q_cap <= 256 rows of 256 B
        }
        p3_pk[u][k2] = pk;
      }
      p3_jb4 |= (i < n_own ? (unsigned)adj_b[i * ELL_ADJ + c8] : 0xFFu) << (8 * u);
      if (8 * (NX * u + xw) < n_own) n_st += 2;
    }
    const float *bias_l = attn + 2 * FUSED_D + FUSED_H + 4 * c8;      // + 32 hf8 / + 32 - 32 hf8 for the lane's first / second piece      // the output bias, read back per snapshot (registers are short)
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): drain the set-up loads on every path (see the Y team's note)
#else
    // P3 rows: degree-sorted inside the tile and dealt in 4-row groups: trip t (0, 1), unit u (0..3) -> group NX (2 u + t) + xw,
    // so both trips and all waves get the same mix of degrees, in descending order.  Static per tile: the (wave-uniform)
    // largest degree of every group in SGPRs, this lane's neighbour byte of every group in VGPRs.
    int p3_dmax[2][U];
    unsigned p3_jb[2][U];
    int p3_joff[2][U], p3_orow[2][U], p3_ic[2][U];      // static per tile: neighbour's hx offset (swizzled), output row offset, own row
    int n_st = 0;                 // output-store instructions this wave issues per snapshot (one per group that has a row)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = 4 * (WS_YG + NX * (2 * u + t) + xw) + rs;
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
        p3_dmax[t][u] = __builtin_amdgcn_readfirstlane(dmx);
        p3_jb[t][u] = adj_b[ic * ELL_ADJ + c16];
        {
          const bool has_ = i < n_own && p3_jb[t][u] != 0xFFu;
          const int jn_ = has_ ? (int)p3_jb[t][u] : 0;
          p3_joff[t][u] = jn_ * (FUSED_D * 4) + ((jn_ & 7) << 4);
          p3_orow[t][u] = prim_ids[ic] * FUSED_D + 4 * c16;
          p3_ic[t][u] = ic;
        }
        if (4 * (WS_YG + NX * (2 * u + t) + xw) < n_own) ++n_st;
      }
    f32x4 bo = f32x4{0.f, 0.f, 0.f, 0.f};
    if (S_.b_out) bo = *reinterpret_cast<const f32x4 *>(S_.b_out + 4 * c16);
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): drain the set-up loads on every path (see the Y team's note)
#endif

#ifndef UDS_WS_MFMA16
    // ---------------- P1: secondary MLP, registers -> MFMA (32x32x16) -> sec ----------------
    auto phase1 = [&](int buf) __attribute__((always_inline)) {
      float *secw = sec + buf * sec_buf;
      static_for<SJ32>([&](auto j_) {
        constexpr int j = decltype(j_)::value;
        const int blk = xw + NX * j;
        if (blk * 32 < n_sec) {
          f32x16 acc;
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {      // bias of this lane's features (r & 3) + 8 (r >> 2) + 4 hf
            const f32x4 b4 = *reinterpret_cast<const f32x4 *>(attn + 2 * FUSED_D + 8 * gq + 4 * hf);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[4 * gq + q] = b4[q];
          }
          static_for<KS32>([&](auto t_) {
            constexpr int t = decltype(t_)::value;
            const f32x4 u0 = sp[j][2 * t], u1 = sp[j][2 * t + 1];
            bf16x8 dh, dl;
            split8(make_float4(u0[0], u0[1], u0[2], u0[3]), make_float4(u1[0], u1[1], u1[2], u1[3]), dh, dl);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1h[t], dl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1l[t], dh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1h[t], dh, acc, 0, 0, 0);
          });
          const int lrow = blk * 32 + n32;
          if (lrow < n_sec) {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
              f32x4 o;
#pragma unroll
              for (int q = 0; q < 4; ++q) o[q] = fused_act<ACT>(acc[4 * gq + q], a.act);
              *reinterpret_cast<f32x4 *>(secw + lrow * SEC_STRIDE + 8 * gq + 4 * hf) = o;
            }
          }
        }
      });
    };
    auto pin_rows = [&]() __attribute__((always_inline)) {
      static_for<SJ32>([&](auto j_) {
        constexpr int j = decltype(j_)::value;
        ws_pin4(sp[j][0], sp[j][1], sp[j][2], sp[j][3]);
        ws_pin4(sp[j][4], sp[j][5], sp[j][6], sp[j][7]);
      });
    };
#else
    // ---------------- P1: secondary MLP, registers -> MFMA -> sec ----------------
    auto phase1 = [&](int buf) __attribute__((always_inline)) {
      float *secw = sec + buf * sec_buf;
      static_for<SJ>([&](auto j_) {
        constexpr int j = decltype(j_)::value;
        const int blk = xw + NX * j;
        if (blk * 16 < n_sec) {
          bf16x8 dh[KT_S], dl[KT_S];
          static_for<KT_S>([&](auto t_) {
            constexpr int t = decltype(t_)::value;
            const f32x4 u0 = sp[j][2 * t], u1 = sp[j][2 * t + 1];
            split8(make_float4(u0[0], u0[1], u0[2], u0[3]), make_float4(u1[0], u1[1], u1[2], u1[3]), dh[t], dl[t]);
          });
          const int lrow = blk * 16 + r16;
          f32x4 acc[MB_S];
#pragma unroll
          for (int m = 0; m < MB_S; ++m) acc[m] = *reinterpret_cast<const f32x4 *>(attn + 2 * FUSED_D + 16 * m + 4 * qd);
#pragma unroll
          for (int t = 0; t < KT_S; ++t)
#pragma unroll
            for (int m = 0; m < MB_S; ++m) acc[m] = mfma3(wsh[t][m], wsl[t][m], dh[t], dl[t], acc[m]);
          if (lrow < n_sec) {
#pragma unroll
            for (int m = 0; m < MB_S; ++m) {
              f32x4 o;
#pragma unroll
              for (int q = 0; q < 4; ++q) o[q] = fused_act<ACT>(acc[m][q], a.act);
              *reinterpret_cast<f32x4 *>(secw + lrow * SEC_STRIDE + 16 * m + 4 * qd) = o;
            }
          }
        }
      });
    };
    // tie the row registers to a point of the program: nothing that reads them is scheduled above, nothing that writes them below
    auto pin_rows = [&]() __attribute__((always_inline)) {
      static_for<SJ>([&](auto j_) {
        constexpr int j = decltype(j_)::value;
        ws_pin4(sp[j][0], sp[j][1], sp[j][2], sp[j][3]);
      });
    };

#endif

    // first snapshot: rows -> registers -> P1 -> sec[0]; the second snapshot's rows into the registers
    if (n_snap > 0) {
      load_sec(s_begin);
      ws_vmcnt(0);
      pin_rows();
      phase1(0);
      if (n_snap > 1) load_sec(s_begin + 1);
    }
    WS_STAMP(0);
    lds_barrier();

#ifndef UDS_WS_P3_GROUPS
    // ---------------- P3 (octets): segmented softmax + neighbour sum -> HBM, the wave's four octets in one pass ----------------
    auto phase3o = [&](float *out_snap, int buf) __attribute__((always_inline)) {      // out_snap: the snapshot's (n_prim_glob, 64) output
      const float *hxr = hx + buf * hx_buf;
      const float *ssr = s_self + buf * a.p_cap, *snr = s_nbr + buf * a.p_cap;
      int d0 = p3_dmax[0], d1 = p3_dmax[1], d2 = p3_dmax[2], d3 = p3_dmax[3];
      asm volatile("" : "+s"(d0), "+s"(d1), "+s"(d2), "+s"(d3));      // keep the slot tests on the scalar unit
      const int e3 = d3, e2 = max(e3, d2), e1 = max(e2, d1), e0 = max(e1, d0);
      if (e0 == 0) return;            // this wave has no row in the tile

      float ss[U], lg0[U], lg1[U], w0[U], w1[U], den[U];
      bool ok[U], has0[U], has1[U];
      unsigned jb0[U], jb1[U];
      int joff1[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        ok[u] = 8 * (NX * u + xw) + ro < n_own;
        // every LDS read of this prologue is unconditional (a row past n_own reads some other word of the tile block, a missing
        // slot the score of row 255; the selects below discard both): under a condition the compiler branches around each read
        // and waits for it alone
        jb0[u] = (p3_jb4 >> (8 * u)) & 0xFFu;      // static per tile (one packed register): no LDS round trip in front of the score gather
        ss[u] = ssr[8 * (NX * u + xw) + ro];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) has0[u] = ok[u] && jb0[u] != 0xFFu;
      f32x4 acc0[U], acc1[U];
#pragma unroll
      for (int u = 0; u < U; ++u) acc0[u] = acc1[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (e0 <= 16) {
        const bool ext = e0 > 8;      // some row of this wave has more than eight neighbours: the second slot set takes part
        float sn[U];
#pragma unroll
        for (int u = 0; u < U; ++u) sn[u] = snr[jb0[u]];
        asm volatile("" : "+v"(sn[0]), "+v"(sn[1]), "+v"(sn[2]), "+v"(sn[3]));
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const float sv = ss[u] + sn[u];
          lg0[u] = has0[u] ? fmaxf(sv, 0.2f * sv) : -INFINITY;      // leaky_relu(0.2)
        }
        float mx[U];
        row8_max4(lg0, mx);
        if (ext) {
          float m1[U];
#pragma unroll
          for (int u = 0; u < U; ++u) jb1[u] = adj_b[(8 * (NX * u + xw) + ro) * ELL_ADJ + 8 + c8];
          asm volatile("" : "+v"(jb1[0]), "+v"(jb1[1]), "+v"(jb1[2]), "+v"(jb1[3]));
#pragma unroll
          for (int u = 0; u < U; ++u) sn[u] = snr[jb1[u]];
          asm volatile("" : "+v"(sn[0]), "+v"(sn[1]), "+v"(sn[2]), "+v"(sn[3]));
#pragma unroll
          for (int u = 0; u < U; ++u) {
            has1[u] = ok[u] && jb1[u] != 0xFFu;
            const int j1 = has1[u] ? (int)jb1[u] : 0;
            joff1[u] = j1 * (FUSED_D * 4) + ((j1 & 7) << 4);
            const float sv = ss[u] + sn[u];
            lg1[u] = has1[u] ? fmaxf(sv, 0.2f * sv) : -INFINITY;
          }
          row8_max4(lg1, m1);
#pragma unroll
          for (int u = 0; u < U; ++u) mx[u] = fmaxf(mx[u], m1[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) w0[u] = has0[u] ? __builtin_amdgcn_exp2f(lg0[u] - mx[u]) : 0.f;
        row8_sum4(w0, den);
        if (ext) {
          float d1s[U];
#pragma unroll
          for (int u = 0; u < U; ++u) w1[u] = has1[u] ? __builtin_amdgcn_exp2f(lg1[u] - mx[u]) : 0.f;
          row8_sum4(w1, d1s);
#pragma unroll
          for (int u = 0; u < U; ++u) den[u] += d1s[u];
        }
        typedef const __attribute__((address_space(3))) f32x4 *lds_f4;
        // the first piece a lane reads is the 128-B half `hf8` of the neighbour's row, the second the other one: the 16 lanes
        // ds_read_b128 serves per LDS cycle hold two rows' low-numbered and two rows' high-numbered lanes, and rows of
        // opposite parity among them -- both halves of the 256-B bank row are in use, as in the 16-lane layout
        const unsigned hxa0 = lds_addr(hxr) + 128 * hf8, hxa1 = lds_addr(hxr) + 128 - 128 * hf8;
        const int cx = c8 << 4;                        // this lane's first 16-byte piece as a byte offset inside a row
        const int bp = (lane & 56) << 2;               // ds_bpermute address of lane 0 of this row's octet
        using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
        using I3 = std::integral_constant<int, 3>; using I4 = std::integral_constant<int, 4>;
        // slots K .. K + Wd - 1 (0 <= K < 8) of slot set SET for the first A octets of the wave
        auto step = [&](auto SET_, auto K_, auto A_) __attribute__((always_inline)) {
          constexpr int SET = decltype(SET_)::value, K = decltype(K_)::value, A = decltype(A_)::value,
                        Wd = (8 - K < UDS_WS_OCT_WD ? 8 - K : UDS_WS_OCT_WD);
          constexpr int G = UDS_WS_OCT_GRP;            // octets per batch of reads (registers: 8 x G x Wd for the pieces)
          static_for<(A + G - 1) / G>([&](auto g_) {
            constexpr int u0 = decltype(g_)::value * G, GA = (A - u0 < G ? A - u0 : G);
            f32x4 h0[GA][Wd], h1[GA][Wd];
            static_for<GA>([&](auto u_) {
              constexpr int ul = decltype(u_)::value, u = u0 + ul;
              static_for<Wd>([&](auto i_) {
                constexpr int i = decltype(i_)::value, k = K + i;
                unsigned ad = 0;
                if (SET) ad = (unsigned)__builtin_amdgcn_ds_bpermute(bp + 4 * k, joff1[u]) ^ cx;
                unsigned ad0, ad1;
                if (SET) {
                  ad0 = ad + hxa0;
                  ad1 = ad + hxa1;
                } else {                 // half k & 1 of the packed pair + the buffer's base in one instruction (ws_add_half)
                  ad0 = ws_add_half<k & 1>(p3_pk[u][k >> 1], hxa0);
                  ad1 = ws_add_half<k & 1>(p3_pk[u][k >> 1], hxa1);
                }
                h0[ul][i] = *(lds_f4)(uintptr_t)ad0;
                h1[ul][i] = *(lds_f4)(uintptr_t)ad1;
              });
            });
            static_for<GA>([&](auto u_) {
              constexpr int ul = decltype(u_)::value, u = u0 + ul;
              static_for<Wd>([&](auto i_) {
                constexpr int i = decltype(i_)::value, k = K + i;
                float wv;
                if (SET) wv = __int_as_float(__builtin_amdgcn_ds_bpermute(bp + 4 * k, __float_as_int(w1[u])));
                else {     // lane k of the octet: row broadcast of lane k into the low eight lanes of every 16, of lane 8 + k into the high eight
                  const int wi = __float_as_int(w0[u]);
                  int t = __builtin_amdgcn_update_dpp(wi, wi, 0x150 + k, 0xf, 0xf, true);      // every lane written: no `old` to set up
                  t = __builtin_amdgcn_update_dpp(t, wi, 0x150 + 8 + k, 0xf, 0xc, false);
                  wv = __int_as_float(t);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  acc0[u][q] = fmaf(wv, h0[ul][i][q], acc0[u][q]);
                  acc1[u][q] = fmaf(wv, h1[ul][i][q], acc1[u][q]);
                }
              });
            });
          });
        };
        static_for<(8 + UDS_WS_OCT_WD - 1) / UDS_WS_OCT_WD>([&](auto t_) {                  // slot set 0: slots 0..7, UDS_WS_OCT_WD per step
          constexpr int K = decltype(t_)::value * UDS_WS_OCT_WD;
          using IK = std::integral_constant<int, K>;
          using S0 = std::integral_constant<int, 0>;
          if (K < e3) step(S0{}, IK{}, I4{});
          else if (K < e2) step(S0{}, IK{}, I3{});
          else if (K < e1) step(S0{}, IK{}, I2{});
          else if (K < e0) step(S0{}, IK{}, I1{});
        });
        if (ext)
          static_for<(8 + UDS_WS_OCT_WD - 1) / UDS_WS_OCT_WD>([&](auto t_) {                // slot set 1: slots 8..15
            constexpr int K = decltype(t_)::value * UDS_WS_OCT_WD;
            using IK = std::integral_constant<int, K>;
            using S1 = std::integral_constant<int, 1>;
            if (K + 8 < e3) step(S1{}, IK{}, I4{});
            else if (K + 8 < e2) step(S1{}, IK{}, I3{});
            else if (K + 8 < e1) step(S1{}, IK{}, I2{});
            else if (K + 8 < e0) step(S1{}, IK{}, I1{});
          });
      } else {   // some row has more than 16 neighbours: every lane walks its row's whole list (tiles with ELL_FLAG_LONG_ROWS)
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int i = 8 * (NX * u + xw) + ro;
          const int b0 = i < n_own ? adj_ptr[i] : 0;
          const int dg = i < n_own ? adj_ptr[i + 1] - b0 : 0;
          float mx = -INFINITY;
          for (int p = b0; p < b0 + dg; ++p) mx = fmaxf(mx, leaky02(ss[u] + snr[adj_loc[p]]));
          den[u] = 0.f;
          for (int p = b0; p < b0 + dg; ++p) {
            const int jj = adj_loc[p];
            const float wv = __builtin_amdgcn_exp2f(leaky02(ss[u] + snr[jj]) - mx);
            const f32x4 v0 = *reinterpret_cast<const f32x4 *>(hxr + jj * FUSED_D + 32 * hf8 + ((c8 ^ (jj & 7)) << 2));
            const f32x4 v1 = *reinterpret_cast<const f32x4 *>(hxr + jj * FUSED_D + 32 - 32 * hf8 + ((c8 ^ (jj & 7)) << 2));
            den[u] += wv;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              acc0[u][q] = fmaf(wv, v0[q], acc0[u][q]);
              acc1[u][q] = fmaf(wv, v1[q], acc1[u][q]);
            }
          }
        }
      }
      // Epilogue.  The six LDS reads it needs (bias pieces, the four output row ids) are issued together and waited for once, and the
      // snapshot's output base is formed once: written the plain way the compiler put an id read + its wait, a reload of the output
      // pointer from the kernel arguments and a 64-bit multiply inside each of the four exec-masked store branches -- five dependent
      // round trips per snapshot.
      f32x4 bo0 = *reinterpret_cast<const f32x4 *>(bias_l + 32 * hf8), bo1 = *reinterpret_cast<const f32x4 *>(bias_l + 32 - 32 * hf8);
      int pid[U];
#pragma unroll
      for (int u = 0; u < U; ++u) pid[u] = prim_ids[8 * (NX * u + xw) + ro];      // rows past n_own: some other word of the tile block, unused
      asm volatile("" : "+v"(pid[0]), "+v"(pid[1]), "+v"(pid[2]), "+v"(pid[3]), "+v"(bo0), "+v"(bo1));
      float *out_s = out_snap + (4 * c8 + 32 * hf8);
      const int o1_off = 32 - 64 * hf8;      // the second piece relative to the first
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (ok[u]) {
          const float inv = __builtin_amdgcn_rcpf(den[u]);
          f32x4 o0, o1;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            o0[q] = fused_act<ACT>(fmaf(acc0[u][q], inv, bo0[q]), a.act);
            o1[q] = fused_act<ACT>(fmaf(acc1[u][q], inv, bo1[q]), a.act);
          }
          float *orow = out_s + (int64_t)pid[u] * FUSED_D;
          *reinterpret_cast<f32x4 *>(orow) = o0;
          *reinterpret_cast<f32x4 *>(orow + o1_off) = o1;
        }
      }
    };
#else
    // ---------------- P3: segmented softmax + neighbour sum -> HBM, one trip = 4 row groups of this wave ----------------
    auto phase3 = [&](int TR, int s, int buf) __attribute__((always_inline)) {      // TR: trip 0 / 1 (a constant after inlining, or the
                                                                                        // counter of a real loop under UDS_WS_P3_LOOP: half the code)
#define P3S(arr, u) (TR ? arr[1][u] : arr[0][u])
      const float *hxr = hx + buf * hx_buf;
      const float *ssr = s_self + buf * a.p_cap, *snr = s_nbr + buf * a.p_cap;
      int jn[U], dmax[U], orow[U];
      float ss[U], lg[U], wgt[U], den[U];
      bool ok[U], has[U];
      int dm = 0;
#pragma unroll
      for (int u = 0; u < U; ++u) {   // unconditional loads on clamped indices: the four groups' reads overlap
        const int i = 4 * (WS_YG + NX * (2 * u + TR) + xw) + rs;
        const int ic = min(i, n_own - 1);
        ok[u] = i < n_own;
        has[u] = ok[u] && P3S(p3_jb, u) != 0xFFu;      // neighbour slot c16 of this row exists (slots fill from 0)
        jn[u] = has[u] ? (int)P3S(p3_jb, u) : 0;
        dmax[u] = P3S(p3_dmax, u);
        ss[u] = ssr[P3S(p3_ic, u)];
        orow[u] = P3S(p3_orow, u);
        dm = max(dm, dmax[u]);
      }
      if (dm == 0) return;            // no row of this trip exists
      f32x4 acc[U];
#pragma unroll
      for (int u = 0; u < U; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (dm <= 16) {
        int d0 = dmax[0], d1 = dmax[1], d2 = dmax[2], d3 = dmax[3];
        asm volatile("" : "+s"(d0), "+s"(d1), "+s"(d2), "+s"(d3));      // keep the slot tests on the scalar unit (see k_fused_tile)
        const int e3 = d3, e2 = max(e3, d2), e1 = max(e2, d1), e0 = max(e1, d0);
        const char *hxb = reinterpret_cast<const char *>(hxr);
        const int cx = c16 << 4;
        int joff[U];      // byte offset of the neighbour's hx row with its swizzle key in bits 4-6: j*256 + (j&7)*16
#pragma unroll
        for (int u = 0; u < U; ++u) {
          joff[u] = P3S(p3_joff, u);
          const float sv = ss[u] + snr[jn[u]];
          const float sc = fmaxf(sv, 0.2f * sv);      // leaky_relu(0.2)
          lg[u] = has[u] ? sc : -INFINITY;
        }
        using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
        using I3 = std::integral_constant<int, 3>; using I4 = std::integral_constant<int, 4>;
        constexpr int R = WS_PRE > 0 ? WS_PRE : 1;
        f32x4 hh0[U][R];
        auto pre_read = [&](auto K_, auto A_) __attribute__((always_inline)) {      // slot K of the A groups that still have one
          constexpr int K = decltype(K_)::value, A = decltype(A_)::value;
          static_for<A>([&](auto u_) {
            constexpr int u = decltype(u_)::value;
            hh0[u][K] = *reinterpret_cast<const f32x4 *>(hxb + (row16_bcast<K>(joff[u]) ^ cx));
          });
        };
        if constexpr (WS_PRE > 0)
          static_for<WS_PRE>([&](auto K_) {
            constexpr int K = decltype(K_)::value;
            if (K < e3) pre_read(K_, I4{});
            else if (K < e2) pre_read(K_, I3{});
            else if (K < e1) pre_read(K_, I2{});
            else if (K < e0) pre_read(K_, I1{});
          });
        static_assert(U == 4, "the four-at-once reductions below");
        float mx[U];
        row16_max4(lg, mx);
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const float ex = __builtin_amdgcn_exp2f(lg[u] - mx[u]);
          wgt[u] = has[u] ? ex : 0.f;
        }
        row16_sum4(wgt, den);
        auto pre_fma = [&](auto K_, auto A_) __attribute__((always_inline)) {
          constexpr int K = decltype(K_)::value, A = decltype(A_)::value;
          static_for<A>([&](auto u_) {
            constexpr int u = decltype(u_)::value;
            const float wv = __int_as_float(row16_bcast<K>(__float_as_int(wgt[u])));
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[u][q] = fmaf(wv, hh0[u][K][q], acc[u][q]);
          });
        };
        if constexpr (WS_PRE > 0)
          static_for<WS_PRE>([&](auto K_) {
            constexpr int K = decltype(K_)::value;
            if (K < e3) pre_fma(K_, I4{});
            else if (K < e2) pre_fma(K_, I3{});
            else if (K < e1) pre_fma(K_, I2{});
            else if (K < e0) pre_fma(K_, I1{});
          });
        auto step = [&](auto K_, auto A_) __attribute__((always_inline)) {
          constexpr int K = decltype(K_)::value, A = decltype(A_)::value, Wd = WS_SLOTS;
          f32x4 hh[A][Wd];
#pragma unroll
          for (int u = 0; u < A; ++u)
            static_for<Wd>([&](auto i_) {
              constexpr int i = decltype(i_)::value;
              hh[u][i] = *reinterpret_cast<const f32x4 *>(hxb + (row16_bcast<K + i>(joff[u]) ^ cx));
            });
#pragma unroll
          for (int u = 0; u < A; ++u)
            static_for<Wd>([&](auto i_) {
              constexpr int i = decltype(i_)::value;
              const float wv = __int_as_float(row16_bcast<K + i>(__float_as_int(wgt[u])));
#pragma unroll
              for (int q = 0; q < 4; ++q) acc[u][q] = fmaf(wv, hh[u][i][q], acc[u][q]);
            });
        };
        bool more = e0 > WS_PRE;
        static_for<(16 - WS_PRE) / WS_SLOTS>([&](auto t_) {
          constexpr int K = WS_PRE + decltype(t_)::value * WS_SLOTS;
          if (more) {
            using IK = std::integral_constant<int, K>;
#ifdef UDS_WS_P3_FLAT
            step(IK{}, I4{});      // experiment: all four groups at every step (exhausted groups read row 0 with weight 0): no if-chain, a quarter of the code
#else
            if (K < e3) step(IK{}, I4{});
            else if (K < e2) step(IK{}, I3{});
            else if (K < e1) step(IK{}, I2{});
            else step(IK{}, I1{});
#endif
            more = e0 > K + WS_SLOTS;
          }
        });
      } else {   // some row has more than 16 neighbours: every lane walks its row's whole list (tiles with ELL_FLAG_LONG_ROWS)
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int i = 4 * (WS_YG + NX * (2 * u + TR) + xw) + rs;
          const int b0 = i < n_own ? adj_ptr[i] : 0;
          const int dg = i < n_own ? adj_ptr[i + 1] - b0 : 0;
          float mx = -INFINITY;
          for (int p = b0; p < b0 + dg; ++p) mx = fmaxf(mx, leaky02(ss[u] + snr[adj_loc[p]]));
          den[u] = 0.f;
          for (int p = b0; p < b0 + dg; ++p) {
            const int jj = adj_loc[p];
            const float wv = __builtin_amdgcn_exp2f(leaky02(ss[u] + snr[jj]) - mx);
            const f32x4 hv = *reinterpret_cast<const f32x4 *>(hxr + jj * FUSED_D + ((c16 ^ (jj & 7)) << 2));
            den[u] += wv;
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[u][q] = fmaf(wv, hv[q], acc[u][q]);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (ok[u]) {
          const float inv = __builtin_amdgcn_rcpf(den[u]);
          f32x4 o;
#pragma unroll
          for (int q = 0; q < 4; ++q) o[q] = fused_act<ACT>(fmaf(acc[u][q], inv, bo[q]), a.act);
          *reinterpret_cast<f32x4 *>(S_.out + ((int64_t)s * S_.n_prim_glob * FUSED_D + orow[u])) = o;
        }
      }
    };

#endif
#ifdef UDS_WS_RUNNING_PTRS
    // the base of the rows requested next and of the snapshot P3 writes next, as running pointers: formed from the kernel arguments
    // inside the loop, each costs a scalar load + s_waitcnt lgkmcnt(0) per interval -- a wait that also drains the wave's LDS queue
    const float *sec_next = S_.sec_in + (int64_t)(s_begin + 2) * sec_stride;
    float *out_next = S_.out + (int64_t)s_begin * S_.n_prim_glob * FUSED_D;
    const int64_t out_stride = (int64_t)S_.n_prim_glob * FUSED_D;
#endif
    for (int k = 0; k <= n_snap; ++k) {
      const int s = s_begin + k;
      if (k + 1 < n_snap) {
        // the rows of snapshot k + 1 were requested in interval k - 1, right after its P1; the only younger vector-memory
        // operations of this wave are the output stores of THAT interval's P3 -- which exists from interval 1 on.  (k >= 1
        // here was a race found by the C3-size repeatability test: interval 0 stores nothing, so in interval 1 vmcnt(n_st)
        // let n_st of the sixteen row loads themselves stay in flight.)
        ws_vmcnt(k >= 2 ? n_st : 0);
        pin_rows();
#ifndef UDS_WS_ABL_NO_P1
        phase1((k + 1) & 1);
#endif
#ifdef UDS_WS_RUNNING_PTRS
        if (k + 2 < n_snap) load_sec_at(sec_next);
        sec_next += sec_stride;
#else
        if (k + 2 < n_snap) load_sec(s + 2);
#endif
      }
      WS_STAMP(3);
#ifndef UDS_WS_ABL_NO_P3
      if (k >= 1) {
#ifndef UDS_WS_P3_GROUPS
#ifdef UDS_WS_RUNNING_PTRS
        phase3o(out_next, (k - 1) & 1);
        out_next += out_stride;
#else
        phase3o(S_.out + (int64_t)(s - 1) * S_.n_prim_glob * FUSED_D, (k - 1) & 1);
#endif
        WS_STAMP(4);
#elif defined(UDS_WS_P3_LOOP)
#pragma nounroll
        for (int tr = 0; tr < 2; ++tr) phase3(tr, s - 1, (k - 1) & 1);
#else
        phase3(0, s - 1, (k - 1) & 1);
        WS_STAMP(4);
#ifndef UDS_WS_ABL_HALF_P3
        phase3(1, s - 1, (k - 1) & 1);
#endif
        WS_STAMP(5);
#endif
      }
#endif
      WS_STAMP(1);
      lds_barrier();
      WS_STAMP(2);
    }
    WS_DUMP();
  }
}

}  // namespace uds
