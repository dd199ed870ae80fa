// libuds_hip.so -- C ABI + launchers (gfx950 only).  Kernels live in kernels_*.hpp.
// Interface contract and the reference call sites each entry replaces: include/uds_hip.h.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/uds_hip.h"
#include <cstdlib>

#include "kernels_dense.hpp"
#include "kernels_sparse.hpp"
#include "kernels_backward.hpp"
#include "kernels_fused.hpp"
#include "kernels_fused_ws.hpp"
#include "kernels_rowgemm.hpp"
#include "kernels_wgrad.hpp"
#include "kernels_conv_stream.hpp"
#include "kernels_fused128.hpp"
#include "kernels_gemm.hpp"
#include "kernels_recurrent.hpp"
#include "kernels_recurrent_bwd.hpp"
#include "tile_plan.hpp"

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define UDS_HIP_TRY(call)                                                              \
  do {                                                                                 \
    hipError_t e_ = (call);                                                            \
    if (e_ != hipSuccess) return fail(UDS_EHIP, "%s -> %s", #call, hipGetErrorString(e_)); \
  } while (0)

#define UDS_REQUIRE(cond, ...) \
  do {                         \
    if (!(cond)) return fail(UDS_EINVAL, __VA_ARGS__); \
  } while (0)

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline int64_t align4(int64_t floats) { return (floats + 3) & ~int64_t(3); }  // keep 16-B carve offsets

}  // namespace

struct uds_csr {
  int64_t n_rows = 0, n_cols = 0, nnz = 0;
  int32_t max_degree = 0;
  int32_t *d_rowptr = nullptr, *d_col = nullptr, *d_order = nullptr, *d_rowidx = nullptr;   // rowidx: row of every entry
  std::vector<int32_t> h_order;
  uds::HostCsr host;   // kept for the tile planner
};

struct uds_tile_plan {
  uds::NetworkPlan plan;
};

// One tile plan per input width class: the DMA stage holds raw rows, so wider rows need smaller tiles.
struct uds_plan_slot {
  bool built = false;             // planning was attempted
  bool ok = false;                // a tile plan that fits the LDS budget exists
  int fp = 0, fs = 0;             // primary / secondary row widths (floats) of the kernel variant the plan was sized for
  uds::NetworkPlan plan;
  int32_t *d_hdr = nullptr, *d_pool = nullptr;
  int32_t *d_hdr_side[2] = {nullptr, nullptr};   // headers of one side's tiles only (same pool): single-side launches
  // k_fused_tile: header + metadata of every tile as one fixed-stride block (tile_block_ints(meta_cap) ints), so a workgroup
  // fetches both in ONE round trip without knowing the header first; one array per tile list
  int32_t *d_blocks = nullptr, *d_blocks_side[2] = {nullptr, nullptr};
  int blk_cap = 0;                // ints per tile block of k_fused_tile (tile_plan.hpp: ell_block_cap)
  int64_t lds_bytes = 0;
};

struct uds_network {
  const uds_csr *adj = nullptr, *edge_adj = nullptr, *inc_n = nullptr, *inc_e = nullptr;
  uds_plan_slot slot[7];          // d = 64 variants <FP, FS>: [0] <64,64>, [1] <64,96>, [2] <96,64>, [3] <96,96>; d = 128: [4] <128,128>, [5] <128,64>, [6] <64,128>
};

namespace {

constexpr int64_t FUSED_LDS_BUDGET = 160 * 1024;   // one 8-wave workgroup per CU owns the whole 160 KiB LDS
constexpr int WS_BIG32_OFF = 2 * (768 + 2048), WS_BIG32_LEN = 6 * 2 * 2 * 64;      // uint4: behind the four 16x16x32-order kernels of a d = 64 layer
constexpr int WS_SMALL32_OFF = WS_BIG32_OFF + 2 * WS_BIG32_LEN, WS_SMALL32_LEN = 4 * 1 * 2 * 64;
constexpr int64_t PACKED_WEIGHT_FLOATS = 2 * (2048 + 6144) * 4;   // both sides at d = 128 (128->64 and 192->128 kernels): uint4 = 4 floats

inline int slot_index(int fp, int fs) {
  if (fp == 128 || fs == 128) return fp == 128 ? (fs == 128 ? 4 : 5) : 6;      // the d = 128 kernel's variants
  return (fp > 64 ? 2 : 0) + (fs > 64 ? 1 : 0);
}

// Tile plan for the kernel variant <fp, fs> (primary / secondary input rows of fp / fs floats: they set the size of the
// DMA stage) under the LDS budget: fix the footprint limits (primary / secondary rows staged per tile, multiples of 16)
// first; the planner then fills them (tile_plan.hpp: merge_clusters).
bool plan_network(const uds::HostCsr &adj, const uds::HostCsr &eadj, const uds::HostCsr &inc_n, const uds::HostCsr &inc_e,
                  int fp, int fs, uds::NetworkPlan &out, int64_t &lds, int &blk_cap) {
  // candidate (p_limit, q_limit) pairs, largest first; meta is bounded by the limits (checked after planning)
  const int cand[][2] = {{128, 208}, {128, 192}, {128, 176}, {128, 160}, {128, 144}, {112, 160}, {112, 144}, {96, 144}, {96, 128},
                         {80, 128}, {64, 96}, {48, 64}, {32, 48}, {16, 32}};
  for (const auto &c : cand) {
    int p_lim = c[0], q_lim = c[1];
#ifdef UDS_KNOBS
    if (const char *ov = std::getenv("UDS_PLIM")) p_lim = std::min(p_lim, std::atoi(ov));      // experiment builds: smaller tiles
    if (const char *ov = std::getenv("UDS_QLIM")) q_lim = std::min(q_lim, std::atoi(ov));
#endif
    if (uds::fused_lds_bytes(p_lim, q_lim, 0, uds::FUSED_H, uds::FUSED_D, fp, fs) > FUSED_LDS_BUDGET) continue;
    const int t = std::min(p_lim, 4 * uds::FUSED_WAVES * uds::FUSED_U);        // own rows: P3 covers a tile in one trip
    out = uds::build_network_plan(adj, eadj, inc_n, inc_e, t, t, p_lim, q_lim);
    blk_cap = uds::ell_block_cap(out.hdr, out.pool, out.n_tiles);      // the kernel's tile blocks: fixed-width index lists
    if (blk_cap < 0) continue;                                           // a tile beyond the byte-wide local indices (a hub row)
    lds = uds::fused_lds_bytes(out.p_cap, out.q_cap, blk_cap, uds::FUSED_H, uds::FUSED_D, fp, fs);
    if (lds <= FUSED_LDS_BUDGET && out.p_cap <= 4 * uds::FUSED_WAVES * uds::FUSED_U) return true;   // P3 covers a tile in one trip
  }
  return false;
}

// Tile plan of the d = 128 kernel (k_fused128): <= 64 own / primary rows, secondary rows as the LDS allows.
bool plan_network128(const uds::HostCsr &adj, const uds::HostCsr &eadj, const uds::HostCsr &inc_n, const uds::HostCsr &inc_e, int fp, int fs,
                     uds::NetworkPlan &out, int64_t &lds) {
  const int cand[][2] = {{64, 128}, {64, 112}, {64, 96}, {64, 80}, {64, 64}, {48, 64}, {32, 48}, {16, 32}};
  for (const auto &c : cand) {
    const int p_lim = c[0], q_lim = c[1];
    if (uds::fused128_lds_bytes(p_lim, q_lim, 0, fp, fs) > FUSED_LDS_BUDGET) continue;
    const int t = std::min(p_lim, 4 * uds::FUSED_WAVES * uds::F128_U);
    out = uds::build_network_plan(adj, eadj, inc_n, inc_e, t, t, p_lim, q_lim);
    lds = uds::fused128_lds_bytes(out.p_cap, out.q_cap, out.meta_cap, fp, fs);
    if (lds <= FUSED_LDS_BUDGET && out.p_cap <= 4 * uds::FUSED_WAVES * uds::F128_U) return true;
  }
  return false;
}

}  // namespace

namespace {

template <int FP, int FS, int ACT>
hipError_t launch_fused128_act(const uds::FusedArgs &a, int grid, int64_t lds, hipStream_t st) {
  static unsigned long long attr_done = 0;
  if (hipError_t e = uds::set_max_lds_once(reinterpret_cast<const void *>(&uds::k_fused_cs<128, FP, FS, uds::FUSED_WAVES, ACT>), (int)FUSED_LDS_BUDGET, attr_done); e != hipSuccess) return e;
  hipLaunchKernelGGL((uds::k_fused_cs<128, FP, FS, uds::FUSED_WAVES, ACT>), dim3(grid), dim3(uds::FUSED_WAVES * 64), (size_t)lds, st, a);
  return hipGetLastError();
}

template <int FP, int FS>
hipError_t launch_fused128(const uds::FusedArgs &a, int grid, int64_t lds, hipStream_t st) {
  if (a.act == UDS_ACT_RELU) return launch_fused128_act<FP, FS, UDS_ACT_RELU>(a, grid, lds, st);
  return launch_fused128_act<FP, FS, -1>(a, grid, lds, st);
}

template <int FP, int FS, int ACT>
hipError_t launch_fused_act(const uds::FusedArgs &a, int grid, int64_t lds, hipStream_t st) {
  static unsigned long long attr_done = 0;
  if (hipError_t e = uds::set_max_lds_once(reinterpret_cast<const void *>(&uds::k_fused_tile<FP, FS, ACT>), (int)FUSED_LDS_BUDGET, attr_done); e != hipSuccess) return e;
  hipLaunchKernelGGL((uds::k_fused_tile<FP, FS, ACT>), dim3(grid), dim3(uds::FUSED_WAVES * 64), (size_t)lds, st, a);
  return hipGetLastError();
}

template <int ACT>
hipError_t launch_fused_ws_act(const uds::FusedArgs &a, int grid, int64_t lds, hipStream_t st) {
  static unsigned long long attr_done = 0;
  if (hipError_t e = uds::set_max_lds_once(reinterpret_cast<const void *>(&uds::k_fused_ws<ACT>), (int)FUSED_LDS_BUDGET, attr_done); e != hipSuccess) return e;
  hipLaunchKernelGGL((uds::k_fused_ws<ACT>), dim3(grid), dim3(uds::FUSED_WAVES * 64), (size_t)lds, st, a);
  return hipGetLastError();
}
// the wave-specialised kernel (kernels_fused_ws.hpp): 64-wide rows on both sides, one tensor per side
hipError_t launch_fused_ws(const uds::FusedArgs &a, int grid, int64_t lds, hipStream_t st) {
  if (a.act == UDS_ACT_RELU) return launch_fused_ws_act<UDS_ACT_RELU>(a, grid, lds, st);
  return launch_fused_ws_act<-1>(a, grid, lds, st);
}

// relu (the reference's activation in every shipped model) is compiled in; other activations are decided at run time
template <int FP, int FS>
hipError_t launch_fused(const uds::FusedArgs &a, int grid, int64_t lds, hipStream_t st) {
  if (a.act == UDS_ACT_RELU) return launch_fused_act<FP, FS, UDS_ACT_RELU>(a, grid, lds, st);
  return launch_fused_act<FP, FS, -1>(a, grid, lds, st);
}

hipError_t pack_weights(const float *W, int K, int F_out, uint4 *out, hipStream_t st) {
  const int total = (K / 32) * (F_out / 16) * 64;
  hipLaunchKernelGGL(uds::k_pack_weight_frags, dim3((total + 255) / 256), dim3(256), 0, st, W, K, F_out, out);
  return hipGetLastError();
}

}  // namespace

extern "C" {

int uds_abi_version(void) { return UDS_ABI_VERSION; }
const char *uds_last_error(void) { return g_err.c_str(); }

int uds_csr_create(const int32_t *rowptr, const int32_t *col, int64_t n_rows, int64_t n_cols,
                   int64_t nnz, uds_csr_t **out) {
  UDS_REQUIRE(out != nullptr, "uds_csr_create: out is NULL");
  *out = nullptr;
  UDS_REQUIRE(rowptr != nullptr && (col != nullptr || nnz == 0), "uds_csr_create: NULL index array");
  UDS_REQUIRE(n_rows >= 0 && n_cols >= 0 && nnz >= 0 && n_rows < INT32_MAX && nnz < INT32_MAX,
              "uds_csr_create: sizes out of int32 range");
  UDS_REQUIRE(rowptr[0] == 0 && rowptr[n_rows] == nnz, "uds_csr_create: rowptr[0]=%d rowptr[n]=%d nnz=%lld",
              rowptr[0], rowptr[n_rows], (long long)nnz);
  int32_t max_deg = 0;
  for (int64_t r = 0; r < n_rows; ++r) {
    UDS_REQUIRE(rowptr[r + 1] >= rowptr[r], "uds_csr_create: rowptr decreases at row %lld", (long long)r);
    max_deg = std::max(max_deg, rowptr[r + 1] - rowptr[r]);
  }
  for (int64_t p = 0; p < nnz; ++p)
    UDS_REQUIRE(col[p] >= 0 && col[p] < n_cols, "uds_csr_create: col[%lld]=%d outside [0,%lld)",
                (long long)p, col[p], (long long)n_cols);
  uds_csr *c = new (std::nothrow) uds_csr;
  if (!c) return fail(UDS_ENOMEM, "uds_csr_create: host allocation failed");
  c->n_rows = n_rows;
  c->n_cols = n_cols;
  c->nnz = nnz;
  c->max_degree = max_deg;
  c->host.n_rows = n_rows;
  c->host.n_cols = n_cols;
  c->host.rowptr.assign(rowptr, rowptr + n_rows + 1);
  c->host.col.assign(col, col + nnz);
  // degree-sorted schedule inside windows of 1024 consecutive rows: descending degree, ties by row index (stable).
  // Windowed so that neighbouring workgroups gather neighbouring rows (L2 hits); a global sort scatters them.
  constexpr int64_t ORDER_WINDOW = 1024;
  c->h_order.resize(n_rows);
  std::iota(c->h_order.begin(), c->h_order.end(), 0);
  for (int64_t r0 = 0; r0 < n_rows; r0 += ORDER_WINDOW)
    std::stable_sort(c->h_order.begin() + r0, c->h_order.begin() + std::min(n_rows, r0 + ORDER_WINDOW), [&](int32_t a, int32_t b) {
      return (rowptr[a + 1] - rowptr[a]) > (rowptr[b + 1] - rowptr[b]);
    });
  auto cleanup = [&](int code) {
    hipFree(c->d_rowptr);
    hipFree(c->d_col);
    hipFree(c->d_order);
    hipFree(c->d_rowidx);
    delete c;
    return code;
  };
  hipError_t e;
  if ((e = hipMalloc(&c->d_rowptr, sizeof(int32_t) * (n_rows + 1))) != hipSuccess ||
      (e = hipMalloc(&c->d_col, sizeof(int32_t) * std::max<int64_t>(nnz, 1))) != hipSuccess ||
      (e = hipMalloc(&c->d_order, sizeof(int32_t) * std::max<int64_t>(n_rows, 1))) != hipSuccess ||
      (e = hipMalloc(&c->d_rowidx, sizeof(int32_t) * std::max<int64_t>(nnz, 1))) != hipSuccess)
    return cleanup(fail(UDS_ENOMEM, "uds_csr_create: hipMalloc -> %s", hipGetErrorString(e)));
  std::vector<int32_t> rowidx((size_t)nnz);
  for (int64_t r = 0; r < n_rows; ++r)
    for (int32_t p = rowptr[r]; p < rowptr[r + 1]; ++p) rowidx[p] = (int32_t)r;
  if ((e = hipMemcpy(c->d_rowptr, rowptr, sizeof(int32_t) * (n_rows + 1), hipMemcpyHostToDevice)) != hipSuccess ||
      (nnz && (e = hipMemcpy(c->d_rowidx, rowidx.data(), sizeof(int32_t) * nnz, hipMemcpyHostToDevice)) != hipSuccess) ||
      (nnz && (e = hipMemcpy(c->d_col, col, sizeof(int32_t) * nnz, hipMemcpyHostToDevice)) != hipSuccess) ||
      (n_rows && (e = hipMemcpy(c->d_order, c->h_order.data(), sizeof(int32_t) * n_rows, hipMemcpyHostToDevice)) != hipSuccess))
    return cleanup(fail(UDS_EHIP, "uds_csr_create: hipMemcpy -> %s", hipGetErrorString(e)));
  *out = c;
  return UDS_OK;
}

int uds_csr_destroy(uds_csr_t *c) {
  if (!c) return UDS_OK;
  hipFree(c->d_rowptr);
  hipFree(c->d_col);
  hipFree(c->d_order);
  hipFree(c->d_rowidx);
  delete c;
  return UDS_OK;
}

int uds_csr_shape(const uds_csr_t *c, int64_t *n_rows, int64_t *n_cols, int64_t *nnz, int32_t *max_degree) {
  UDS_REQUIRE(c != nullptr, "uds_csr_shape: NULL handle");
  if (n_rows) *n_rows = c->n_rows;
  if (n_cols) *n_cols = c->n_cols;
  if (nnz) *nnz = c->nnz;
  if (max_degree) *max_degree = c->max_degree;
  return UDS_OK;
}

int uds_csr_row_order(const uds_csr_t *c, int32_t *out_host) {
  UDS_REQUIRE(c != nullptr && out_host != nullptr, "uds_csr_row_order: NULL argument");
  std::memcpy(out_host, c->h_order.data(), sizeof(int32_t) * c->n_rows);
  return UDS_OK;
}

int uds_dense_act(const float *xa, int64_t fa, const float *xb, int64_t fb, int64_t rows, const float *W,
                  const float *bias, int64_t f_out, int act, const float *a_self, const float *a_nbr,
                  float *out, float *s_self, float *s_nbr, uds_stream_t stream) {
  UDS_REQUIRE(xa && W && out, "uds_dense_act: NULL xa/W/out");
  UDS_REQUIRE(((f_out & 3) != 0 || aligned16(out)), "uds_dense_act: out must be 16-byte aligned");
  UDS_REQUIRE(fa > 0 && fb >= 0 && (fb == 0) == (xb == nullptr), "uds_dense_act: fa=%lld fb=%lld xb=%p inconsistent",
              (long long)fa, (long long)fb, (const void *)xb);
  UDS_REQUIRE(rows >= 0 && f_out > 0 && f_out <= 256 && fa + fb <= 4096, "uds_dense_act: rows=%lld f_out=%lld (max 256) f_in=%lld",
              (long long)rows, (long long)f_out, (long long)(fa + fb));
  UDS_REQUIRE(act >= UDS_ACT_LINEAR && act <= UDS_ACT_HARD_SIGMOID, "uds_dense_act: unknown activation %d", act);
  const bool attn = a_self != nullptr;
  UDS_REQUIRE(attn == (a_nbr != nullptr) && attn == (s_self != nullptr) && attn == (s_nbr != nullptr),
              "uds_dense_act: a_self/a_nbr/s_self/s_nbr must be given together");
  if (rows == 0) return UDS_OK;
  uds::DenseArgs a{xa, xb, W, bias, a_self, a_nbr, out, s_self, s_nbr, (int)fa, (int)fb, (int)f_out, act, rows};
  hipError_t e = uds::launch_dense_act(a, static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return fail(UDS_EHIP, "uds_dense_act: launch -> %s", hipGetErrorString(e));
  return UDS_OK;
}

int uds_conv1d_causal(const float *x, int64_t B, int64_t T, int64_t R, int64_t F, const float *kernel, const float *bias,
                      int64_t taps, int64_t dil, int64_t H, int act, float *out, uds_stream_t stream) {
  UDS_REQUIRE(x && kernel && out, "uds_conv1d_causal: NULL x/kernel/out");
  UDS_REQUIRE(B >= 0 && T > 0 && R > 0 && F > 0 && taps > 0 && taps <= 16 && dil != 0 && H > 0 && H <= 256 && taps * F <= 4096,
              "uds_conv1d_causal: bad sizes B=%lld T=%lld R=%lld F=%lld taps=%lld dil=%lld H=%lld", (long long)B, (long long)T,
              (long long)R, (long long)F, (long long)taps, (long long)dil, (long long)H);
  UDS_REQUIRE(act >= UDS_ACT_LINEAR && act <= UDS_ACT_HARD_SIGMOID, "uds_conv1d_causal: unknown activation %d", act);
  UDS_REQUIRE(((H & 3) != 0 || aligned16(out)), "uds_conv1d_causal: out must be 16-byte aligned");
  if (B == 0) return UDS_OK;
  uds::DenseArgs a{x, nullptr, kernel, bias, nullptr, nullptr, out, nullptr, nullptr, (int)F, 0, (int)H, act, B * T * R};
  a.taps = (int)taps;
  a.dil = (int)dil;
  a.T = (int)T;
  a.t_rows = (int)R;
  hipError_t e = uds::launch_dense_act(a, static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return fail(UDS_EHIP, "uds_conv1d_causal: launch -> %s", hipGetErrorString(e));
  return UDS_OK;
}

int uds_recurrent_fused(const float *x, int64_t F, const void *packed, const float *b_in, const float *b_rec, int64_t B, int64_t T, int64_t R,
                        int kind, float *out, uds_stream_t stream) {
  UDS_REQUIRE(x && packed && out && (F == 0 || b_in), "uds_recurrent_fused: NULL argument");
  UDS_REQUIRE(kind == 0 || kind == 1, "uds_recurrent_fused: kind %d (0 = GRU, 1 = LSTM)", kind);
  const int G = kind == 0 ? 3 : 4;
  UDS_REQUIRE(uds::recurrent_mfma_supported(G, (int)F), "uds_recurrent_fused: input width %lld (64 or 128 where W + U fit the LDS, 0 = given projection)",
              (long long)F);
  UDS_REQUIRE(B >= 0 && T >= 0 && R >= 0, "uds_recurrent_fused: bad sizes B=%lld T=%lld R=%lld", (long long)B, (long long)T, (long long)R);
  UDS_REQUIRE(aligned16(x) && aligned16(out) && aligned16(packed) && aligned16(b_in) && aligned16(b_rec),
              "uds_recurrent_fused: buffers must be 16-byte aligned");
  if (B == 0 || T == 0 || R == 0) return UDS_OK;
  const int64_t n_blocks = (R + 15) / 16;
  UDS_REQUIRE(B * n_blocks < INT32_MAX && T < INT32_MAX, "uds_recurrent_fused: too many rows");
  uds::RecurrentMfmaArgs a{x, b_in, b_rec, reinterpret_cast<const uint4 *>(packed), out, (int)B, (int)T, (int)R, (int)n_blocks};
  hipError_t e = uds::launch_recurrent_mfma(a, G, (int)F, static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return fail(UDS_EHIP, "uds_recurrent_fused: launch -> %s", hipGetErrorString(e));
  return UDS_OK;
}

int uds_recurrent_fused_supported(int64_t F, int kind) { return uds::recurrent_mfma_supported(kind == 0 ? 3 : 4, (int)F) ? 1 : 0; }

int uds_recurrent_forward(const float *xp, const float *U, const float *rb, int64_t B, int64_t T, int64_t R, int64_t H, int kind,
                          float *out, uds_stream_t stream) {
  return uds_recurrent_forward_train(xp, U, rb, B, T, R, H, kind, out, nullptr, stream);
}

int uds_recurrent_forward_train(const float *xp, const float *U, const float *rb, int64_t B, int64_t T, int64_t R, int64_t H, int kind,
                                float *out, float *c_out, uds_stream_t stream) {
  UDS_REQUIRE(xp && U && out, "uds_recurrent_forward: NULL argument");
  UDS_REQUIRE(kind == 0 || kind == 1, "uds_recurrent_forward: kind %d (0 = GRU, 1 = LSTM)", kind);
  UDS_REQUIRE(B >= 0 && T >= 0 && R >= 0 && H > 0 && H <= 256, "uds_recurrent_forward: bad sizes B=%lld T=%lld R=%lld H=%lld", (long long)B,
              (long long)T, (long long)R, (long long)H);
  if (B == 0 || T == 0 || R == 0) return UDS_OK;
  const int G = kind == 0 ? 3 : 4;
  const int rows = (int)std::max<int64_t>(1, 256 / H);
  const size_t lds = ((size_t)H * G * H + (size_t)rows * H) * sizeof(float);
  UDS_REQUIRE(lds <= 160 * 1024, "uds_recurrent_forward: the recurrent kernel (%lld x %lld floats) does not fit the 160 KiB LDS", (long long)H,
              (long long)(G * H));
  UDS_REQUIRE((B * R + rows - 1) / rows < INT32_MAX, "uds_recurrent_forward: too many rows");
  uds::RecurrentArgs a{xp, U, rb, out, c_out, (int)B, (int)T, (int)R, (int)H, G, rows};
  hipError_t e = uds::launch_recurrent(a, static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return fail(UDS_EHIP, "uds_recurrent_forward: launch -> %s", hipGetErrorString(e));
  return UDS_OK;
}

int uds_recurrent_backward(const float *xp, const void *packed, const float *b_rec, const float *h, const float *c, const float *gh,
                           int64_t B, int64_t T, int64_t R, int kind, float *dxp, float *darec, uds_stream_t stream) {
  UDS_REQUIRE(xp && packed && h && gh && dxp && darec, "uds_recurrent_backward: NULL argument");
  UDS_REQUIRE(kind == 0 || kind == 1, "uds_recurrent_backward: kind %d (0 = GRU, 1 = LSTM)", kind);
  UDS_REQUIRE(kind == 0 || c, "uds_recurrent_backward: the LSTM needs the cell states of the forward pass (uds_recurrent_forward_train)");
  UDS_REQUIRE(B >= 0 && T >= 0 && R >= 0, "uds_recurrent_backward: bad sizes B=%lld T=%lld R=%lld", (long long)B, (long long)T, (long long)R);
  UDS_REQUIRE(aligned16(xp) && aligned16(packed) && aligned16(b_rec) && aligned16(h) && aligned16(c) && aligned16(gh) && aligned16(dxp) &&
                  aligned16(darec), "uds_recurrent_backward: buffers must be 16-byte aligned");
  if (B == 0 || T == 0 || R == 0) return UDS_OK;
  const int64_t n_blocks = (R + 15) / 16;
  UDS_REQUIRE(B * n_blocks < INT32_MAX && T < INT32_MAX, "uds_recurrent_backward: too many rows");
  uds::RecurrentBwdArgs a{xp, b_rec, reinterpret_cast<const uint4 *>(packed), h, c, gh, dxp, darec, (int)B, (int)T, (int)R, (int)n_blocks};
  hipError_t e = kind == 0 ? uds::launch_recurrent_bwd_t<3>(a, static_cast<hipStream_t>(stream))
                           : uds::launch_recurrent_bwd_t<4>(a, static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return fail(UDS_EHIP, "uds_recurrent_backward: launch -> %s", hipGetErrorString(e));
  return UDS_OK;
}

int64_t uds_rowgemm_packed_bytes(int64_t k_total, int64_t f_out) {
  if (k_total <= 0 || k_total % 32 || f_out <= 0 || f_out > 64) return 0;
  return (k_total / 32) * uds::rowgemm_mb((int)f_out) * 2 * 64 * 16;
}

int uds_rowgemm_pack(const float *W, int64_t k_total, int64_t f_out, void *packed, uds_stream_t stream) {
  UDS_REQUIRE(W && packed && aligned16(packed), "uds_rowgemm_pack: NULL / misaligned argument");
  UDS_REQUIRE(uds_rowgemm_packed_bytes(k_total, f_out) > 0, "uds_rowgemm_pack: needs K %% 32 == 0 and f_out <= 64 (K=%lld f_out=%lld)",
              (long long)k_total, (long long)f_out);
  const int mb = uds::rowgemm_mb((int)f_out);
  const int total = (int)(k_total / 32) * mb * 64;
  hipLaunchKernelGGL(uds::k_pack_weight_frags_padded, dim3((total + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), W,
                     (int)k_total, (int)f_out, mb, reinterpret_cast<uint4 *>(packed));
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(UDS_EHIP, "uds_rowgemm_pack: launch -> %s", hipGetErrorString(e));
  return UDS_OK;
}

static int64_t pad_k(int64_t k) { return (k + 63) / 64 * 64; }      // the remainder GEMM's k-step
constexpr int REMAINDER_MAX_PIECES = 512;      // (tile, K piece) pairs of the cut tiles of k_remainder_gemm2: at most two rounds of workgroups

int64_t uds_remainder_packed_bytes(int64_t R, int64_t M) {
  if (R <= 0 || M <= 0) return 0;
  return 2 * R * pad_k(M) * 2;
}

int uds_remainder_pack(const float *rest, int64_t R, int64_t M, void *packed, uds_stream_t stream) {
  UDS_REQUIRE(rest && packed && aligned16(packed), "uds_remainder_pack: NULL / misaligned argument");
  UDS_REQUIRE(R > 0 && M > 0 && R * pad_k(M) / 8 < (int64_t)INT32_MAX * 256, "uds_remainder_pack: bad shape (%lld, %lld)", (long long)R,
              (long long)M);
  const int64_t Kp = pad_k(M), n = R * Kp / 8;
  __bf16 *hi = reinterpret_cast<__bf16 *>(packed), *lo = hi + R * Kp;
  hipLaunchKernelGGL(uds::k_split_rows_bf16, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), rest, R, M,
                     Kp, hi, lo);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(UDS_EHIP, "uds_remainder_pack: launch -> %s", hipGetErrorString(e));
  return UDS_OK;
}

namespace {
// How uds_remainder_forward tiles (R x Kp) x (Nc x Kp): k_remainder_gemm2<2> (256 x 256 tiles, one 8-wave workgroup per CU) when
// the shape has at least one tile's worth of columns; the tiles that fill whole rounds of 256 workgroups go out as they are, the
// rest are cut along K into ks pieces each so that they fill (most of) one more round.
struct RemainderPlan {
  bool v2;
  int64_t n_ctile, t_main, t_rest, ks;
};
RemainderPlan remainder_plan(int64_t R, int64_t Nc, int64_t Kp) {
  RemainderPlan q{false, 0, 0, 0, 1};
  q.v2 = Nc * Kp * 2 < ((int64_t)1 << 32) && R * Kp * 2 < ((int64_t)1 << 32) && Nc >= 256 && R >= 128;
#ifdef UDS_GEMM_V1
  q.v2 = false;
#endif
  if (!q.v2) return q;
  q.n_ctile = (Nc + 255) / 256;
  const int64_t tiles = q.n_ctile * ((R + 255) / 256);
  q.t_main = tiles / 256 * 256;
  q.t_rest = tiles - q.t_main;
  if (q.t_rest) {      // pieces per cut tile: the fewest rounds-of-256 per piece length, ties to the fewer pieces (each piece = 256 KB out and back)
    double best = 1e300;
    for (int64_t k = 1; k <= std::min<int64_t>({(int64_t)REMAINDER_MAX_PIECES / q.t_rest, 8, Kp / 32 / 8}); ++k) {
      const double cost = (double)((q.t_rest * k + 255) / 256) / (double)k + 0.01 * (double)k;
      if (cost < best - 1e-9) {
        best = cost;
        q.ks = k;
      }
    }
  }
#ifdef UDS_KNOBS
  if (const char *ov = std::getenv("UDS_GEMM_KS")) q.ks = std::max<int64_t>(1, std::min<int64_t>(std::atoll(ov), REMAINDER_MAX_PIECES / std::max<int64_t>(1, q.t_rest)));
#endif
  if (q.ks == 1) {
    q.t_main = tiles;
    q.t_rest = 0;
  }
  return q;
}
}  // namespace

int64_t uds_remainder_workspace_bytes(int64_t R, int64_t M, int64_t S, int64_t h) {
  if (R <= 0 || M <= 0 || S <= 0 || h <= 0) return 0;
  // the split activation planes + the accumulator pieces of the K-cut tiles (256 x 256 floats each)
  const RemainderPlan q = remainder_plan(R, S * h, pad_k(M));
  return 2 * S * h * pad_k(M) * 2 + (q.v2 ? q.t_rest * q.ks * 256 * 256 * 4 : 0);
}

namespace {
// the GEMM of uds_remainder_forward on operand planes that are already in the workspace
int remainder_gemm_from_planes(const void *packed, int64_t R, int64_t Kp, int64_t Nc, int64_t h, void *workspace, float *out, hipStream_t st) {
  const __bf16 *wh = reinterpret_cast<const __bf16 *>(packed), *wl = wh + R * Kp;
  __bf16 *xh = reinterpret_cast<__bf16 *>(workspace), *xl = xh + Nc * Kp;
  hipError_t e = hipSuccess;
  // k_remainder_gemm2<2> (256 x 256 tiles, LDS-DMA staged, one 8-wave workgroup per CU): the tiles that fill whole rounds of 256
  // workgroups go out as they are; the rest are cut along K into ks pieces each so that they fill (most of) one more round, their
  // accumulators land in the workspace behind the activation planes and a third launch adds the pieces in a fixed order
  const RemainderPlan plan = remainder_plan(R, Nc, Kp);
  if (plan.v2) {
    using C = uds::Gemm2Cfg<2>;
#ifdef UDS_GEMM_M32
    constexpr bool M32 = true;       // the 32 x 32 x 16 form: measured 3 % slower than 16 x 16 x 32 here (profiles/r03_gemm2.md)
#else
    constexpr bool M32 = false;
#endif
    static unsigned long long done = 0;
    if ((e = uds::set_max_lds_once(reinterpret_cast<const void *>(&uds::k_remainder_gemm2<2, M32>), C::LDS_BYTES, done)) != hipSuccess)
      return fail(UDS_EHIP, "uds_remainder_forward: LDS attribute -> %s", hipGetErrorString(e));
    const int64_t n_ctile = plan.n_ctile, t_main = plan.t_main, t_rest = plan.t_rest, ks = plan.ks;
    float *partial = reinterpret_cast<float *>(reinterpret_cast<char *>(workspace) + 2 * Nc * Kp * 2);
    hipLaunchKernelGGL((uds::k_remainder_gemm2<2, M32>), dim3((unsigned)(t_main + t_rest * ks)), dim3(512), C::LDS_BYTES, st, xh, xl, wh, wl, Nc, R, Kp, (int)h,
                       (int)n_ctile, out, (int)t_main, (int)ks, partial);
    if (t_rest)
      hipLaunchKernelGGL((uds::k_remainder_gemm2_reduce<2, M32>), dim3((unsigned)t_rest), dim3(512), 0, st, partial, (int)ks, (int)t_main, Nc, R, (int)h,
                         (int)n_ctile, out);
    if ((e = hipGetLastError()) != hipSuccess) return fail(UDS_EHIP, "uds_remainder_forward: launch -> %s", hipGetErrorString(e));
    return UDS_OK;
  }
  {
    const int64_t n_ctile = (Nc + 127) / 128, n_rtile = (R + 127) / 128;
    hipLaunchKernelGGL(uds::k_remainder_gemm, dim3((unsigned)(n_ctile * n_rtile)), dim3(256), 0, st, xh, xl, wh, wl, Nc, R, Kp, (int)h,
                       (int)n_ctile, out);
  }
  e = hipGetLastError();
  if (e != hipSuccess) return fail(UDS_EHIP, "uds_remainder_forward: launch -> %s", hipGetErrorString(e));
  return UDS_OK;
}
}  // namespace

int uds_remainder_forward(const void *packed, int64_t R, int64_t M, const float *x, int64_t S, int64_t h, void *workspace, float *out,
                          uds_stream_t stream) {
  UDS_REQUIRE(S >= 0 && R > 0 && M > 0, "uds_remainder_forward: bad shape");
  if (S == 0) return UDS_OK;
  UDS_REQUIRE(packed && x && workspace && out, "uds_remainder_forward: NULL argument");
  UDS_REQUIRE(h > 0 && h <= 64 && h % 4 == 0, "uds_remainder_forward: h = %lld (needs h %% 4 == 0, h <= 64)", (long long)h);
  UDS_REQUIRE(aligned16(packed) && aligned16(workspace) && aligned16(out), "uds_remainder_forward: buffers must be 16-byte aligned");
  const int64_t Kp = pad_k(M), Nc = S * h;
  UDS_REQUIRE(S <= 65535 && ((Nc + 127) / 128) * ((R + 127) / 128) < INT32_MAX, "uds_remainder_forward: shape exceeds the launch grid");
  const __bf16 *wh = reinterpret_cast<const __bf16 *>(packed), *wl = wh + R * Kp;
  __bf16 *xh = reinterpret_cast<__bf16 *>(workspace), *xl = xh + Nc * Kp;
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(uds::k_split_transpose_bf16, dim3((unsigned)(Kp / 64 + (Kp % 64 != 0)), (unsigned)S), dim3(256), 0, st, x, M, (int)h, Kp,
                     xh, xl);
  return remainder_gemm_from_planes(packed, R, Kp, Nc, h, workspace, out, st);
}

int uds_remainder_forward_dense(const void *packed, int64_t R, int64_t M, const float *e, int64_t F, const void *packed_w, const float *bias,
                                int act, int64_t S, int64_t h, void *workspace, float *out, uds_stream_t stream) {
  UDS_REQUIRE(S >= 0 && R > 0 && M > 0, "uds_remainder_forward_dense: bad shape");
  if (S == 0) return UDS_OK;
  UDS_REQUIRE(packed && e && packed_w && workspace && out, "uds_remainder_forward_dense: NULL argument");
  UDS_REQUIRE((F == 64 || F == 128) && (h == 32 || h == 64), "uds_remainder_forward_dense: F = %lld, h = %lld (F 64 or 128, h 32 or 64)", (long long)F,
              (long long)h);
  UDS_REQUIRE(act >= UDS_ACT_LINEAR && act <= UDS_ACT_HARD_SIGMOID, "uds_remainder_forward_dense: unknown activation %d", act);
  UDS_REQUIRE(aligned16(packed) && aligned16(e) && aligned16(packed_w) && aligned16(bias) && aligned16(workspace) && aligned16(out),
              "uds_remainder_forward_dense: buffers must be 16-byte aligned");
  const int64_t Kp = pad_k(M), Nc = S * h;
  UDS_REQUIRE(S <= 65535 && ((Nc + 127) / 128) * ((R + 127) / 128) < INT32_MAX, "uds_remainder_forward_dense: shape exceeds the launch grid");
  __bf16 *xh = reinterpret_cast<__bf16 *>(workspace), *xl = xh + Nc * Kp;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)(Kp / 64), (unsigned)S);
  if (F == 64)
    hipLaunchKernelGGL(uds::k_dense_split_planes<2>, grid, dim3(256), 0, st, e, M, (int)h, Kp, reinterpret_cast<const uint4 *>(packed_w), bias, act, xh, xl);
  else
    hipLaunchKernelGGL(uds::k_dense_split_planes<4>, grid, dim3(256), 0, st, e, M, (int)h, Kp, reinterpret_cast<const uint4 *>(packed_w), bias, act, xh, xl);
  if (hipError_t er = hipGetLastError(); er != hipSuccess) return fail(UDS_EHIP, "uds_remainder_forward_dense: launch -> %s", hipGetErrorString(er));
  return remainder_gemm_from_planes(packed, R, Kp, Nc, h, workspace, out, st);
}

int uds_rowgemm_forward(const float *x, int64_t B, int64_t T, int64_t R, int64_t F, const void *packed, const float *bias,
                        int64_t taps, int64_t dil, int64_t f_out, int act, float *out, uds_stream_t stream) {
  return uds_rowgemm_forward_cat(x, F, nullptr, 0, B, T, R, packed, bias, taps, dil, f_out, act, out, f_out, 0, stream);
}

int uds_rowgemm_forward_cat(const float *x, int64_t F1, const float *x2, int64_t F2, int64_t B, int64_t T, int64_t R, const void *packed,
                            const float *bias, int64_t taps, int64_t dil, int64_t f_out, int act, float *out, int64_t ldo, int64_t col0,
                            uds_stream_t stream) {
  const int64_t F = F1 + F2;
  UDS_REQUIRE((x2 != nullptr) == (F2 > 0), "uds_rowgemm_forward_cat: x2 / F2 disagree");
  UDS_REQUIRE(!x2 || (taps == 1 && F1 % 32 == 0 && F2 % 32 == 0 && aligned16(x2)),
              "uds_rowgemm_forward_cat: a two-tensor row needs taps = 1 and both widths multiples of 32");
  UDS_REQUIRE(ldo >= col0 + f_out && col0 >= 0 && (ldo == f_out || (ldo % 4 == 0 && col0 % 4 == 0)),
              "uds_rowgemm_forward_cat: output block [%lld, %lld) does not fit rows of %lld floats (4-float aligned)", (long long)col0,
              (long long)(col0 + f_out), (long long)ldo);
  UDS_REQUIRE(x && packed && out, "uds_rowgemm_forward: NULL x/packed/out");
  UDS_REQUIRE(B >= 0 && T > 0 && R > 0 && F > 0 && F % 32 == 0 && taps > 0 && taps <= 16 && dil != 0 && f_out > 0 && f_out <= 64,
              "uds_rowgemm_forward: needs F %% 32 == 0, f_out <= 64 (B=%lld T=%lld R=%lld F=%lld taps=%lld f_out=%lld)", (long long)B,
              (long long)T, (long long)R, (long long)F, (long long)taps, (long long)f_out);
  UDS_REQUIRE(act >= UDS_ACT_LINEAR && act <= UDS_ACT_HARD_SIGMOID, "uds_rowgemm_forward: unknown activation %d", act);
  UDS_REQUIRE(aligned16(x) && aligned16(packed) && aligned16(out) && aligned16(bias), "uds_rowgemm_forward: pointers must be 16-byte aligned");
  UDS_REQUIRE(uds::rowgemm_ring((int)(taps * F), uds::rowgemm_mb((int)f_out)) != 0,
              "uds_rowgemm_forward: K=%lld x f_out=%lld weights do not fit the LDS", (long long)(taps * F), (long long)f_out);
  UDS_REQUIRE(B * T * R < INT32_MAX, "uds_rowgemm_forward: %lld rows exceed the int32 row index", (long long)(B * T * R));
  if (B == 0) return UDS_OK;
  const int64_t n_blocks = (R + 15) / 16;
  if (uds::conv_stream_supported((int)taps, (int)F, (int)f_out, (int)dil) && B * n_blocks >= 256 && ldo == f_out &&
      !std::getenv("UDS_NO_CONV_STREAM")) {
    // enough (batch element, 16-row block) streams to fill the CUs: read every row once instead of once per tap
    uds::ConvStreamArgs ca{x, bias, reinterpret_cast<const uint4 *>(packed), out, (int)B, (int)T, (int)R, act, dil > 0 ? 1 : -1, (int)n_blocks, 1, (int)T};
    hipError_t ec = uds::launch_conv_stream(ca, (int)dil, static_cast<hipStream_t>(stream));
    if (ec != hipSuccess) return fail(UDS_EHIP, "uds_rowgemm_forward: streaming conv launch -> %s", hipGetErrorString(ec));
    return UDS_OK;
  }
  uds::RowGemmArgs a{x, bias, reinterpret_cast<const uint4 *>(packed), out, B * T * R, (int)F, (int)taps, (int)dil, (int)T, (int)R,
                     (int)f_out, act, 0, x2, (int)F1, (int)ldo, (int)col0};
  hipError_t e = uds::launch_rowgemm(a, static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return fail(UDS_EHIP, "uds_rowgemm_forward: launch -> %s", hipGetErrorString(e));
  return UDS_OK;
}

int uds_rowgemm_forward_pair(const float *x0, int64_t R0, const void *packed0, const float *bias0, float *out0, const float *x1, int64_t R1,
                             const void *packed1, const float *bias1, float *out1, int64_t B, int64_t T, int64_t F, int64_t taps, int64_t dil,
                             int64_t f_out, int act, uds_stream_t stream) {
  UDS_REQUIRE(x0 && x1 && packed0 && packed1 && out0 && out1, "uds_rowgemm_forward_pair: NULL argument");
  UDS_REQUIRE(B >= 0 && T > 0 && R0 > 0 && R1 > 0 && F > 0 && F % 32 == 0 && taps > 0 && taps <= 16 && dil > 0 && f_out > 0 && f_out <= 64,
              "uds_rowgemm_forward_pair: needs F %% 32 == 0, f_out <= 64, dil > 0");
  UDS_REQUIRE(act >= UDS_ACT_LINEAR && act <= UDS_ACT_HARD_SIGMOID, "uds_rowgemm_forward_pair: unknown activation %d", act);
  UDS_REQUIRE(aligned16(x0) && aligned16(x1) && aligned16(packed0) && aligned16(packed1) && aligned16(out0) && aligned16(out1) && aligned16(bias0) &&
                  aligned16(bias1), "uds_rowgemm_forward_pair: pointers must be 16-byte aligned");
  if (B == 0) return UDS_OK;
  uds::RowGemmArgs a0{x0, bias0, reinterpret_cast<const uint4 *>(packed0), out0, B * T * R0, (int)F, (int)taps, (int)dil, (int)T, (int)R0,
                      (int)f_out, act, 0, nullptr, 0, (int)f_out, 0};
  uds::RowGemmArgs a1{x1, bias1, reinterpret_cast<const uint4 *>(packed1), out1, B * T * R1, (int)F, (int)taps, (int)dil, (int)T, (int)R1,
                      (int)f_out, act, 0, nullptr, 0, (int)f_out, 0};
  hipError_t e = hipSuccess;
  bool done = false;
  switch (uds::rowgemm_mb((int)f_out)) {
    case 1: done = uds::launch_rowgemm_small_pair<1>(a0, a1, static_cast<hipStream_t>(stream), e); break;
    case 2: done = uds::launch_rowgemm_small_pair<2>(a0, a1, static_cast<hipStream_t>(stream), e); break;
    default: done = uds::launch_rowgemm_small_pair<4>(a0, a1, static_cast<hipStream_t>(stream), e); break;
  }
  if (!done) {      // too large (or a depth the pair kernel is not built for): the two ordinary launches
    int rc = uds_rowgemm_forward(x0, B, T, R0, F, packed0, bias0, taps, dil, f_out, act, out0, stream);
    if (rc != UDS_OK) return rc;
    return uds_rowgemm_forward(x1, B, T, R1, F, packed1, bias1, taps, dil, f_out, act, out1, stream);
  }
  if (e != hipSuccess) return fail(UDS_EHIP, "uds_rowgemm_forward_pair: launch -> %s", hipGetErrorString(e));
  return UDS_OK;
}

int uds_dense_cumsum(const float *x, int64_t B, int64_t T, int64_t R, const void *packed, const float *bias, const float *res, int act,
                     float *out, uds_stream_t stream) {
  UDS_REQUIRE(x && packed && out, "uds_dense_cumsum: NULL x/packed/out");
  UDS_REQUIRE(B >= 0 && T > 0 && R > 0, "uds_dense_cumsum: bad sizes B=%lld T=%lld R=%lld", (long long)B, (long long)T, (long long)R);
  UDS_REQUIRE(act >= UDS_ACT_LINEAR && act <= UDS_ACT_HARD_SIGMOID, "uds_dense_cumsum: unknown activation %d", act);
  UDS_REQUIRE(aligned16(x) && aligned16(packed) && aligned16(out) && aligned16(bias) && aligned16(res),
              "uds_dense_cumsum: pointers must be 16-byte aligned");
  UDS_REQUIRE(B * T * R < INT32_MAX, "uds_dense_cumsum: %lld rows exceed the int32 row index", (long long)(B * T * R));
  if (B == 0) return UDS_OK;
  uds::DenseCumsumArgs a{x, bias, res, reinterpret_cast<const uint4 *>(packed), out, (int)B, (int)T, (int)R, act, (int)((R + 15) / 16)};
  hipError_t e = uds::launch_dense_cumsum(a, static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return fail(UDS_EHIP, "uds_dense_cumsum: launch -> %s", hipGetErrorString(e));
  return UDS_OK;
}

int uds_dense_cumsum_heads(const float *x, int64_t B, int64_t T, int64_t R, const void *packed, const float *bias, const float *res, int act,
                           const uds_heads_t *heads, float *out, uds_stream_t stream) {
  UDS_REQUIRE(x && packed && out && heads, "uds_dense_cumsum_heads: NULL x/packed/heads/out");
  UDS_REQUIRE(B >= 0 && T > 0 && R > 0, "uds_dense_cumsum_heads: bad sizes B=%lld T=%lld R=%lld", (long long)B, (long long)T, (long long)R);
  UDS_REQUIRE(act >= UDS_ACT_LINEAR && act <= UDS_ACT_HARD_SIGMOID, "uds_dense_cumsum_heads: unknown activation %d", act);
  UDS_REQUIRE(heads->a_packed && heads->n_a >= 1 && heads->n_a <= 4, "uds_dense_cumsum_heads: first head needs 1..4 outputs (got %d)", heads->n_a);
  UDS_REQUIRE(heads->n_hidden >= 0 && heads->n_hidden <= 5, "uds_dense_cumsum_heads: 0..5 hidden layers in the second head (got %d)", heads->n_hidden);
  for (int i = 0; i < heads->n_hidden; ++i) UDS_REQUIRE(heads->h_packed[i], "uds_dense_cumsum_heads: hidden layer %d has no weights", i);
  UDS_REQUIRE(heads->n_hidden == 0 || heads->f_packed, "uds_dense_cumsum_heads: the second head has no output layer");
  for (int v : {heads->act_a, heads->act_h, heads->act_f})
    UDS_REQUIRE(v >= UDS_ACT_LINEAR && v <= UDS_ACT_HARD_SIGMOID, "uds_dense_cumsum_heads: unknown head activation %d", v);
  UDS_REQUIRE(aligned16(x) && aligned16(packed) && aligned16(bias) && aligned16(res) && aligned16(heads->a_packed) && aligned16(heads->f_packed),
              "uds_dense_cumsum_heads: pointers must be 16-byte aligned");
  UDS_REQUIRE(B * T * R < INT32_MAX, "uds_dense_cumsum_heads: %lld rows exceed the int32 row index", (long long)(B * T * R));
  if (B == 0) return UDS_OK;
  uds::DenseCumsumArgs a{x, bias, res, reinterpret_cast<const uint4 *>(packed), out, (int)B, (int)T, (int)R, act, (int)((R + 15) / 16)};
  uds::HeadsArgs hd{};
  hd.a_packed = reinterpret_cast<const uint4 *>(heads->a_packed);
  hd.a_bias = heads->a_bias;
  for (int i = 0; i < 5; ++i) {
    hd.h_packed[i] = reinterpret_cast<const uint4 *>(heads->h_packed[i]);
    hd.h_bias[i] = heads->h_bias[i];
  }
  hd.f_packed = reinterpret_cast<const uint4 *>(heads->f_packed);
  hd.f_bias = heads->f_bias;
  hd.n_a = heads->n_a, hd.act_a = heads->act_a, hd.n_hidden = heads->n_hidden, hd.act_h = heads->act_h, hd.act_f = heads->act_f;
  hipError_t e = uds::launch_dense_cumsum_heads(a, hd, static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return fail(UDS_EHIP, "uds_dense_cumsum_heads: launch -> %s", hipGetErrorString(e));
  return UDS_OK;
}

int uds_cumsum_act(const float *x, const float *res, int64_t B, int64_t T, int64_t R, int64_t F, int act, float *out,
                   uds_stream_t stream) {
  UDS_REQUIRE(x && out, "uds_cumsum_act: NULL x/out");
  UDS_REQUIRE(B >= 0 && T > 0 && R > 0 && F > 0 && F % 4 == 0, "uds_cumsum_act: bad sizes (F must be a multiple of 4)");
  UDS_REQUIRE(aligned16(x) && aligned16(out) && aligned16(res), "uds_cumsum_act: x/res/out must be 16-byte aligned");
  UDS_REQUIRE(act >= UDS_ACT_LINEAR && act <= UDS_ACT_HARD_SIGMOID, "uds_cumsum_act: unknown activation %d", act);
  if (B == 0) return UDS_OK;
  uds::CumsumArgs a{x, res, out, (int)B, (int)T, (int)R, (int)(F / 4), act};
  hipError_t e = uds::launch_cumsum(a, static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return fail(UDS_EHIP, "uds_cumsum_act: launch -> %s", hipGetErrorString(e));
  return UDS_OK;
}

int uds_attn_sum_pool(const float *x, const float *k, int64_t B, int64_t R, int64_t F, float *out, uds_stream_t stream) {
  UDS_REQUIRE(x && k && out, "uds_attn_sum_pool: NULL argument");
  UDS_REQUIRE(B >= 0 && R > 0 && F >= 4 && F <= 256 && (F & (F - 1)) == 0, "uds_attn_sum_pool: B=%lld R=%lld F=%lld (F a power of two, 4 .. 256)",
              (long long)B, (long long)R, (long long)F);
  UDS_REQUIRE(aligned16(x) && aligned16(k) && aligned16(out), "uds_attn_sum_pool: x/k/out must be 16-byte aligned");
  UDS_REQUIRE(B < INT32_MAX && R < INT32_MAX, "uds_attn_sum_pool: too many rows");
  if (B == 0) return UDS_OK;
  uds::AttnPoolArgs a{x, k, out, (int)R, (int)(F / 4)};
  hipLaunchKernelGGL(uds::k_attn_sum_pool, dim3((unsigned)B), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(UDS_EHIP, "uds_attn_sum_pool: launch -> %s", hipGetErrorString(e));
  return UDS_OK;
}

int uds_dropout(const float *x, int64_t n, float rate, uint64_t seed, uint64_t offset, float *out, uds_stream_t stream) {
  UDS_REQUIRE(n >= 0 && (n == 0 || (x && out)), "uds_dropout: NULL argument");
  UDS_REQUIRE(rate >= 0.f && rate < 1.f, "uds_dropout: rate=%g outside [0, 1)", (double)rate);
  if (n == 0) return UDS_OK;
  const int64_t groups = (n + (int64_t)(offset & 3) + 3) / 4;
  UDS_REQUIRE((groups + 255) / 256 < INT32_MAX, "uds_dropout: n=%lld too large for one launch", (long long)n);
  uds::DropoutArgs a{x, out, n, (unsigned long long)seed, (unsigned long long)offset,
                     (unsigned)std::min<double>(4294967295.0, std::ceil((double)rate * 4294967296.0)), 1.0f / (1.0f - rate)};
  hipLaunchKernelGGL(uds::k_dropout, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(UDS_EHIP, "uds_dropout: launch -> %s", hipGetErrorString(e));
  return UDS_OK;
}

int uds_flow_balance(const uds_csr_t *inc_n, const float *sign, const float *flow, int64_t S, const float *scale_in,
                     const float *scale_out, float *q_in, float *q_out, uds_stream_t stream) {
  UDS_REQUIRE(inc_n && sign && flow && scale_in && scale_out && q_in && q_out, "uds_flow_balance: NULL argument");
  UDS_REQUIRE(S >= 0 && S <= 65535, "uds_flow_balance: S=%lld outside [0,65535]", (long long)S);
  if (S == 0 || inc_n->n_rows == 0) return UDS_OK;
  uds::FlowArgs a{inc_n->d_rowptr, inc_n->d_col, sign, flow, scale_in, scale_out, q_in, q_out, (int)inc_n->n_rows,
                  (int)inc_n->n_cols, (int)S};
  hipError_t e = uds::launch_flow_balance(a, static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return fail(UDS_EHIP, "uds_flow_balance: launch -> %s", hipGetErrorString(e));
  return UDS_OK;
}

int uds_diffusion_forward(const uds_csr_t *csr, const float *vals, const float *c0, const float *r, const float *tot, int64_t S, int64_t C,
                          int act, float *out, uds_stream_t stream) {
  UDS_REQUIRE(csr && vals && c0 && r && tot && out, "uds_diffusion_forward: NULL argument");
  UDS_REQUIRE(S >= 0 && S <= 65535 && C > 0 && C % 4 == 0, "uds_diffusion_forward: S=%lld C=%lld (needs S <= 65535, C %% 4 == 0)", (long long)S,
              (long long)C);
  UDS_REQUIRE(act >= 0 && act <= 4, "uds_diffusion_forward: unknown activation %d", act);
  UDS_REQUIRE(aligned16(vals) && aligned16(c0) && aligned16(out), "uds_diffusion_forward: vals / c0 / out must be 16-byte aligned");
  if (S == 0 || csr->n_rows == 0) return UDS_OK;
  uds::DiffusionArgs a{csr->d_rowptr, csr->d_col, vals, c0, r, tot, out, (int)csr->n_rows, (int)csr->n_cols, (int)(C / 4), act};
  hipError_t e = uds::launch_diffusion(a, (int)S, static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return fail(UDS_EHIP, "uds_diffusion_forward: launch -> %s", hipGetErrorString(e));
  return UDS_OK;
}

static int halo_rows(const char *what, bool pack, float *x, int64_t n_x, float *e, int64_t n_e, int64_t S, int64_t F, const int32_t *idx_x,
                     int64_t nx, const int32_t *idx_e, int64_t ne, float *buf, uds_stream_t stream) {
  UDS_REQUIRE(S >= 0 && nx >= 0 && ne >= 0 && n_x >= 0 && n_e >= 0 && F > 0 && F % 4 == 0, "%s: bad sizes (S=%lld nx=%lld ne=%lld F=%lld)", what,
              (long long)S, (long long)nx, (long long)ne, (long long)F);
  if (S == 0 || nx + ne == 0) return UDS_OK;
  UDS_REQUIRE(buf && (nx == 0 || (x && idx_x)) && (ne == 0 || (e && idx_e)), "%s: NULL argument", what);
  UDS_REQUIRE(aligned16(buf) && aligned16(x) && aligned16(e), "%s: buffers must be 16-byte aligned", what);
  UDS_REQUIRE(nx + ne < INT32_MAX && S * (nx + ne) * (F / 4) < (int64_t)INT32_MAX * 256, "%s: message too large", what);
  uds::HaloArgs a{x, e, buf, idx_x, idx_e, n_x, n_e, S * (nx + ne) * (F / 4), (int)nx, (int)ne, (int)(F / 4)};
  hipError_t err = uds::launch_halo_rows(a, pack, static_cast<hipStream_t>(stream));
  if (err != hipSuccess) return fail(UDS_EHIP, "%s: launch -> %s", what, hipGetErrorString(err));
  return UDS_OK;
}

int uds_halo_pack(const float *x, int64_t n_x, const float *e, int64_t n_e, int64_t S, int64_t F, const int32_t *idx_x, int64_t nx,
                  const int32_t *idx_e, int64_t ne, float *buf, uds_stream_t stream) {
  return halo_rows("uds_halo_pack", true, const_cast<float *>(x), n_x, const_cast<float *>(e), n_e, S, F, idx_x, nx, idx_e, ne, buf, stream);
}

int uds_halo_unpack(const float *buf, int64_t S, int64_t F, const int32_t *idx_x, int64_t nx, const int32_t *idx_e, int64_t ne, float *x,
                    int64_t n_x, float *e, int64_t n_e, uds_stream_t stream) {
  return halo_rows("uds_halo_unpack", false, x, n_x, e, n_e, S, F, idx_x, nx, idx_e, ne, const_cast<float *>(buf), stream);
}

int uds_roll_update(const uds_csr_t *inc_n, const float *sign, const float *span_e, const float *mini_e, const float *scale_in,
                    const float *scale_out, const float *y, int64_t cy, const float *ey, int64_t ce, const float *b, int64_t B, int64_t so,
                    int64_t T, int flood, float *x, float *ex, float *preds, uds_stream_t stream) {
  UDS_REQUIRE(inc_n && sign && span_e && mini_e && scale_in && scale_out && y && ey && b && x && ex && preds, "uds_roll_update: NULL argument");
  UDS_REQUIRE(B >= 0 && so >= 1 && T >= so && cy >= 1 && cy <= 8 && ce >= 1 && ce <= 8, "uds_roll_update: bad sizes B=%lld so=%lld T=%lld cy=%lld ce=%lld",
              (long long)B, (long long)so, (long long)T, (long long)cy, (long long)ce);
  UDS_REQUIRE(B * std::max(inc_n->n_rows, inc_n->n_cols) < INT32_MAX, "uds_roll_update: batch x rows exceeds the int32 index");
  if (B == 0) return UDS_OK;
  uds::RollArgs a{inc_n->d_rowptr, inc_n->d_col, sign, span_e, mini_e, scale_in, scale_out, y, ey, b, x, ex, preds,
                  (int)B, (int)so, (int)T, (int)inc_n->n_rows, (int)inc_n->n_cols, (int)cy, (int)ce, flood};
  hipError_t e = uds::launch_roll_update(a, static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return fail(UDS_EHIP, "uds_roll_update: launch -> %s", hipGetErrorString(e));
  return UDS_OK;
}

int uds_csr_spmm(const uds_csr_t *csr, const float *val, const float *x, int64_t S, int64_t F, const float *bias,
                 int act, float *out, uds_stream_t stream) {
  UDS_REQUIRE(csr && x && out, "uds_csr_spmm: NULL csr/x/out");
  UDS_REQUIRE(S >= 0 && F > 0 && F % 4 == 0, "uds_csr_spmm: S=%lld F=%lld (F must be a positive multiple of 4)",
              (long long)S, (long long)F);
  UDS_REQUIRE(aligned16(x) && aligned16(out) && aligned16(bias), "uds_csr_spmm: x/out/bias must be 16-byte aligned");
  UDS_REQUIRE(act >= UDS_ACT_LINEAR && act <= UDS_ACT_HARD_SIGMOID, "uds_csr_spmm: unknown activation %d", act);
  UDS_REQUIRE(S <= 65535, "uds_csr_spmm: S=%lld exceeds 65535 snapshots per call", (long long)S);
  if (S == 0 || csr->n_rows == 0) return UDS_OK;
  uds::SpmmArgs a{csr->d_rowptr, csr->d_col, csr->d_order, val, x, bias, out,
                  (int)csr->n_rows, (int)csr->n_cols, (int)(F / 4), act, (int)S};
  hipError_t e = uds::launch_csr_spmm(a, static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return fail(UDS_EHIP, "uds_csr_spmm: launch -> %s", hipGetErrorString(e));
  return UDS_OK;
}

int64_t uds_gat_workspace_floats(int64_t n, int64_t S, int64_t d) { return align4(S * n * (d + 2)); }

int uds_gat_forward(const uds_csr_t *g, const float *xa, int64_t fa, const float *xb, int64_t fb, int64_t S,
                    const float *W, const float *a_self, const float *a_nbr, const float *bias, int64_t d, int act,
                    float *ws, float *out, uds_stream_t stream) {
  UDS_REQUIRE(g && xa && W && a_self && a_nbr && ws && out, "uds_gat_forward: NULL argument");
  UDS_REQUIRE(g->n_rows == g->n_cols, "uds_gat_forward: pattern must be square (%lld x %lld)", (long long)g->n_rows,
              (long long)g->n_cols);
  UDS_REQUIRE(d > 0 && d % 4 == 0 && d <= 256, "uds_gat_forward: d=%lld must be a multiple of 4, at most 256", (long long)d);
  UDS_REQUIRE(S >= 0 && S <= 65535, "uds_gat_forward: S=%lld outside [0,65535]", (long long)S);
  UDS_REQUIRE(aligned16(ws) && aligned16(out) && aligned16(bias), "uds_gat_forward: workspace/out/bias must be 16-byte aligned");
  if (S == 0 || g->n_rows == 0) return UDS_OK;
  const int64_t n = g->n_rows;
  float *hx = ws;
  float *s_self = hx + S * n * d;
  float *s_nbr = s_self + S * n;
  int rc = uds_dense_act(xa, fa, xb, fb, S * n, W, nullptr, d, UDS_ACT_LINEAR, a_self, a_nbr, hx, s_self, s_nbr, stream);
  if (rc != UDS_OK) return rc;
  uds::GatArgs a{g->d_rowptr, g->d_col, g->d_order, hx, s_self, s_nbr, bias, out, (int)n, (int)(d / 4), act, (int)S};
  hipError_t e = uds::launch_gat_aggregate(a, static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return fail(UDS_EHIP, "uds_gat_forward: launch -> %s", hipGetErrorString(e));
  return UDS_OK;
}

int uds_gat_aggregate(const uds_csr_t *g, const float *hx, const float *s_self, const float *s_nbr, const float *bias, int64_t S,
                      int64_t d, int act, float *out, uds_stream_t stream) {
  UDS_REQUIRE(g && hx && s_self && s_nbr && out, "uds_gat_aggregate: NULL argument");
  UDS_REQUIRE(g->n_rows == g->n_cols, "uds_gat_aggregate: pattern must be square");
  UDS_REQUIRE(d > 0 && d % 4 == 0 && d <= 256, "uds_gat_aggregate: d=%lld must be a multiple of 4, at most 256", (long long)d);
  UDS_REQUIRE(S >= 0 && S <= 65535, "uds_gat_aggregate: S=%lld outside [0,65535]", (long long)S);
  UDS_REQUIRE(act >= UDS_ACT_LINEAR && act <= UDS_ACT_HARD_SIGMOID, "uds_gat_aggregate: unknown activation %d", act);
  UDS_REQUIRE(aligned16(hx) && aligned16(out) && aligned16(bias), "uds_gat_aggregate: hx/out/bias must be 16-byte aligned");
  if (S == 0 || g->n_rows == 0) return UDS_OK;
  uds::GatArgs a{g->d_rowptr, g->d_col, g->d_order, hx, s_self, s_nbr, bias, out, (int)g->n_rows, (int)(d / 4), act, (int)S};
  hipError_t e = uds::launch_gat_aggregate(a, static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return fail(UDS_EHIP, "uds_gat_aggregate: launch -> %s", hipGetErrorString(e));
  return UDS_OK;
}

int uds_gat_aggregate_coef(const uds_csr_t *g, const float *hx, const float *s_self, const float *s_nbr, const float *bias,
                           const float *coef, int64_t S, int64_t d, int act, float *out, uds_stream_t stream) {
  UDS_REQUIRE(g && hx && s_self && s_nbr && coef && out, "uds_gat_aggregate_coef: NULL argument");
  UDS_REQUIRE(g->n_rows == g->n_cols, "uds_gat_aggregate_coef: pattern must be square");
  UDS_REQUIRE(d > 0 && d % 4 == 0 && d <= 256, "uds_gat_aggregate_coef: d=%lld must be a multiple of 4, at most 256", (long long)d);
  UDS_REQUIRE(S >= 0 && S <= 65535, "uds_gat_aggregate_coef: S=%lld outside [0,65535]", (long long)S);
  UDS_REQUIRE(act >= UDS_ACT_LINEAR && act <= UDS_ACT_HARD_SIGMOID, "uds_gat_aggregate_coef: unknown activation %d", act);
  UDS_REQUIRE(aligned16(hx) && aligned16(out) && aligned16(bias), "uds_gat_aggregate_coef: hx/out/bias must be 16-byte aligned");
  if (S == 0 || g->n_rows == 0) return UDS_OK;
  uds::GatArgs a{g->d_rowptr, g->d_col, g->d_order, hx, s_self, s_nbr, bias, out, (int)g->n_rows, (int)(d / 4), act, (int)S};
  hipError_t e = uds::launch_gat_aggregate_coef(a, coef, g->nnz, static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return fail(UDS_EHIP, "uds_gat_aggregate_coef: launch -> %s", hipGetErrorString(e));
  return UDS_OK;
}

int uds_gat_aggregate_masked(const uds_csr_t *g, const float *hx, const float *s_self, const float *s_nbr, const float *bias,
                             const float *edge_mask, int64_t S, int64_t d, int act, float *out, uds_stream_t stream) {
  UDS_REQUIRE(g && hx && s_self && s_nbr && edge_mask && out, "uds_gat_aggregate_masked: NULL argument");
  UDS_REQUIRE(g->n_rows == g->n_cols, "uds_gat_aggregate_masked: pattern must be square");
  UDS_REQUIRE(d > 0 && d % 4 == 0 && d <= 256, "uds_gat_aggregate_masked: d=%lld must be a multiple of 4, at most 256", (long long)d);
  UDS_REQUIRE(S >= 0 && S <= 65535, "uds_gat_aggregate_masked: S=%lld outside [0,65535]", (long long)S);
  UDS_REQUIRE(act >= UDS_ACT_LINEAR && act <= UDS_ACT_HARD_SIGMOID, "uds_gat_aggregate_masked: unknown activation %d", act);
  UDS_REQUIRE(aligned16(hx) && aligned16(out) && aligned16(bias), "uds_gat_aggregate_masked: hx/out/bias must be 16-byte aligned");
  if (S == 0 || g->n_rows == 0) return UDS_OK;
  uds::GatArgs a{g->d_rowptr, g->d_col, g->d_order, hx, s_self, s_nbr, bias, out, (int)g->n_rows, (int)(d / 4), act, (int)S};
  hipError_t e = uds::launch_gat_aggregate_masked(a, edge_mask, g->nnz, static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return fail(UDS_EHIP, "uds_gat_aggregate_masked: launch -> %s", hipGetErrorString(e));
  return UDS_OK;
}

int uds_gat_backward(const uds_csr_t *g, const uds_csr_t *gt, const int32_t *perm_t, const float *grad, const float *hx,
                     const float *s_self, const float *s_nbr, const float *a_self, const float *a_nbr, int64_t S, int64_t d,
                     float *alpha_ws, float *de_ws, float *d_hx, float *ds_self, float *ds_nbr, uds_stream_t stream) {
  return uds_gat_backward_coef(g, gt, perm_t, grad, hx, s_self, s_nbr, a_self, a_nbr, nullptr, S, d, alpha_ws, de_ws, d_hx, ds_self, ds_nbr, stream);
}

int uds_gat_backward_coef(const uds_csr_t *g, const uds_csr_t *gt, const int32_t *perm_t, const float *grad, const float *hx,
                          const float *s_self, const float *s_nbr, const float *a_self, const float *a_nbr, const float *coef, int64_t S,
                          int64_t d, float *alpha_ws, float *de_ws, float *d_hx, float *ds_self, float *ds_nbr, uds_stream_t stream) {
  UDS_REQUIRE(g && gt && perm_t && grad && hx && s_self && s_nbr && a_self && a_nbr && alpha_ws && de_ws && d_hx && ds_self && ds_nbr,
              "uds_gat_backward: NULL argument");
  UDS_REQUIRE(g->n_rows == g->n_cols && gt->n_rows == g->n_rows && gt->n_cols == g->n_cols && gt->nnz == g->nnz,
              "uds_gat_backward: the pattern and its transpose must be square with the same shape and entry count");
  UDS_REQUIRE(d > 0 && d % 4 == 0 && d <= 256, "uds_gat_backward: d=%lld must be a multiple of 4, at most 256", (long long)d);
  UDS_REQUIRE(S >= 0 && S <= 65535, "uds_gat_backward: S=%lld outside [0,65535]", (long long)S);
  UDS_REQUIRE(aligned16(grad) && aligned16(hx) && aligned16(d_hx) && aligned16(a_self) && aligned16(a_nbr),
              "uds_gat_backward: grad/hx/d_hx/a_self/a_nbr must be 16-byte aligned");
  if (S == 0 || g->n_rows == 0) return UDS_OK;
  const int d4 = (int)(d / 4);
  uds::GatBwdRowsArgs ra{g->d_rowptr, g->d_col, grad, hx, s_self, s_nbr, alpha_ws, de_ws, ds_self,
                         (int)g->n_rows, d4, (int)S, uds::lanes_per_item(d4), g->nnz, coef};
  hipError_t e = uds::launch_gat_bwd_rows(ra, static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return fail(UDS_EHIP, "uds_gat_backward: row pass launch -> %s", hipGetErrorString(e));
  uds::GatBwdColsArgs ca{gt->d_rowptr, gt->d_col, perm_t, grad, alpha_ws, de_ws, ds_self, a_self, a_nbr, d_hx, ds_nbr,
                         (int)g->n_rows, d4, (int)S, g->nnz};
  e = uds::launch_gat_bwd_cols(ca, static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return fail(UDS_EHIP, "uds_gat_backward: column pass launch -> %s", hipGetErrorString(e));
  return UDS_OK;
}

int uds_csr_sddmm(const uds_csr_t *csr, const float *a, const float *b, int64_t S, int64_t F, float *out, uds_stream_t stream) {
  UDS_REQUIRE(csr && a && b && out, "uds_csr_sddmm: NULL argument");
  UDS_REQUIRE(S >= 0 && F > 0 && F % 4 == 0, "uds_csr_sddmm: S=%lld F=%lld (F must be a positive multiple of 4)", (long long)S,
              (long long)F);
  UDS_REQUIRE(aligned16(a) && aligned16(b), "uds_csr_sddmm: a/b must be 16-byte aligned");
  if (csr->nnz == 0) return UDS_OK;
  const int f4 = (int)(F / 4);
  uds::SddmmArgs sa{csr->d_rowidx, csr->d_col, a, b, out, (int)csr->n_rows, (int)csr->n_cols, f4, (int)S, uds::lanes_per_item(f4),
                    csr->nnz};
  hipError_t e = uds::launch_csr_sddmm(sa, static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return fail(UDS_EHIP, "uds_csr_sddmm: launch -> %s", hipGetErrorString(e));
  return UDS_OK;
}

int64_t uds_wgrad_workspace_floats(int64_t rows, int64_t F, int64_t H, int with_bias) {
  const int mt = uds::wgrad_mt((int)(F + (with_bias ? 1 : 0))), nt = uds::wgrad_nt((int)H);
  if (!mt || !nt || rows <= 0) return 0;
  return (int64_t)uds::wgrad_grid(rows) * mt * 16 * nt * 16;
}

int uds_wgrad(const float *a, const float *g, int64_t B, int64_t T, int64_t R, int64_t F, int64_t H, int64_t shift, int with_bias,
              float *workspace, float *d_kernel, float *d_bias, uds_stream_t stream) {
  UDS_REQUIRE(a && g && workspace && d_kernel, "uds_wgrad: NULL argument");
  UDS_REQUIRE(!with_bias || d_bias, "uds_wgrad: with_bias needs d_bias");
  UDS_REQUIRE(B >= 0 && T > 0 && R > 0 && F > 0 && H > 0 && shift >= 0, "uds_wgrad: bad sizes");
  const int fr = (int)(F + (with_bias ? 1 : 0));
  const int mt = uds::wgrad_mt(fr), nt = uds::wgrad_nt((int)H);
  UDS_REQUIRE(mt && nt, "uds_wgrad: F=%lld (at most 128 rows incl. the bias row) or H=%lld (at most 64) not supported", (long long)F, (long long)H);
  UDS_REQUIRE(B * T * R < INT32_MAX, "uds_wgrad: %lld rows exceed the int32 row index", (long long)(B * T * R));
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t rows = B * T * R;
  if (rows == 0) {
    hipError_t e0 = hipMemsetAsync(d_kernel, 0, sizeof(float) * F * H, st);
    if (e0 == hipSuccess && with_bias) e0 = hipMemsetAsync(d_bias, 0, sizeof(float) * H, st);
    if (e0 != hipSuccess) return fail(UDS_EHIP, "uds_wgrad: memset -> %s", hipGetErrorString(e0));
    return UDS_OK;
  }
  const int grid = uds::wgrad_grid(rows);
  int64_t rpw = (rows + (int64_t)grid * 4 - 1) / ((int64_t)grid * 4);
  rpw = (rpw + 31) / 32 * 32;
  uds::WgradArgs wa{a, g, workspace, rows, (int)F, (int)H, with_bias ? (int)F : -1, (int)shift, (int)T, (int)R, (int)rpw};
  hipError_t e = uds::launch_wgrad(wa, mt, nt, grid, st);
  if (e != hipSuccess) return fail(UDS_EHIP, "uds_wgrad: launch -> %s", hipGetErrorString(e));
  hipLaunchKernelGGL(uds::k_wgrad_reduce, dim3((unsigned)((F * H + 15) / 16)), dim3(256), 0, st, workspace, grid, mt * 16, nt * 16,
                     (int)F, (int)H, d_kernel);
  if (with_bias)
    hipLaunchKernelGGL(uds::k_wgrad_reduce, dim3((unsigned)((H + 15) / 16)), dim3(256), 0, st, workspace + F * (nt * 16), grid,
                       mt * 16, nt * 16, 1, (int)H, d_bias);
  e = hipGetLastError();
  if (e != hipSuccess) return fail(UDS_EHIP, "uds_wgrad: reduce launch -> %s", hipGetErrorString(e));
  return UDS_OK;
}

// Build (once) and upload the tile plan of kernel variant <fp, fs>.  A plan that does not fit the LDS leaves the slot
// !ok (the layer then runs unfused); only an allocation / copy failure is an error.
static int build_slot(uds_network *n, int fp, int fs) {
  uds_plan_slot &sl = n->slot[slot_index(fp, fs)];
  if (sl.built) return UDS_OK;
  sl.built = true;
  sl.fp = fp;
  sl.fs = fs;
  if ((fp == 128 || fs == 128) ? !plan_network128(n->adj->host, n->edge_adj->host, n->inc_n->host, n->inc_e->host, fp, fs, sl.plan, sl.lds_bytes)
                : !plan_network(n->adj->host, n->edge_adj->host, n->inc_n->host, n->inc_e->host, fp, fs, sl.plan, sl.lds_bytes, sl.blk_cap))
    return UDS_OK;
  hipError_t e;
  if ((e = hipMalloc(&sl.d_hdr, sizeof(int32_t) * sl.plan.hdr.size())) != hipSuccess ||
      (e = hipMalloc(&sl.d_pool, sizeof(int32_t) * sl.plan.pool.size())) != hipSuccess ||
      (e = hipMemcpy(sl.d_hdr, sl.plan.hdr.data(), sizeof(int32_t) * sl.plan.hdr.size(), hipMemcpyHostToDevice)) != hipSuccess ||
      (e = hipMemcpy(sl.d_pool, sl.plan.pool.data(), sizeof(int32_t) * sl.plan.pool.size(), hipMemcpyHostToDevice)) != hipSuccess)
    return fail(UDS_ENOMEM, "tile plan upload -> %s", hipGetErrorString(e));
  auto upload_blocks = [&](const std::vector<int32_t> &hdr, int n_tiles, int32_t **dst) -> hipError_t {
    if (fp == 128 || fs == 128) return hipSuccess;
    std::vector<int32_t> bl;
    uds::build_ell_blocks(hdr, sl.plan.pool, n_tiles, sl.blk_cap, bl);
    hipError_t e2 = hipMalloc(dst, sizeof(int32_t) * bl.size());
    if (e2 != hipSuccess) return e2;
    return hipMemcpy(*dst, bl.data(), sizeof(int32_t) * bl.size(), hipMemcpyHostToDevice);
  };
  if ((e = upload_blocks(sl.plan.hdr, sl.plan.n_tiles, &sl.d_blocks)) != hipSuccess)
    return fail(UDS_ENOMEM, "tile block upload -> %s", hipGetErrorString(e));
  for (int side = 0; side < 2; ++side) {   // compact per-side header lists, in the merged (locality) order
    std::vector<int32_t> hs;
    for (int t = 0; t < sl.plan.n_tiles; ++t)
      if (sl.plan.hdr[(size_t)t * uds::TILE_HDR_INTS + 6] == side)
        hs.insert(hs.end(), sl.plan.hdr.begin() + (size_t)t * uds::TILE_HDR_INTS, sl.plan.hdr.begin() + (size_t)(t + 1) * uds::TILE_HDR_INTS);
    if ((e = hipMalloc(&sl.d_hdr_side[side], sizeof(int32_t) * std::max<size_t>(hs.size(), 1))) != hipSuccess ||
        (!hs.empty() && (e = hipMemcpy(sl.d_hdr_side[side], hs.data(), sizeof(int32_t) * hs.size(), hipMemcpyHostToDevice)) != hipSuccess))
      return fail(UDS_ENOMEM, "tile plan upload -> %s", hipGetErrorString(e));
    if ((e = upload_blocks(hs, (int)(hs.size() / uds::TILE_HDR_INTS), &sl.d_blocks_side[side])) != hipSuccess)
      return fail(UDS_ENOMEM, "tile block upload -> %s", hipGetErrorString(e));
  }
  sl.ok = true;
  return UDS_OK;
}

int uds_network_prepare(uds_network_t *net, int64_t fx, int64_t fe) {
  UDS_REQUIRE(net != nullptr, "uds_network_prepare: NULL network");
  if (net->adj->n_rows == 0 || net->edge_adj->n_rows == 0) return UDS_OK;
  if (fx == 128 && (fe == 128 || fe == 64)) {      // the d = 128 kernel: node tiles <fx, fe>, link tiles <fe, fx>
    int rc = build_slot(net, 128, (int)fe);
    if (rc == UDS_OK && fe == 64) rc = build_slot(net, 64, 128);
    return rc;
  }
  if (!((fx == 64 || fx == 96) && (fe == 64 || fe == 96))) return UDS_OK;
  int rc = build_slot(net, (int)fx, (int)fe);            // node tiles run <fx, fe>
  if (rc == UDS_OK) rc = build_slot(net, (int)fe, (int)fx);   // link tiles <fe, fx>
  return rc;
}

int uds_network_create(const uds_csr_t *adj, const uds_csr_t *edge_adj, const uds_csr_t *inc_n, const uds_csr_t *inc_e,
                       uds_network_t **out) {
  UDS_REQUIRE(out != nullptr, "uds_network_create: out is NULL");
  *out = nullptr;
  UDS_REQUIRE(adj && edge_adj && inc_n && inc_e, "uds_network_create: NULL pattern");
  const int64_t N = adj->n_rows, E = edge_adj->n_rows;
  UDS_REQUIRE(adj->n_cols == N && edge_adj->n_cols == E, "uds_network_create: adjacency patterns must be square");
  UDS_REQUIRE(inc_n->n_rows == N && inc_n->n_cols == E && inc_e->n_rows == E && inc_e->n_cols == N,
              "uds_network_create: incidence shapes (%lld x %lld), (%lld x %lld) do not match N=%lld E=%lld",
              (long long)inc_n->n_rows, (long long)inc_n->n_cols, (long long)inc_e->n_rows, (long long)inc_e->n_cols,
              (long long)N, (long long)E);
  uds_network *n = new (std::nothrow) uds_network;
  if (!n) return fail(UDS_ENOMEM, "uds_network_create: host allocation failed");
  n->adj = adj;
  n->edge_adj = edge_adj;
  n->inc_n = inc_n;
  n->inc_e = inc_e;
  // the tile plan of the 64-wide kernel variant is built now; 96-wide variants on request (uds_network_prepare)
  if (N > 0 && E > 0) {
    int rc = build_slot(n, 64, 64);
    if (rc != UDS_OK) {
      uds_network_destroy(n);
      return rc;
    }
  }
  *out = n;
  return UDS_OK;
}

int uds_network_destroy(uds_network_t *net) {
  if (!net) return UDS_OK;
  for (uds_plan_slot &sl : net->slot) {
    hipFree(sl.d_hdr);
    hipFree(sl.d_pool);
    hipFree(sl.d_hdr_side[0]);
    hipFree(sl.d_hdr_side[1]);
    hipFree(sl.d_blocks);
    hipFree(sl.d_blocks_side[0]);
    hipFree(sl.d_blocks_side[1]);
  }
  delete net;
  return UDS_OK;
}

int uds_network_plan_info(const uds_network_t *net, int32_t *info8) {
  UDS_REQUIRE(net && info8, "uds_network_plan_info: NULL argument");
  const uds_plan_slot &sl = net->slot[0];   // the plan for 64-float rows (d = 64 layers)
  info8[0] = (sl.ok ? 1 : 0) | ((net->slot[1].ok && net->slot[2].ok) ? 2 : 0) | (net->slot[3].ok ? 4 : 0) | (net->slot[4].ok ? 8 : 0) |
             ((net->slot[5].ok && net->slot[6].ok) ? 16 : 0);
  info8[1] = sl.plan.side[0].n_tiles;
  info8[2] = sl.plan.side[1].n_tiles;
  info8[3] = sl.plan.p_cap;
  info8[4] = sl.plan.q_cap;
  info8[5] = sl.blk_cap ? sl.blk_cap : sl.plan.meta_cap;
  info8[6] = (int32_t)sl.lds_bytes;
  info8[7] = sl.plan.t_max[0] * 1000 + sl.plan.t_max[1];
  return UDS_OK;
}

// ---- host-only tile planner access (integer bookkeeping, testable without a GPU) ----
int uds_tile_plan_create(const int32_t *adj_rowptr, const int32_t *adj_col, const int32_t *eadj_rowptr, const int32_t *eadj_col,
                         const int32_t *incn_rowptr, const int32_t *incn_col, const int32_t *ince_rowptr, const int32_t *ince_col,
                         int64_t n_node, int64_t n_edge, int32_t t_node, int32_t t_link, int32_t p_limit, int32_t q_limit,
                         uds_tile_plan_t **out) {
  UDS_REQUIRE(out != nullptr, "uds_tile_plan_create: out is NULL");
  *out = nullptr;
  UDS_REQUIRE(adj_rowptr && adj_col && eadj_rowptr && eadj_col && incn_rowptr && ince_rowptr, "uds_tile_plan_create: NULL array");
  UDS_REQUIRE(n_node > 0 && n_edge > 0 && t_node >= 1 && t_link >= 1, "uds_tile_plan_create: bad sizes");
  auto mk = [](const int32_t *rp, const int32_t *c, int64_t r, int64_t cc) {
    uds::HostCsr h;
    h.n_rows = r;
    h.n_cols = cc;
    h.rowptr.assign(rp, rp + r + 1);
    h.col.assign(c, c + rp[r]);
    return h;
  };
  uds_tile_plan *tp = new (std::nothrow) uds_tile_plan;
  if (!tp) return fail(UDS_ENOMEM, "uds_tile_plan_create: host allocation failed");
  tp->plan = uds::build_network_plan(mk(adj_rowptr, adj_col, n_node, n_node), mk(eadj_rowptr, eadj_col, n_edge, n_edge),
                                     mk(incn_rowptr, incn_col, n_node, n_edge), mk(ince_rowptr, ince_col, n_edge, n_node),
                                     t_node, t_link, p_limit, q_limit);
  *out = tp;
  return UDS_OK;
}

int uds_tile_plan_destroy(uds_tile_plan_t *tp) {
  delete tp;
  return UDS_OK;
}

int uds_tile_plan_sizes(const uds_tile_plan_t *tp, int64_t *n_tiles, int64_t *pool_len, int32_t *caps3) {
  UDS_REQUIRE(tp != nullptr, "uds_tile_plan_sizes: NULL plan");
  if (n_tiles) *n_tiles = tp->plan.n_tiles;
  if (pool_len) *pool_len = (int64_t)tp->plan.pool.size();
  if (caps3) {
    caps3[0] = tp->plan.p_cap;
    caps3[1] = tp->plan.q_cap;
    caps3[2] = tp->plan.meta_cap;
  }
  return UDS_OK;
}

int uds_tile_plan_copy(const uds_tile_plan_t *tp, int32_t *hdr_out, int32_t *pool_out) {
  UDS_REQUIRE(tp && hdr_out && pool_out, "uds_tile_plan_copy: NULL argument");
  std::memcpy(hdr_out, tp->plan.hdr.data(), sizeof(int32_t) * tp->plan.hdr.size());
  std::memcpy(pool_out, tp->plan.pool.data(), sizeof(int32_t) * tp->plan.pool.size());
  return UDS_OK;
}

int uds_tile_plan_blocks(const uds_tile_plan_t *tp, int32_t *blocks_out, int64_t *stride_out) {
  UDS_REQUIRE(tp != nullptr, "uds_tile_plan_blocks: NULL plan");
  const int cap = uds::ell_block_cap(tp->plan.hdr, tp->plan.pool, tp->plan.n_tiles);
  UDS_REQUIRE(cap >= 0, "uds_tile_plan_blocks: a tile has more than 255 primary / 256 secondary rows (byte-wide local indices)");
  if (stride_out) *stride_out = cap;
  if (!blocks_out) return UDS_OK;
  std::vector<int32_t> bl;
  uds::build_ell_blocks(tp->plan.hdr, tp->plan.pool, tp->plan.n_tiles, cap, bl);
  std::memcpy(blocks_out, bl.data(), sizeof(int32_t) * (size_t)cap * tp->plan.n_tiles);
  return UDS_OK;
}

int64_t uds_spatial_workspace_floats(const uds_network_t *net, int64_t S, int64_t h, int64_t d) {
  if (!net) return 0;
  const int64_t N = net->adj->n_rows, E = net->edge_adj->n_rows;
  // packed weight fragments (fused path) + x_e (E,h) + e_x (N,h) + agg_n (N,h) + agg_e (E,h) + hx,s (N,d+2) + he,s (E,d+2)
  return PACKED_WEIGHT_FLOATS + S * 2 * (N + E) * h + align4(S * N * (d + 2)) + align4(S * E * (d + 2));
}


int64_t uds_spatial_packed_bytes(void) { return PACKED_WEIGHT_FLOATS * 4; }

int uds_spatial_pack_weights(const uds_spatial_params_t *p, int64_t fx, int64_t fe, int64_t h, int64_t d, void *packed_out,
                             uds_stream_t stream) {
  UDS_REQUIRE(p && packed_out && p->xe_k && p->ex_k && p->gx_k && p->ge_k, "uds_spatial_pack_weights: NULL argument");
  const bool wide = h == uds::F128_H && d == uds::F128_D && fx == 128 && (fe == 128 || fe == 64);
  UDS_REQUIRE(wide || (h == uds::FUSED_H && d == uds::FUSED_D && (fx == 64 || fx == 96) && (fe == 64 || fe == 96)),
              "uds_spatial_pack_weights: the fused kernels take h=32, d=64, fx, fe in {64, 96} or h=64, d=128, fx=128, fe in {64, 128} (got h=%lld d=%lld "
              "fx=%lld fe=%lld)", (long long)h, (long long)d, (long long)fx, (long long)fe);
  UDS_REQUIRE(aligned16(packed_out), "uds_spatial_pack_weights: output must be 16-byte aligned");
  hipStream_t st = static_cast<hipStream_t>(stream);
  uint4 *wq = reinterpret_cast<uint4 *>(packed_out);
  hipError_t he;
  // node side: secondary = links (xe_k on e), big = gx_k; link side: secondary = nodes (ex_k on x), big = ge_k
  const int o_small = wide ? 2048 : 768, o_big = wide ? 6144 : 2048;      // uint4 per packed kernel
  if ((he = pack_weights(p->xe_k, (int)fe, (int)h, wq, st)) != hipSuccess ||
      (he = pack_weights(p->gx_k, (int)(fx + h), (int)d, wq + o_small, st)) != hipSuccess ||
      (he = pack_weights(p->ex_k, (int)fx, (int)h, wq + o_small + o_big, st)) != hipSuccess ||
      (he = pack_weights(p->ge_k, (int)(fe + h), (int)d, wq + 2 * o_small + o_big, st)) != hipSuccess)
    return fail(UDS_EHIP, "uds_spatial_pack_weights: launch -> %s", hipGetErrorString(he));
  if (!wide && fx == 64 && fe == 64) {      // the wave-specialised kernel's P2 multiplies with 32x32x16 MFMAs: the big kernels in that fragment order too
    const int total = (int)((fx + h) / 16) * (int)(d / 32) * 64;
    hipLaunchKernelGGL(uds::k_pack_weight_frags32, dim3((total + 255) / 256), dim3(256), 0, st, p->gx_k, (int)(fx + h), (int)d, wq + WS_BIG32_OFF);
    hipLaunchKernelGGL(uds::k_pack_weight_frags32, dim3((total + 255) / 256), dim3(256), 0, st, p->ge_k, (int)(fe + h), (int)d, wq + WS_BIG32_OFF + WS_BIG32_LEN);
    const int ts = (int)(fe / 16) * (int)(h / 32) * 64;
    hipLaunchKernelGGL(uds::k_pack_weight_frags32, dim3((ts + 255) / 256), dim3(256), 0, st, p->xe_k, (int)fe, (int)h, wq + WS_SMALL32_OFF);
    hipLaunchKernelGGL(uds::k_pack_weight_frags32, dim3((ts + 255) / 256), dim3(256), 0, st, p->ex_k, (int)fx, (int)h, wq + WS_SMALL32_OFF + WS_SMALL32_LEN);
    if ((he = hipGetLastError()) != hipSuccess) return fail(UDS_EHIP, "uds_spatial_pack_weights: launch -> %s", hipGetErrorString(he));
  }
  return UDS_OK;
}

int uds_spatial_layer_forward(const uds_network_t *net, const uds_spatial_params_t *p, const float *x, int64_t fx,
                              const float *e, int64_t fe, int64_t S, int64_t h, int64_t d, int act, int flags, float *ws,
                              float *out_x, float *out_e, uds_stream_t stream) {
  return uds_spatial_layer_forward_split(net, p, x, fx, nullptr, 0, e, fe, nullptr, 0, S, h, d, act, flags, ws, out_x, out_e, stream);
}

static int spatial_forward_impl(const uds_network_t *net, const uds_spatial_params_t *p, const float *x, int64_t fxa,
                                const float *xb, int64_t fxb, const float *e, int64_t fea, const float *eb, int64_t feb,
                                const float *rem_x, const float *rem_e, int64_t S, int64_t h, int64_t d, int act, int flags,
                                float *ws, float *out_x, float *out_e, uds_stream_t stream);

int uds_spatial_layer_forward_split(const uds_network_t *net, const uds_spatial_params_t *p, const float *x, int64_t fxa,
                                    const float *xb, int64_t fxb, const float *e, int64_t fea, const float *eb, int64_t feb,
                                    int64_t S, int64_t h, int64_t d, int act, int flags, float *ws, float *out_x, float *out_e,
                                    uds_stream_t stream) {
  return spatial_forward_impl(net, p, x, fxa, xb, fxb, e, fea, eb, feb, nullptr, nullptr, S, h, d, act, flags, ws, out_x, out_e, stream);
}

int uds_spatial_layer_forward_rem(const uds_network_t *net, const uds_spatial_params_t *p, const float *x, int64_t fx,
                                  const float *e, int64_t fe, const float *rem_x, const float *rem_e, int64_t S, int64_t h,
                                  int64_t d, int act, int flags, float *ws, float *out_x, float *out_e, uds_stream_t stream) {
  UDS_REQUIRE(rem_x && rem_e, "uds_spatial_layer_forward_rem: NULL remainder");
  UDS_REQUIRE(aligned16(rem_x) && aligned16(rem_e), "uds_spatial_layer_forward_rem: rem_x / rem_e must be 16-byte aligned");
  return spatial_forward_impl(net, p, x, fx, nullptr, 0, e, fe, nullptr, 0, rem_x, rem_e, S, h, d, act, flags | UDS_FLAG_REQUIRE_FUSED, ws,
                              out_x, out_e, stream);
}

static int spatial_forward_impl(const uds_network_t *net, const uds_spatial_params_t *p, const float *x, int64_t fxa,
                                const float *xb, int64_t fxb, const float *e, int64_t fea, const float *eb, int64_t feb,
                                const float *rem_x, const float *rem_e, int64_t S, int64_t h, int64_t d, int act, int flags,
                                float *ws, float *out_x, float *out_e, uds_stream_t stream) {
  UDS_REQUIRE(net && p && x && e && ws && out_x && out_e, "uds_spatial_layer_forward: NULL argument");
  UDS_REQUIRE((xb != nullptr) == (fxb > 0) && (eb != nullptr) == (feb > 0), "uds_spatial_layer_forward_split: xb/fxb or eb/feb disagree");
  UDS_REQUIRE((!xb || (fxa == 64 && fxb == 32)) && (!eb || (fea == 64 && feb == 32)),
              "uds_spatial_layer_forward_split: a split input must be 64 + 32 columns (got %lld+%lld, %lld+%lld)", (long long)fxa,
              (long long)fxb, (long long)fea, (long long)feb);
  UDS_REQUIRE(aligned16(xb) && aligned16(eb), "uds_spatial_layer_forward_split: xb/eb must be 16-byte aligned");
  const int64_t fx = fxa + fxb, fe = fea + feb;
  if (xb || eb) flags |= UDS_FLAG_REQUIRE_FUSED;     // only the fused kernel reads split rows
  UDS_REQUIRE(p->xe_k && p->ex_k && p->ne_n_val && p->ne_e_val && p->gx_k && p->gx_as && p->gx_an && p->ge_k &&
                  p->ge_as && p->ge_an,
              "uds_spatial_layer_forward: NULL parameter tensor");
  UDS_REQUIRE(h > 0 && h % 4 == 0 && d > 0 && d % 4 == 0, "uds_spatial_layer_forward: h=%lld d=%lld must be multiples of 4",
              (long long)h, (long long)d);
  UDS_REQUIRE(act >= UDS_ACT_LINEAR && act <= UDS_ACT_HARD_SIGMOID, "uds_spatial_layer_forward: unknown activation %d", act);
  UDS_REQUIRE(out_x != x && out_e != e, "uds_spatial_layer_forward: outputs must not alias inputs");
  UDS_REQUIRE(aligned16(x) && aligned16(e) && aligned16(ws) && aligned16(out_x) && aligned16(out_e),
              "uds_spatial_layer_forward: x/e/workspace/outputs must be 16-byte aligned");
  const int64_t N = net->adj->n_rows, E = net->edge_adj->n_rows;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (S == 0) return UDS_OK;

  if (h == uds::F128_H && d == uds::F128_D && fx == 128 && (fe == 128 || fe == 64) && !xb && !eb && !(flags & UDS_FLAG_EXACT_FP32) &&
      net->slot[slot_index(128, (int)fe)].ok && net->slot[slot_index((int)fe, 128)].ok) {
    // d = 128: the column-split fused kernel (kernels_fused128.hpp); 64-wide link rows -> one launch per side
    UDS_REQUIRE(aligned16(p->xe_b) && aligned16(p->ex_b) && aligned16(p->gx_as) && aligned16(p->gx_an) && aligned16(p->gx_b) &&
                    aligned16(p->ge_as) && aligned16(p->ge_an) && aligned16(p->ge_b),
                "uds_spatial_layer_forward: bias / attention vectors must be 16-byte aligned");
    const uint4 *wq = reinterpret_cast<const uint4 *>(p->packed);
    if (!wq) {
      int rc = uds_spatial_pack_weights(p, fx, fe, h, d, ws, stream);
      if (rc != UDS_OK) return rc;
      wq = reinterpret_cast<const uint4 *>(ws);
    } else {
      UDS_REQUIRE(aligned16(wq), "uds_spatial_layer_forward: packed weights must be 16-byte aligned");
    }
    uds::FusedArgs a;
    a.blocks = nullptr;
    a.side[0] = uds::FusedSide{x, e, nullptr, nullptr, out_x, wq, wq + 2048, nullptr, nullptr, p->xe_b, p->gx_as, p->gx_an, p->gx_b, p->ne_n_val, (int)N, (int)E, rem_x};
    a.side[1] = uds::FusedSide{e, x, nullptr, nullptr, out_e, wq + 8192, wq + 10240, nullptr, nullptr, p->ex_b, p->ge_as, p->ge_an, p->ge_b, p->ne_e_val, (int)E, (int)N, rem_e};
    a.S = (int)S;
    a.act = act;
    a.dbg = nullptr;
#ifdef UDS_PHASE_TIMING
    a.dbg = reinterpret_cast<unsigned long long *>(ws + PACKED_WEIGHT_FLOATS);
#endif
    auto launch_side = [&](const uds_plan_slot &u, int side, auto FP_, auto FS_) {       // side < 0: both sides in one launch
      a.hdr = side < 0 ? u.d_hdr : u.d_hdr_side[side];
      a.pool = u.d_pool;
      a.n_tiles = side < 0 ? u.plan.n_tiles : u.plan.side[side].n_tiles;
      a.p_cap = u.plan.p_cap;
      a.q_cap = u.plan.q_cap;
      a.meta_cap = u.plan.meta_cap;
      a.side_mask = side < 0 ? 3 : (1 << side);
      if ((rem_x || rem_e) && a.p_cap > 64) return hipErrorInvalidValue;      // one P1.5 unit per wave (kernels_fused128.hpp)
      int64_t chunk = S, best = INT64_MAX;
      for (int64_t c = 1; c <= S; ++c) {
        const int64_t rounds = (((S + c - 1) / c) * a.n_tiles + 255) / 256;
        const int64_t cost = rounds * (3 + 2 * c);
        if (cost < best || (cost == best && c < chunk)) {
          best = cost;
          chunk = c;
        }
      }
      a.chunk = (int)chunk;
      const int grid = (int)(((S + chunk - 1) / chunk) * a.n_tiles);
      return launch_fused128<decltype(FP_)::value, decltype(FS_)::value>(a, grid, u.lds_bytes, st);
    };
    using I64 = std::integral_constant<int, 64>;
    using I128 = std::integral_constant<int, 128>;
    hipError_t he;
    if (fe == 128) {
      he = launch_side(net->slot[4], -1, I128{}, I128{});
    } else {
      he = launch_side(net->slot[5], 0, I128{}, I64{});                       // node tiles: 128-wide nodes fed by 64-wide links
      if (he == hipSuccess) he = launch_side(net->slot[6], 1, I64{}, I128{});   // link tiles: 64-wide links fed by 128-wide nodes
    }
    if (he != hipSuccess) return fail(UDS_EHIP, "uds_spatial_layer_forward: fused d=128 launch -> %s", hipGetErrorString(he));
    return UDS_OK;
  }
  UDS_REQUIRE((!rem_x && !rem_e) || (h == uds::FUSED_H && d == uds::FUSED_D && fxa == 64 && fea == 64),
              "uds_spatial_layer_forward_rem: the remainder goes into the d = 128 fused kernel or the 64-wide d = 64 one (fx=%lld fe=%lld h=%lld d=%lld, plan %d)",
              (long long)fxa, (long long)fea, (long long)h, (long long)d, (int)(net->slot[4].ok));
  // (rows of one snapshot are addressed with a 32-bit byte offset from a per-snapshot base: N, E < 2^31 / 384)
  const bool shape_ok = h == uds::FUSED_H && d == uds::FUSED_D && (fx == 64 || fx == 96) && (fe == 64 || fe == 96) &&
                        std::max(N, E) * 384 < ((int64_t)1 << 31);
  // node tiles run the variant <fx, fe>, link tiles <fe, fx>: one launch when they coincide, else one per side
  const uds_plan_slot &sl = net->slot[slot_index((int)fx, (int)fe)];
  const uds_plan_slot &sl_link = net->slot[slot_index((int)fe, (int)fx)];
  if (flags & UDS_FLAG_REQUIRE_FUSED)
    UDS_REQUIRE(sl.ok && sl_link.ok && shape_ok && !(flags & UDS_FLAG_EXACT_FP32),
                "uds_spatial_layer_forward: fused kernel unavailable (plan %d, fx=%lld fe=%lld h=%lld d=%lld)", (int)sl.ok,
                (long long)fx, (long long)fe, (long long)h, (long long)d);
  if (sl.ok && sl_link.ok && shape_ok && !(flags & UDS_FLAG_EXACT_FP32)) {
    UDS_REQUIRE(aligned16(p->xe_b) && aligned16(p->ex_b) && aligned16(p->gx_as) && aligned16(p->gx_an) && aligned16(p->gx_b) &&
                    aligned16(p->ge_as) && aligned16(p->ge_an) && aligned16(p->ge_b),
                "uds_spatial_layer_forward: bias / attention vectors must be 16-byte aligned");
    hipError_t he;
    const uint4 *wq = reinterpret_cast<const uint4 *>(p->packed);
    if (!wq) {   // no pre-packed weights: split them now into the head of the workspace
      int rc = uds_spatial_pack_weights(p, fx, fe, h, d, ws, stream);
      if (rc != UDS_OK) return rc;
      wq = reinterpret_cast<const uint4 *>(ws);
    } else {
      UDS_REQUIRE(aligned16(wq), "uds_spatial_layer_forward: packed weights must be 16-byte aligned");
    }
    const uint4 *w_small_n = wq, *w_big_n = wq + 768, *w_small_e = wq + 768 + 2048, *w_big_e = wq + 2 * 768 + 2048;
    uds::FusedArgs a;
    a.blocks = nullptr;
    const bool has32 = fx == 64 && fe == 64;
    a.side[0] = uds::FusedSide{x, e, xb, eb, out_x, w_small_n, w_big_n, has32 ? wq + WS_BIG32_OFF : nullptr, has32 ? wq + WS_SMALL32_OFF : nullptr, p->xe_b, p->gx_as, p->gx_an, p->gx_b, p->ne_n_val, (int)N, (int)E, rem_x};
    a.side[1] = uds::FusedSide{e, x, eb, xb, out_e, w_small_e, w_big_e, has32 ? wq + WS_BIG32_OFF + WS_BIG32_LEN : nullptr, has32 ? wq + WS_SMALL32_OFF + WS_SMALL32_LEN : nullptr, p->ex_b, p->ge_as, p->ge_an, p->ge_b, p->ne_e_val, (int)E, (int)N, rem_e};
    int64_t lds_need = 0;
    auto use_plan = [&](const uds_plan_slot &u, int side) {     // side < 0: both sides (merged tile list), else that side's tiles only
      a.hdr = side < 0 ? u.d_hdr : u.d_hdr_side[side];
      a.pool = u.d_pool;
      a.n_tiles = side < 0 ? u.plan.n_tiles : u.plan.side[side].n_tiles;
      a.p_cap = u.plan.p_cap;
      a.q_cap = u.plan.q_cap;
      a.meta_cap = u.blk_cap;      // LDS ints of the tile block (header + fixed-width index lists)
      a.blocks = side < 0 ? u.d_blocks : u.d_blocks_side[side];
      lds_need = u.lds_bytes;
    };
    use_plan(sl, -1);
    a.S = (int)S;
    a.act = act;
    a.dbg = nullptr;
#ifdef UDS_PHASE_TIMING
    a.dbg = reinterpret_cast<unsigned long long *>(ws + PACKED_WEIGHT_FLOATS);   // diagnostic build: stamps go to the (otherwise unused) workspace
#endif
    // Snapshots are cut into chunks; one workgroup = (tile, chunk).  All working workgroups take about the same time
    // (setup ~1.5 snapshots' worth + its snapshots), one per CU at a time, so a launch lasts ~ceil(working / 256) rounds:
    // pick the chunk length that minimises rounds * (setup + chunk).  (Metadata, weights and the DMA pipeline are set
    // up once per workgroup, so long chunks are cheap; a launch for one side only skips the other side's tiles at once.)
    auto set_chunk = [&](int64_t working_tiles) {
      int64_t chunk = S, best = INT64_MAX;
      for (int64_t c = 1; c <= S; ++c) {
        const int64_t n_c = (S + c - 1) / c;
        const int64_t rounds = (n_c * working_tiles + 255) / 256;
        const int64_t cost = rounds * (3 + 2 * c);          // in half snapshots
        if (cost < best || (cost == best && c < chunk)) {
          best = cost;
          chunk = c;
        }
      }
#ifdef UDS_KNOBS
      if (const char *ov = std::getenv("UDS_CHUNK")) chunk = std::max<int64_t>(1, std::min<int64_t>(S, std::atoll(ov)));   // experiment builds only (UDS_DEFINES=-DUDS_KNOBS)
#endif
      a.chunk = (int)chunk;
      return (int)(((S + chunk - 1) / chunk) * a.n_tiles);
    };
    if (fx == fe) {
      a.side_mask = 3;
      const int grid = set_chunk(a.n_tiles);
      const int64_t ws_lds = uds::fused_ws_lds_bytes(a.p_cap, a.q_cap, a.meta_cap);
      bool ws = fx == 64 && !xb && !eb && ws_lds <= FUSED_LDS_BUDGET && a.p_cap <= 128 && a.q_cap <= 256;
#ifdef UDS_NO_WS
      ws = false;
#endif
      UDS_REQUIRE(ws || (!rem_x && !rem_e), "uds_spatial_layer_forward_rem: the wave-specialised d = 64 kernel cannot take this network's plan (p_cap %d, q_cap %d, %lld B of LDS)",
                  a.p_cap, a.q_cap, (long long)ws_lds);
      if (ws) he = launch_fused_ws(a, grid, ws_lds, st);
      else he = (fx == 64) ? launch_fused<64, 64>(a, grid, lds_need, st) : launch_fused<96, 96>(a, grid, lds_need, st);
    } else {   // node tiles: FP = fx, FS = fe; link tiles: FP = fe, FS = fx -> one launch per side
      a.side_mask = 1;
      use_plan(sl, 0);      // workgroups go round-robin to the XCDs: a grid of working tiles only keeps the XCDs level
      int grid = set_chunk(a.n_tiles);
      he = (fx == 64) ? launch_fused<64, 96>(a, grid, lds_need, st) : launch_fused<96, 64>(a, grid, lds_need, st);
      if (he == hipSuccess) {
        a.side_mask = 2;
        use_plan(sl_link, 1);
        grid = set_chunk(a.n_tiles);
        he = (fe == 64) ? launch_fused<64, 96>(a, grid, lds_need, st) : launch_fused<96, 64>(a, grid, lds_need, st);
      }
    }
    if (he != hipSuccess) return fail(UDS_EHIP, "uds_spatial_layer_forward: fused launch -> %s", hipGetErrorString(he));
    return UDS_OK;
  }

  float *base = ws + PACKED_WEIGHT_FLOATS;
  float *x_e = base;                  // (S,E,h)
  float *e_x = x_e + S * E * h;       // (S,N,h)
  float *agg_n = e_x + S * N * h;     // (S,N,h)
  float *agg_e = agg_n + S * N * h;   // (S,E,h)
  float *gat_ws_n = agg_e + S * E * h;        // S*N*(d+2)
  float *gat_ws_e = gat_ws_n + align4(S * N * (d + 2));
  int rc;
  if ((rc = uds_dense_act(e, fe, nullptr, 0, S * E, p->xe_k, p->xe_b, h, act, nullptr, nullptr, x_e, nullptr, nullptr, stream))) return rc;
  if ((rc = uds_dense_act(x, fx, nullptr, 0, S * N, p->ex_k, p->ex_b, h, act, nullptr, nullptr, e_x, nullptr, nullptr, stream))) return rc;
  if ((rc = uds_csr_spmm(net->inc_n, p->ne_n_val, x_e, S, h, nullptr, UDS_ACT_LINEAR, agg_n, stream))) return rc;
  if ((rc = uds_csr_spmm(net->inc_e, p->ne_e_val, e_x, S, h, nullptr, UDS_ACT_LINEAR, agg_e, stream))) return rc;
  if ((rc = uds_gat_forward(net->adj, x, fx, agg_n, h, S, p->gx_k, p->gx_as, p->gx_an, p->gx_b, d, act, gat_ws_n, out_x, stream))) return rc;
  if ((rc = uds_gat_forward(net->edge_adj, e, fe, agg_e, h, S, p->ge_k, p->ge_as, p->ge_an, p->ge_b, d, act, gat_ws_e, out_e, stream))) return rc;
  return UDS_OK;
}

}  // extern "C"
