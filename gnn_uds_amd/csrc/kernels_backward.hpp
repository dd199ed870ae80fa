// Reverse-mode kernels of the sparse operators (gfx950) -- the training step, emulator.py:457-484 (GradientTape over
// the layers of emulator.py:18-45,225-230).  The dense parts of the backward pass are row GEMMs (the forward kernels
// with transposed weights) and plain weight-gradient GEMMs; what is specific to the graph operators lives here:
//
//   k_gat_bwd_rows : softmax / leaky-relu backward of GATConv along the rows of the pattern (CSR walk)
//   k_gat_bwd_cols : gradient of the transformed features, gathered along the columns (CSR of the transpose)
//   k_csr_sddmm    : gradient of a per-entry support weight (NodeEdge.weight on its support)
//
// Thread mapping as in kernels_sparse.hpp: no atomics, deterministic results.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels_sparse.hpp"

namespace uds {

// Forward (per snapshot): l_ij = leaky(s_self_i + s_nbr_j), alpha_ij = softmax_j(l_ij), pre_i = sum_j alpha_ij hx_j.
// Given g_i = dL/dpre_i:   q_ij = <g_i, hx_j>,  c_i = sum_j alpha_ij q_ij,
//   dl_ij = alpha_ij (q_ij - c_i),  de_ij = dl_ij * leaky'(s_self_i + s_nbr_j),  ds_self_i = sum_j de_ij.
// Writes alpha and de per pattern entry (S x nnz each) for the column pass.
struct GatBwdRowsArgs {
  const int32_t *rowptr, *col;
  const float *g, *hx, *s_self, *s_nbr;
  float *alpha, *de, *ds_self;
  int n, d4, S, G;        // G lanes per row (power of two, <= 64, <= d4 rounded down)
  int64_t nnz;
  // attention dropout (k_gat_aggregate_coef): pre_i = sum_j alpha_ij m_ij hx_j with m (S, nnz) = 0 or 1 / (1 - rate), else NULL.
  // Then q_ij = m_ij <g_i, hx_j> in the formulas above, and the alpha handed to the column pass is alpha_ij m_ij.
  const float *coef;
};

__global__ __launch_bounds__(256) void k_gat_bwd_rows(GatBwdRowsArgs a) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int lg = (int)(t % a.G);
  const int64_t r = t / a.G;
  const bool row_ok = r < a.n;
  const int i = (int)(row_ok ? r : a.n - 1);      // surplus groups shadow the last row (they take part in the shuffles)
  const int s = blockIdx.y;
  const int beg = a.rowptr[i], end = a.rowptr[i + 1];
  const float *sn = a.s_nbr + (int64_t)s * a.n;
  const float ss = a.s_self[(int64_t)s * a.n + i];
  const float4 *g4 = reinterpret_cast<const float4 *>(a.g) + ((int64_t)s * a.n + i) * a.d4;
  const float4 *hx4 = reinterpret_cast<const float4 *>(a.hx) + (int64_t)s * a.n * a.d4;
  float *al = a.alpha + (int64_t)s * a.nnz, *de = a.de + (int64_t)s * a.nnz;
  const float *cf = a.coef ? a.coef + (int64_t)s * a.nnz : nullptr;
  float m = -INFINITY;
  for (int p = beg; p < end; ++p) m = fmaxf(m, leaky02(ss + sn[a.col[p]]));
  float den = 0.f, cn = 0.f;
  if (end - beg <= a.G) {
    // the usual case (degree <= lanes per row): lane k of the row keeps entry k's (exp, q, logit sign) in registers, the
    // row sums are known to every lane after the loop, and the lanes write alpha / de of their entries side by side --
    // no second walk through memory by one lane.  Same operation order per value as the general path below.
    float wk = 0.f, qk = 0.f;
    bool pos = false;
    for (int p = beg; p < end; ++p) {
      const int j = a.col[p];
      const float lgt = ss + sn[j];
      const float w = expf(leaky02(lgt) - m);
      float q = 0.f;
      for (int c = lg; c < a.d4; c += a.G) {
        const float4 gv = g4[c], hv = hx4[(int64_t)j * a.d4 + c];
        q = fmaf(gv.x, hv.x, fmaf(gv.y, hv.y, fmaf(gv.z, hv.z, fmaf(gv.w, hv.w, q))));
      }
      for (int o = a.G >> 1; o > 0; o >>= 1) q += __shfl_xor(q, o);
      if (cf) q *= cf[p];
      den += w;
      cn = fmaf(w, q, cn);
      if (p - beg == lg) {
        wk = w;
        qk = q;
        pos = lgt > 0.0f;
      }
    }
    const float inv = end > beg ? 1.0f / den : 0.0f;
    const float cbar = cn * inv;
    const bool mine = lg < end - beg;
    const float w = wk * inv;
    const float dl = w * (qk - cbar);
    const float dv = mine ? (pos ? dl : 0.2f * dl) : 0.f;
    if (mine && row_ok) {
      al[beg + lg] = cf ? w * cf[beg + lg] : w;
      de[beg + lg] = dv;
    }
    // ds_self = sum of the row's de in entry order (the order the one-lane walk below adds them in)
    float dss = 0.f;
    for (int k = 0; k < end - beg; ++k) dss += __shfl(dv, k, a.G);
    if (lg == 0 && row_ok) a.ds_self[(int64_t)s * a.n + i] = dss;
    return;
  }
  for (int p = beg; p < end; ++p) {
    const int j = a.col[p];
    const float w = expf(leaky02(ss + sn[j]) - m);
    float q = 0.f;
    for (int c = lg; c < a.d4; c += a.G) {
      const float4 gv = g4[c], hv = hx4[(int64_t)j * a.d4 + c];
      q = fmaf(gv.x, hv.x, fmaf(gv.y, hv.y, fmaf(gv.z, hv.z, fmaf(gv.w, hv.w, q))));
    }
    for (int o = a.G >> 1; o > 0; o >>= 1) q += __shfl_xor(q, o);
    if (cf) q *= cf[p];
    den += w;
    cn = fmaf(w, q, cn);
    if (lg == 0 && row_ok) {
      al[p] = w;
      de[p] = q;
    }
  }
  if (lg != 0 || !row_ok) return;
  const float inv = end > beg ? 1.0f / den : 0.0f;
  const float cbar = cn * inv;
  float dss = 0.f;
  for (int p = beg; p < end; ++p) {       // same lane wrote al / de above
    const float w = al[p] * inv;
    const float dl = w * (de[p] - cbar);
    const float dv = ss + sn[a.col[p]] > 0.0f ? dl : 0.2f * dl;
    al[p] = cf ? w * cf[p] : w;
    de[p] = dv;
    dss += dv;
  }
  a.ds_self[(int64_t)s * a.n + i] = dss;
}

// Grouped variant (see kernels_sparse.hpp): the G lanes of a row load G entries' indices and scores side by side, each
// lane ends up holding (exp, q, sign) of "its" entry, and alpha / de are written side by side.  Rows of more than G entries
// park the raw (exp, q) pairs in alpha / de per chunk and finish them in a second walk (same lane wrote them).
template <int G, int NC>
__global__ __launch_bounds__(256) void k_gat_bwd_rows_g(GatBwdRowsArgs a) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int c = (int)(t % G);
  const int64_t r = t / G;
  const bool row_ok = r < a.n;
  const int i = (int)(row_ok ? r : a.n - 1);
  const int s = blockIdx.y;
  const int beg = a.rowptr[i], end = a.rowptr[i + 1];
  const float *sn = a.s_nbr + (int64_t)s * a.n;
  const float ss = a.s_self[(int64_t)s * a.n + i];
  const float4 *g4 = reinterpret_cast<const float4 *>(a.g) + ((int64_t)s * a.n + i) * a.d4 + c;
  const float4 *hx4 = reinterpret_cast<const float4 *>(a.hx) + (int64_t)s * a.n * a.d4 + c;
  float *al = a.alpha + (int64_t)s * a.nnz, *de = a.de + (int64_t)s * a.nnz;
  float4 gv[NC];
#pragma unroll
  for (int q = 0; q < NC; ++q) gv[q] = g4[G * q];
  float m = -INFINITY, l0 = 0.f;
  int j0 = 0;
  for (int b0 = beg; b0 < end; b0 += G) {
    const int p = b0 + c;
    const int j = a.col[min(p, end - 1)];
    const float lgt = ss + sn[j];
    if (b0 == beg) {
      j0 = j;
      l0 = lgt;
    }
    m = fmaxf(m, p < end ? leaky02(lgt) : -INFINITY);
  }
  m = group_max<G>(m);
  const bool single = end - beg <= G;
  float den = 0.f, cn = 0.f, wk = 0.f, qk = 0.f, lk = 0.f;
  for (int b0 = beg; b0 < end; b0 += G) {
    const int p = b0 + c;
    int j = j0;
    float lgt = l0;
    if (b0 != beg) {
      j = a.col[min(p, end - 1)];
      lgt = ss + sn[j];
    }
    const float w = p < end ? expf(leaky02(lgt) - m) : 0.f;
    const int nk = min(G, end - b0);
    float qm = 0.f;
    for (int k0 = 0; k0 < nk; k0 += GU) {
      float ww[GU], qq[GU];
      float4 hv[GU][NC];
#pragma unroll
      for (int u = 0; u < GU; ++u) {
        const int k = min(k0 + u, nk - 1);
        const int jj = __shfl(j, k, G);
        ww[u] = __shfl(w, k, G);
#pragma unroll
        for (int v = 0; v < NC; ++v) hv[u][v] = hx4[(int64_t)jj * a.d4 + G * v];
      }
#pragma unroll
      for (int u = 0; u < GU; ++u) {
        float q = 0.f;
#pragma unroll
        for (int v = 0; v < NC; ++v)      // the one-lane-per-chunk kernel adds a lane's chunks c, c + G, .. in this order
          q = fmaf(gv[v].x, hv[u][v].x, fmaf(gv[v].y, hv[u][v].y, fmaf(gv[v].z, hv[u][v].z, fmaf(gv[v].w, hv[u][v].w, q))));
        qq[u] = q;
      }
#pragma unroll
      for (int o = G >> 1; o > 0; o >>= 1)
#pragma unroll
        for (int u = 0; u < GU; ++u) qq[u] += __shfl_xor(qq[u], o);
#pragma unroll
      for (int u = 0; u < GU; ++u)
        if (k0 + u < nk) {
          den += ww[u];
          cn = fmaf(ww[u], qq[u], cn);
          if (k0 + u == c) qm = qq[u];
        }
    }
    if (single) {
      wk = w;
      qk = qm;
      lk = lgt;
    } else if (p < end && row_ok) {
      al[p] = w;
      de[p] = qm;
    }
  }
  const float inv = end > beg ? 1.0f / den : 0.0f;
  const float cbar = cn * inv;
  if (single) {
    const bool mine = c < end - beg;
    const float w = wk * inv;
    const float dl = w * (qk - cbar);
    const float dv = mine ? (lk > 0.0f ? dl : 0.2f * dl) : 0.f;
    if (mine && row_ok) {
      al[beg + c] = w;
      de[beg + c] = dv;
    }
    float dss = 0.f;
    for (int k = 0; k < end - beg; ++k) dss += __shfl(dv, k, G);      // entry order, as the one-lane walk adds them
    if (c == 0 && row_ok) a.ds_self[(int64_t)s * a.n + i] = dss;
    return;
  }
  float dss = 0.f;
  for (int b0 = beg; b0 < end; b0 += G) {
    const int p = b0 + c;
    float dv = 0.f;
    if (p < end && row_ok) {
      const float w = al[p] * inv;
      const float dl = w * (de[p] - cbar);
      dv = ss + sn[a.col[p]] > 0.0f ? dl : 0.2f * dl;
      al[p] = w;
      de[p] = dv;
    }
    dss += dv;
  }
#pragma unroll
  for (int o = G >> 1; o > 0; o >>= 1) dss += __shfl_xor(dss, o);
  if (c == 0 && row_ok) a.ds_self[(int64_t)s * a.n + i] = dss;
}

inline hipError_t launch_gat_bwd_rows(const GatBwdRowsArgs &a, hipStream_t st) {
  int G, NC;
  group_shape(a.d4, G, NC);
  if (G && a.n > 0 && !a.coef)      // (attention dropout: the plain kernel below)
    return launch_grouped(a, a.n, a.S, a.d4, st, [&](auto g_, auto nc_, dim3 grid) {
      hipLaunchKernelGGL((k_gat_bwd_rows_g<decltype(g_)::value, decltype(nc_)::value>), grid, dim3(256), 0, st, a);
    });
  const int64_t total = (int64_t)a.n * a.G;
  hipLaunchKernelGGL(k_gat_bwd_rows, dim3((unsigned)((total + 255) / 256), (unsigned)a.S), dim3(256), 0, st, a);
  return hipGetLastError();
}

// d_hx_j = sum_{i : j in row i} alpha_ij g_i + a_nbr * ds_nbr_j + a_self * ds_self_j,   ds_nbr_j = sum_i de_ij.
// (rowptr_t, col_t) = CSR of the transposed pattern; perm_t[p] = position of that entry in the row-major arrays.
struct GatBwdColsArgs {
  const int32_t *rowptr_t, *col_t, *perm_t;
  const float *g, *alpha, *de, *ds_self, *a_self, *a_nbr;
  float *d_hx, *ds_nbr;
  int n, d4, S;
  int64_t nnz;
};

__global__ __launch_bounds__(256) void k_gat_bwd_cols(GatBwdColsArgs a) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= (int64_t)a.n * a.d4) return;
  const int s = blockIdx.y;
  const int c = (int)(t % a.d4), j = (int)(t / a.d4);
  const float4 *g4 = reinterpret_cast<const float4 *>(a.g) + (int64_t)s * a.n * a.d4 + c;
  const float *al = a.alpha + (int64_t)s * a.nnz, *de = a.de + (int64_t)s * a.nnz;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  float dsn = 0.f;
  for (int p = a.rowptr_t[j]; p < a.rowptr_t[j + 1]; ++p) {
    const int i = a.col_t[p], k = a.perm_t[p];
    const float w = al[k];
    const float4 gv = g4[(int64_t)i * a.d4];
    acc.x = fmaf(w, gv.x, acc.x);
    acc.y = fmaf(w, gv.y, acc.y);
    acc.z = fmaf(w, gv.z, acc.z);
    acc.w = fmaf(w, gv.w, acc.w);
    dsn += de[k];
  }
  const float dss = a.ds_self[(int64_t)s * a.n + j];
  const float4 as = reinterpret_cast<const float4 *>(a.a_self)[c], an = reinterpret_cast<const float4 *>(a.a_nbr)[c];
  acc.x = fmaf(an.x, dsn, fmaf(as.x, dss, acc.x));
  acc.y = fmaf(an.y, dsn, fmaf(as.y, dss, acc.y));
  acc.z = fmaf(an.z, dsn, fmaf(as.z, dss, acc.z));
  acc.w = fmaf(an.w, dsn, fmaf(as.w, dss, acc.w));
  reinterpret_cast<float4 *>(a.d_hx)[((int64_t)s * a.n + j) * a.d4 + c] = acc;
  if (c == 0) a.ds_nbr[(int64_t)s * a.n + j] = dsn;
}

template <int G, int NC>
__global__ __launch_bounds__(256) void k_gat_bwd_cols_g(GatBwdColsArgs a) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int c = (int)(t % G);
  const int64_t r = t / G;
  const bool row_ok = r < a.n;
  const int j = (int)(row_ok ? r : a.n - 1);
  const int s = blockIdx.y;
  const float4 *g4 = reinterpret_cast<const float4 *>(a.g) + (int64_t)s * a.n * a.d4 + c;
  const float *al = a.alpha + (int64_t)s * a.nnz, *de = a.de + (int64_t)s * a.nnz;
  float4 acc[NC];
#pragma unroll
  for (int q = 0; q < NC; ++q) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
  float dsn = 0.f;
  const int beg = a.rowptr_t[j], end = a.rowptr_t[j + 1];
  for (int b0 = beg; b0 < end; b0 += G) {
    const int p = min(b0 + c, end - 1);
    const int i = a.col_t[p], k = a.perm_t[p];
    const float w = al[k], d = de[k];
    const int nk = min(G, end - b0);
    for (int e0 = 0; e0 < nk; e0 += GU) {
      float ww[GU], dd[GU];
      float4 gv[GU][NC];
#pragma unroll
      for (int u = 0; u < GU; ++u) {
        const int e = min(e0 + u, nk - 1);
        const int ii = __shfl(i, e, G);
        ww[u] = __shfl(w, e, G);
        dd[u] = __shfl(d, e, G);
#pragma unroll
        for (int q = 0; q < NC; ++q) gv[u][q] = g4[(int64_t)ii * a.d4 + G * q];
      }
#pragma unroll
      for (int u = 0; u < GU; ++u)
        if (e0 + u < nk) {
          dsn += dd[u];
#pragma unroll
          for (int q = 0; q < NC; ++q) {
            acc[q].x = fmaf(ww[u], gv[u][q].x, acc[q].x);
            acc[q].y = fmaf(ww[u], gv[u][q].y, acc[q].y);
            acc[q].z = fmaf(ww[u], gv[u][q].z, acc[q].z);
            acc[q].w = fmaf(ww[u], gv[u][q].w, acc[q].w);
          }
        }
    }
  }
  if (!row_ok) return;
  const float dss = a.ds_self[(int64_t)s * a.n + j];
#pragma unroll
  for (int q = 0; q < NC; ++q) {
    const float4 as = reinterpret_cast<const float4 *>(a.a_self)[c + G * q], an = reinterpret_cast<const float4 *>(a.a_nbr)[c + G * q];
    float4 o;
    o.x = fmaf(an.x, dsn, fmaf(as.x, dss, acc[q].x));
    o.y = fmaf(an.y, dsn, fmaf(as.y, dss, acc[q].y));
    o.z = fmaf(an.z, dsn, fmaf(as.z, dss, acc[q].z));
    o.w = fmaf(an.w, dsn, fmaf(as.w, dss, acc[q].w));
    reinterpret_cast<float4 *>(a.d_hx)[((int64_t)s * a.n + j) * a.d4 + c + G * q] = o;
  }
  if (c == 0) a.ds_nbr[(int64_t)s * a.n + j] = dsn;
}

inline hipError_t launch_gat_bwd_cols(const GatBwdColsArgs &a, hipStream_t st) {
  int G, NC;
  group_shape(a.d4, G, NC);
  if (G && a.n > 0)
    return launch_grouped(a, a.n, a.S, a.d4, st, [&](auto g_, auto nc_, dim3 grid) {
      hipLaunchKernelGGL((k_gat_bwd_cols_g<decltype(g_)::value, decltype(nc_)::value>), grid, dim3(256), 0, st, a);
    });
  const int64_t total = (int64_t)a.n * a.d4;
  hipLaunchKernelGGL(k_gat_bwd_cols, dim3((unsigned)((total + 255) / 256), (unsigned)a.S), dim3(256), 0, st, a);
  return hipGetLastError();
}

// out[k] = sum_s <a[s, row(k), :], b[s, col(k), :]> for every pattern entry k: the gradient of a per-entry weight of
// out = A(val) @ x  (a = dL/dout, b = x).  G lanes per entry walk the snapshots.
struct SddmmArgs {
  const int32_t *row, *col;
  const float *a, *b;
  float *out;
  int n_rows, n_cols, f4, S, G;
  int64_t nnz;
};

__global__ __launch_bounds__(256) void k_csr_sddmm(SddmmArgs a) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int lg = (int)(t % a.G);
  const int64_t e = t / a.G;
  const bool ok = e < a.nnz;
  const int64_t k = ok ? e : a.nnz - 1;
  const int r = a.row[k], c = a.col[k];
  const float4 *a4 = reinterpret_cast<const float4 *>(a.a) + (int64_t)r * a.f4;
  const float4 *b4 = reinterpret_cast<const float4 *>(a.b) + (int64_t)c * a.f4;
  float q = 0.f;
  for (int s = 0; s < a.S; ++s) {
    const float4 *as = a4 + (int64_t)s * a.n_rows * a.f4, *bs = b4 + (int64_t)s * a.n_cols * a.f4;
    for (int f = lg; f < a.f4; f += a.G) {
      const float4 av = as[f], bv = bs[f];
      q = fmaf(av.x, bv.x, fmaf(av.y, bv.y, fmaf(av.z, bv.z, fmaf(av.w, bv.w, q))));
    }
  }
  for (int o = a.G >> 1; o > 0; o >>= 1) q += __shfl_xor(q, o);
  if (lg == 0 && ok) a.out[k] = q;
}

inline hipError_t launch_csr_sddmm(const SddmmArgs &a, hipStream_t st) {
  const int64_t total = a.nnz * a.G;
  hipLaunchKernelGGL(k_csr_sddmm, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a);
  return hipGetLastError();
}

inline int lanes_per_item(int f4) {     // largest power of two <= min(f4, 16)
  int g = 1;
  while (g * 2 <= f4 && g * 2 <= 16) g *= 2;
  return g;
}

}  // namespace uds
