"""Sparse-CSR CPU oracle (TEST INFRASTRUCTURE ONLY; PARITY UNPINNED, see oracle/__init__.py).

Second, independent restatement of the same layers as oracle/spektral_dense.py,
in the gather / segmented-softmax / scatter-add form the HIP kernels use.  The
two must agree (tests/test_oracle_dual.py): fp64 to 1e-12, fp32 to 1e-5.
It is also the form timed as `cpu_baseline` on graphs too large for the dense
`(N,N)` formulation.
"""
import numpy as np
import torch

from .spektral_dense import LEAKY_SLOPE, activation, dense


def csr_from_dense(a, add_self_loops=False):
    """Row-major CSR of the non-zero pattern of a dense (R,C) matrix; with
    add_self_loops the diagonal is forced in first (spektral: set_diag(a, 1)).
    Columns ascend inside a row.  Returns int32 (rowptr, col) and the values."""
    a = np.array(a, dtype=np.float64, copy=True)
    if add_self_loops:
        np.fill_diagonal(a, 1.0)
    rows, cols = np.nonzero(a)
    rowptr = np.zeros(a.shape[0] + 1, dtype=np.int64)
    np.add.at(rowptr, rows + 1, 1)
    rowptr = np.cumsum(rowptr)
    return rowptr.astype(np.int32), cols.astype(np.int32), a[rows, cols]


def _rows_of(rowptr):
    rowptr = np.asarray(rowptr, dtype=np.int64)
    return np.repeat(np.arange(len(rowptr) - 1), np.diff(rowptr))


def gat_conv_csr(x, rowptr, col, kernel, attn_self, attn_neigh, bias, act='relu'):
    """Single-head GATConv on a CSR pattern that already holds the self loops.

    x:(S,N,F); kernel (F,1,C) or (F,C); attn_* (C,1,1) or (C,).  For every row i:
    alpha_ij = softmax_j(leaky_relu(s_self_i + s_nbr_j)) over j in row i,
    out_i = act(sum_j alpha_ij hx_j + bias).  Masked (absent) entries of the dense
    form contribute exp(-1e10 - max) == 0 exactly, so they are simply not visited.
    """
    k2 = kernel.reshape(kernel.shape[0], -1)
    a_s = attn_self.reshape(-1)
    a_n = attn_neigh.reshape(-1)
    hx = x @ k2                                             # (S,N,C)
    s_self = hx @ a_s                                       # (S,N)
    s_nbr = hx @ a_n
    rows = torch.from_numpy(_rows_of(rowptr))
    cols = torch.from_numpy(np.asarray(col, dtype=np.int64))
    n = len(rowptr) - 1
    logit = torch.nn.functional.leaky_relu(s_self[:, rows] + s_nbr[:, cols], LEAKY_SLOPE)   # (S,nnz)
    mx = torch.full((x.shape[0], n), -float('inf'), dtype=x.dtype)
    mx = mx.scatter_reduce(1, rows.expand(x.shape[0], -1), logit, reduce='amax')
    ex = torch.exp(logit - mx[:, rows])
    den = torch.zeros((x.shape[0], n), dtype=x.dtype).index_add_(1, rows, ex)
    alpha = ex / den[:, rows]
    out = torch.zeros_like(hx).index_add_(1, rows, alpha[..., None] * hx[:, cols])
    if bias is not None:
        out = out + bias
    return activation(act)(out)


def gcn_conv_csr(x, rowptr, col, val, kernel, bias, act='relu'):
    """GCNConv with the normalised filter given as CSR values."""
    hx = x @ kernel.reshape(kernel.shape[0], -1)
    rows = torch.from_numpy(_rows_of(rowptr))
    cols = torch.from_numpy(np.asarray(col, dtype=np.int64))
    v = torch.as_tensor(val, dtype=x.dtype)
    out = torch.zeros_like(hx).index_add_(1, rows, v[None, :, None] * hx[:, cols])
    if bias is not None:
        out = out + bias
    return activation(act)(out)


def incidence_aggregate_csr(x, rowptr, col, val, n_rows):
    """out[s,r,:] = sum_p val[p] * x[s,col[p],:] over the CSR row r: NodeEdge on its
    support, val[p] = w[r,c]*inci[r,c] + b[r,c] (`emulator.py:42-45`)."""
    rows = torch.from_numpy(_rows_of(rowptr))
    cols = torch.from_numpy(np.asarray(col, dtype=np.int64))
    v = torch.as_tensor(val, dtype=x.dtype)
    out = torch.zeros((x.shape[0], n_rows, x.shape[-1]), dtype=x.dtype)
    return out.index_add_(1, rows, v[None, :, None] * x[:, cols])


def node_edge_sparse(x, inci, w, b):
    """NodeEdge restated as support CSR + dense remainder: exact for ANY trained
    `b` (the off-support part of b is a genuinely dense (R,M) product)."""
    inci_np = np.asarray(inci)
    rowptr, col, ival = csr_from_dense(inci_np)
    rows = _rows_of(rowptr)
    wv = w[rows, col.astype(np.int64)] * torch.as_tensor(ival, dtype=x.dtype) + b[rows, col.astype(np.int64)]
    out = incidence_aggregate_csr(x, rowptr, col, wv, inci_np.shape[0])
    b_off = b.clone()
    b_off[rows, col.astype(np.int64)] = 0.0
    if bool((b_off != 0).any()):
        out = out + b_off @ x
    return out


def node_edge_support(inci, w, b):
    """Dense NodeEdge parameters -> (rowptr, col, support values w*inci+b, off-support rest of b or None)."""
    inci_np = np.asarray(inci)
    rowptr, col, ival = csr_from_dense(inci_np)
    rows = _rows_of(rowptr)
    c64 = col.astype(np.int64)
    v = w[rows, c64] * torch.as_tensor(ival, dtype=w.dtype) + b[rows, c64]
    rest = b.clone()
    rest[rows, c64] = 0.0
    return rowptr, col, v, (rest if bool((rest != 0).any()) else None)


def spatial_layer_csr(x, e, p, adj_csr, eadj_csr, inc_n=None, inc_e=None, node_edge=None, act='relu'):
    """Sparse restatement of `emulator.py:225-230`; p as in spatial_layer_dense.
    adj_csr / eadj_csr = (rowptr, col) WITH self loops (csr_from_dense(f, True)).
    NodeEdge parameters either dense ('ne_n_w','ne_n_b','ne_e_w','ne_e_b' + node_edge (N,E)) or
    already on the support ('ne_n_v','ne_e_v' + inc_n / inc_e = (rowptr, col) of |node_edge| and its
    transpose) -- the latter is the only form that exists for graphs too big for (N,E) matrices."""
    x_e = dense(e, p['xe_k'], p['xe_b'], act)
    e_x = dense(x, p['ex_k'], p['ex_b'], act)
    if 'ne_n_v' in p:
        agg_n = incidence_aggregate_csr(x_e, inc_n[0], inc_n[1], p['ne_n_v'], x.shape[1])
        agg_e = incidence_aggregate_csr(e_x, inc_e[0], inc_e[1], p['ne_e_v'], e.shape[1])
    else:
        inci = node_edge.abs()
        agg_n = node_edge_sparse(x_e, inci, p['ne_n_w'], p['ne_n_b'])
        agg_e = node_edge_sparse(e_x, inci.T, p['ne_e_w'], p['ne_e_b'])
    xc = torch.cat([x, agg_n], dim=-1)
    ec = torch.cat([e, agg_e], dim=-1)
    x_new = gat_conv_csr(xc, adj_csr[0], adj_csr[1], p['gx_k'], p['gx_as'], p['gx_an'], p['gx_b'], act)
    e_new = gat_conv_csr(ec, eadj_csr[0], eadj_csr[1], p['ge_k'], p['ge_as'], p['ge_an'], p['ge_b'], act)
    return x_new, e_new
