"""Dense-masked CPU oracle: op-for-op what the reference computes (TEST INFRASTRUCTURE ONLY).

PARITY UNPINNED (see oracle/__init__.py): Spektral 1.3.1 is restated from its
published algorithm, the reference holds no vectors for it.

Every function works on torch CPU tensors of any float dtype (fp64 for the
oracle-vs-oracle checks, fp32 for the timed CPU baseline) and mirrors the
reference's DENSE formulation: `(N,N)` 0/1 filters, masked softmax, dense
`N x E` incidence products.  Layouts are the reference's: `x:(S,N,F)`,
`e:(S,E,F)` with `S = B*T` snapshots (`emulator.py:217-218`).
"""
import torch

LEAKY_SLOPE = 0.2        # spektral GATConv: tf.nn.leaky_relu(attn_coef, alpha=0.2)
MASK_VALUE = -10e9       # spektral GATConv: tf.where(a == 0.0, -10e9, 0.0)


def activation(name):
    """keras.activations.get(name) for the names the reference uses
    (`emulator.py:66,193,324,330,336`).  hard_sigmoid is Keras-2.10's
    clip(0.2*x+0.5, 0, 1)."""
    if name is None or name == 'linear':
        return lambda t: t
    if name == 'relu':
        return torch.relu
    if name == 'tanh':
        return torch.tanh
    if name == 'sigmoid':
        return torch.sigmoid
    if name == 'hard_sigmoid':
        return lambda t: torch.clamp(0.2 * t + 0.5, 0.0, 1.0)
    raise ValueError('unknown activation %r' % (name,))


def dense(x, kernel, bias, act='linear'):
    """keras Dense: act(x @ kernel + bias) on the last axis (`emulator.py:198,203,225-226`)."""
    y = x @ kernel
    if bias is not None:
        y = y + bias
    return activation(act)(y)


# None = inference.  A test sets it to a callable (coef (...,N,H,N), a_hat (N,N) with the diagonal set) -> coef that applies the
# mask stream it compares with (tests/test_gpu_dropout.py; Spektral draws its mask from TensorFlow's generator: statistical parity only).
ATTN_DROPOUT = None


def gat_conv_dense(x, a, kernel, attn_kernel_self, attn_kernel_neighs, bias,
                   act='relu', add_self_loops=True, concat_heads=True, return_attn=False):
    """spektral.layers.GATConv._call_dense + call() tail, as used through `MixedGAT`
    (`emulator.py:18-25,229-230,282-283`).  Mixed mode: `x:(...,N,F)`, `a:(N,N)`.

    kernel (F,H,C); attn_kernel_self/neighs (C,H,1); bias (H*C) if concat else (C).
    Dropout on the coefficients (`attn_coef_drop = self.dropout(attn_coef)`, rate dropout_rate = 0.5) is inactive unless the
    emulator passes training=True, which it does only when self.dropout > 0 (`emulator.py:411,434`): ATTN_DROPOUT below.
    """
    a = a.to(x.dtype).clone()
    n = a.shape[-1]
    if add_self_loops:                                  # tf.linalg.set_diag(a, ones)
        idx = torch.arange(n)
        a[..., idx, idx] = 1.0
    hx = torch.einsum('...ni,iho->...nho', x, kernel)                     # (...,N,H,C)
    attn_self = torch.einsum('...nhi,iho->...nho', hx, attn_kernel_self)  # (...,N,H,1)
    attn_neigh = torch.einsum('...nhi,iho->...nho', hx, attn_kernel_neighs)
    attn_neigh = attn_neigh.transpose(-1, -3)                             # "...ABC->...CBA": (...,1,H,N)
    coef = attn_self + attn_neigh                                          # (...,N,H,N)
    coef = torch.nn.functional.leaky_relu(coef, LEAKY_SLOPE)
    mask = torch.where(a == 0.0, torch.tensor(MASK_VALUE, dtype=x.dtype), torch.tensor(0.0, dtype=x.dtype))
    coef = coef + mask[..., None, :]                                       # (N,1,N) broadcast
    coef = torch.softmax(coef, dim=-1)
    if ATTN_DROPOUT is not None:                                           # training=True: dropout on the normalised coefficients
        coef = ATTN_DROPOUT(coef, a)
    out = torch.einsum('...nhm,...mhi->...nhi', coef, hx)                  # (...,N,H,C)
    if concat_heads:
        out = out.reshape(out.shape[:-2] + (out.shape[-2] * out.shape[-1],))
    else:
        out = out.mean(dim=-2)
    if bias is not None:
        out = out + bias
    out = activation(act)(out)
    return (out, coef) if return_attn else out


def gcn_preprocess(adj):
    """spektral GCNConv.preprocess = gcn_filter: D^-1/2 (A+I) D^-1/2 with row-sum
    degrees and inf -> 0 (`emulator.py:133-134`)."""
    a = adj.to(torch.float64) + torch.eye(adj.shape[-1], dtype=torch.float64)
    deg = a.sum(dim=1)
    dinv = deg.pow(-0.5)
    dinv[torch.isinf(dinv)] = 0.0
    return dinv[:, None] * a * dinv[None, :]


def gcn_conv_dense(x, a_hat, kernel, bias, act='relu'):
    """spektral GCNConv.call: act(a_hat @ (x @ kernel) + bias) (`emulator.py:131-134,229`)."""
    out = a_hat.to(x.dtype) @ (x @ kernel)
    if bias is not None:
        out = out + bias
    return activation(act)(out)


def diffusion_preprocess(adj):
    """spektral DiffusionConv.preprocess = normalized_adjacency(A): D^-1/2 A D^-1/2, NO self loops added, row-sum degrees,
    inf -> 0 (`emulator.py:137-138`; restated from Spektral 1.3.1 utils/convolution.py -- unverifiable offline)."""
    a = adj.to(torch.float64)
    deg = a.sum(dim=1)
    dinv = deg.pow(-0.5)
    dinv[torch.isinf(dinv)] = 0.0
    return dinv[:, None] * a * dinv[None, :]


def diffusion_conv_dense(x, a_hat, kernel, act='tanh'):
    """spektral DiffusionConv.call (`emulator.py:135-138,229`), restated from Spektral 1.3.1 diffusion_conv.py: `channels`
    DiffuseFeatures filters, each with K + 1 coefficients theta (kernel[q], highest power first), each producing ONE column:
        D_q = tf.math.polyval(theta_q, a_hat)        -- Horner's rule on the ENTRIES of a_hat (element-wise powers, not
                                                        matrix powers): D_q[i, j] = sum_k theta_q[k] * a_hat[i, j]^(K - k);
                                                        an entry that is zero in a_hat gets the constant term theta_q[K]
        H_q = reduce_sum(D_q @ x, axis=-1)           -- summed over the input features
    out = act(concat_q H_q), no bias.  x (S, N, F), a_hat (N, N), kernel (channels, K + 1) -> (S, N, channels)."""
    a = a_hat.to(x.dtype)
    cols = []
    for q in range(kernel.shape[0]):
        d = torch.zeros_like(a) + kernel[q, 0]
        for k in range(1, kernel.shape[1]):
            d = d * a + kernel[q, k]
        cols.append((d @ x).sum(dim=-1, keepdim=True))
    return activation(act)(torch.cat(cols, dim=-1))


def node_edge_dense(x, inci, w, b):
    """NodeEdge.call (`emulator.py:42-45`): (w * inci + b) @ x, `inci:(R,M)`, `x:(...,M,F)`."""
    mat = w * inci.to(x.dtype) + b
    return mat @ x


def spatial_layer_dense(x, e, p, adj_filter, edge_filter, node_edge, act='relu', conv='GAT'):
    """One iteration of the spatial-block loop body (`emulator.py:225-230` / `:278-283`).

    p: dict with
      'xe_k','xe_b' : Dense(d/2) applied to e  -> x_e          (:225)
      'ex_k','ex_b' : Dense(d/2) applied to x  -> e_x          (:226)
      'ne_n_w','ne_n_b' : NodeEdge(|node_edge|) params (N,E)   (:227)
      'ne_e_w','ne_e_b' : NodeEdge(|node_edge|^T) params (E,N) (:228)
      'gx_k','gx_as','gx_an','gx_b' : GAT on nodes             (:229)
      'ge_k','ge_as','ge_an','ge_b' : GAT on the line graph    (:230)
    """
    inci = node_edge.abs()
    x_e = dense(e, p['xe_k'], p['xe_b'], act)
    e_x = dense(x, p['ex_k'], p['ex_b'], act)
    xc = torch.cat([x, node_edge_dense(x_e, inci, p['ne_n_w'], p['ne_n_b'])], dim=-1)
    ec = torch.cat([e, node_edge_dense(e_x, inci.T, p['ne_e_w'], p['ne_e_b'])], dim=-1)
    if conv == 'GAT':
        x_new = gat_conv_dense(xc, adj_filter, p['gx_k'], p['gx_as'], p['gx_an'], p['gx_b'], act)
        e_new = gat_conv_dense(ec, edge_filter, p['ge_k'], p['ge_as'], p['ge_an'], p['ge_b'], act)
    elif conv == 'GCN':
        x_new = gcn_conv_dense(xc, adj_filter, p['gx_k'][:, 0, :], p['gx_b'], act)
        e_new = gcn_conv_dense(ec, edge_filter, p['ge_k'][:, 0, :], p['ge_b'], act)
    elif conv == 'Diffusion':
        x_new = diffusion_conv_dense(xc, adj_filter, p['gx_theta'], act)
        e_new = diffusion_conv_dense(ec, edge_filter, p['ge_theta'], act)
    else:
        raise ValueError(conv)
    return x_new, e_new


def spatial_block_dense(x, e, layers, adj_filter, edge_filter, node_edge, act='relu', conv='GAT'):
    """`for _ in range(n_sp_layer)` (`emulator.py:219-235`)."""
    for p in layers:
        x, e = spatial_layer_dense(x, e, p, adj_filter, edge_filter, node_edge, act, conv)
    return x, e
