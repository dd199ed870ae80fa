"""torch.nn.Module mirrors of the reference's Keras / Spektral layers, running on the HIP engine.

The reference is TensorFlow/Keras + Spektral; "drop-in" therefore means: same layer
constructor arguments, same call convention `layer([x, a]) -> x'`, same weight names and shapes,
same tensor layouts -- re-expressed as `torch.nn.Module`s:

    Dense(units, activation)                      keras.layers.Dense        emulator.py:198,203,225-226,...
    NodeEdge(inci)                                emulator.py:27-45
    GATConv / MixedGAT(channels, activation=..)   spektral GATConv via emulator.py:18-25,229-230
    GCNConv(channels, activation=..) + preprocess spektral GCNConv via emulator.py:131-134
    SpatialLayer / SpatialBlock                   the loop bodies emulator.py:219-235, 272-288

Parameters are created with requires_grad=False (inference: no autograd graph is recorded).  After
`module.requires_grad_(True)` the same modules run through the autograd Functions of `autograd.py`
(HIP backward kernels; the training step of SURVEY.md a10).  All compute is HIP kernels behind the
C ABI (gnn_uds_amd/_lib.py); CPU tensors raise.
"""
import math
import os

import numpy as np
import torch
from torch import nn

from . import _lib
from . import autograd as _ag
from .graph import CSR, DrainageGraph, csr_from_dense


def _param(t):
    return nn.Parameter(t, requires_grad=False)


def _glorot_uniform(shape, device, generator=None):
    """Keras glorot_uniform incl. its fan rule for >2-D kernels (receptive field = prod(shape[:-2]))."""
    if len(shape) == 1:
        fan_in = fan_out = shape[0]
    elif len(shape) == 2:
        fan_in, fan_out = shape
    else:
        rf = int(np.prod(shape[:-2]))
        fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
    limit = math.sqrt(6.0 / (fan_in + fan_out))
    t = torch.rand(shape, generator=generator, dtype=torch.float32) * (2 * limit) - limit
    return t.to(device)


def _flatten_snapshots(x):
    """(..., M, F) -> (S, M, F) contiguous, plus the leading shape to restore."""
    if x.dim() < 2:
        raise _lib.UdsError('expected (..., elements, features), got %r' % (tuple(x.shape),))
    lead = tuple(x.shape[:-2])
    return x.reshape((-1,) + tuple(x.shape[-2:])).contiguous(), lead


def _packed_kernel(module, kernel2d):
    """bf16 hi/lo MFMA fragments of a (K, f_out) kernel, re-split only when the parameter changes."""
    key = (module.kernel._version, module.kernel.data_ptr())
    hit = getattr(module, '_packed', None)
    if hit is None or hit[0] != key:
        module._packed = (key, _lib.rowgemm_pack(kernel2d.contiguous()))
    return module._packed[1]


class Dense(nn.Module):
    """keras.layers.Dense(units, activation): act(x @ kernel + bias) on the last axis.

    precision='fp32': exact-fp32 FMA kernel (any shape).  precision='bf16x3': matrix-core kernel (operands split into
    bf16 hi + lo, 3 MFMA products, fp32 accumulate) when the input width is a multiple of 32 and units <= 64, the
    exact kernel otherwise."""

    def __init__(self, units, activation=None, use_bias=True, in_features=None, generator=None, precision='fp32'):
        super().__init__()
        self.units, self.activation, self.use_bias = int(units), activation or 'linear', use_bias
        self.precision = precision
        self._gen = generator
        self.kernel = self.bias = None
        if self.activation not in _lib.ACT:
            raise ValueError('unknown activation %r' % (activation,))
        if in_features is not None:
            self.build(in_features, 'cpu')

    def build(self, in_features, device):
        self.kernel = _param(_glorot_uniform((int(in_features), self.units), device, self._gen))
        self.bias = _param(torch.zeros(self.units, device=device)) if self.use_bias else None

    def forward(self, x, activation=None):
        """activation: override of the layer's own (the reference applies the embedding's activation outside the layer,
        emulator.py:198-201)."""
        if self.kernel is None:
            self.build(x.shape[-1], x.device)
        act = activation or self.activation
        if _ag.grad_on(x, self.kernel, self.bias):
            return _ag.DenseFn.apply(x, self.kernel, self.bias, self, act)
        xc = x.contiguous()
        fi = xc.shape[-1]
        if self.precision == 'bf16x3' and _lib.rowgemm_supported(fi, fi, self.units):
            return _lib.rowgemm_forward(xc, _packed_kernel(self, self.kernel), self.bias, self.units, act)
        if self.precision == 'bf16x3' and self.units > 64 and self.units % 16 == 0 and _lib.rowgemm_supported(fi, fi, 64) and \
                xc.numel() // fi >= 4096:
            # wide outputs (d = 128): 64-column blocks of the kernel, each written into its columns of one output
            key = (self.kernel._version, self.kernel.data_ptr(), None if self.bias is None else self.bias._version)
            if getattr(self, '_blocks', None) is None or self._blocks[0] != key:
                self._blocks = (key, [(c0, min(64, self.units - c0), _lib.rowgemm_pack(self.kernel[:, c0:c0 + 64].contiguous()),
                                       None if self.bias is None else self.bias[c0:c0 + 64].contiguous())
                                      for c0 in range(0, self.units, 64)])
            out = torch.empty(xc.shape[:-1] + (self.units,), device=xc.device, dtype=torch.float32)
            for c0, w, pk, bb in self._blocks[1]:
                _lib.rowgemm_cat(xc, None, pk, bb, w, act, out=out, col0=c0)
            return out
        return _lib.dense_act(xc, self.kernel, self.bias, act)


class _GraphArg:
    """Turns the `a` of `layer([x, a])` into a device CSR handle, caching dense inputs by identity."""

    def __init__(self):
        self._cache = {}

    def handle(self, a, add_self_loops):
        if isinstance(a, _lib.CsrHandle):
            return a
        if isinstance(a, CSR):
            key = ('csr', id(a))
            if key not in self._cache:
                self._cache[key] = (a, _lib.CsrHandle(a))
            return self._cache[key][1]
        key = ('dense', id(a), add_self_loops)
        hit = self._cache.get(key)
        if hit is not None and hit[0] is a:
            return hit[1]
        dense = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
        h = _lib.CsrHandle(csr_from_dense(dense, add_self_loops=add_self_loops))
        self._cache[key] = (a, h)
        return h


class GATConv(nn.Module):
    """spektral.layers.GATConv in batch/mixed mode, single head (what the reference uses).

    forward([x, a]): x (..., N, F); a = dense (N, N) 0/1 filter as in the reference
    (`emulator.py:143-145,229`), a `graph.CSR`, or a prebuilt `_lib.CsrHandle` (CSR inputs must
    already hold the self loops).  Weights keep Spektral's names and shapes:
    kernel (F,1,C), attn_kernel_self (C,1,1), attn_kernel_neighs (C,1,1), bias (C)."""

    def __init__(self, channels, attn_heads=1, concat_heads=True, dropout_rate=0.5, return_attn_coef=False,
                 add_self_loops=True, activation=None, use_bias=True, in_channels=None, generator=None):
        super().__init__()
        if attn_heads != 1:
            raise NotImplementedError('the HIP engine builds the single-head GAT the reference uses (attn_heads=1)')
        if return_attn_coef:
            raise NotImplementedError('return_attn_coef is not built')
        self.channels, self.attn_heads, self.concat_heads = int(channels), 1, concat_heads
        self.dropout_rate, self.add_self_loops = dropout_rate, add_self_loops
        self.activation, self.use_bias = activation or 'linear', use_bias
        self._gen = generator
        self._graphs = _GraphArg()
        self.precision = 'fp32'      # numerics of the backward row GEMMs (the forward linear part is the exact kernel)
        self.kernel = self.attn_kernel_self = self.attn_kernel_neighs = self.bias = None
        if in_channels is not None:
            self.build(in_channels, 'cpu')

    def build(self, in_channels, device):
        c = self.channels
        self.kernel = _param(_glorot_uniform((int(in_channels), 1, c), device, self._gen))
        self.attn_kernel_self = _param(_glorot_uniform((c, 1, 1), device, self._gen))
        self.attn_kernel_neighs = _param(_glorot_uniform((c, 1, 1), device, self._gen))
        self.bias = _param(torch.zeros(c, device=device)) if self.use_bias else None

    def forward(self, inputs, xb=None, edge_mask=None, attn_dropout=None):
        """edge_mask (..., nnz): per-snapshot 0/1 over the entries of the pattern `a` (`use_adj`, emulator.py:268-271: the
        reference feeds a (S, N, N) adjacency here; the mask is that adjacency gathered at the static pattern's entries --
        `Emulator.get_adj_action`).  The diagonal always takes part (set_diag).  Inference only.
        attn_dropout: a DropoutStream = the layer runs in Keras' training mode and `dropout_rate` > 0: Spektral's dropout on the
        normalised attention coefficients (`attn_coef_drop = self.dropout(attn_coef)`), one mask entry per (snapshot, pattern entry)."""
        x, a = inputs
        if self.kernel is None:
            self.build(x.shape[-1] + (0 if xb is None else xb.shape[-1]), x.device)
        h = self._graphs.handle(a, self.add_self_loops)
        xs, lead = _flatten_snapshots(x)
        xbs = None if xb is None else _flatten_snapshots(xb)[0]
        if edge_mask is not None:
            if _ag.grad_on(xs, xbs, self.kernel, self.attn_kernel_self, self.attn_kernel_neighs, self.bias):
                raise NotImplementedError('GATConv with a per-snapshot edge mask (use_adj) is built for inference')
            fin = xs.shape[-1] + (0 if xbs is None else xbs.shape[-1])
            hx, s_self, s_nbr = _lib.dense_act(xs, self.kernel.reshape(fin, self.channels), None, 'linear', xbs,
                                               attn=(self.attn_kernel_self.reshape(-1), self.attn_kernel_neighs.reshape(-1)))
            mk = edge_mask.reshape(-1, edge_mask.shape[-1]).to(torch.float32).contiguous()
            out = _lib.gat_aggregate(h, hx, s_self, s_nbr, self.bias, self.activation, edge_mask=mk)
            return out.reshape(lead + out.shape[-2:])
        coef = None
        if attn_dropout is not None and self.dropout_rate:
            if edge_mask is not None:
                raise NotImplementedError('attention dropout together with a per-snapshot edge mask (use_adj) is not built')
            ones = torch.ones((xs.shape[0], h.nnz), device=xs.device, dtype=torch.float32)
            coef = _lib.dropout(ones, self.dropout_rate, attn_dropout.seed, attn_dropout.take(ones.numel()))
        if coef is not None or _ag.grad_on(xs, xbs, self.kernel, self.attn_kernel_self, self.attn_kernel_neighs, self.bias):
            out = _ag.GatFn.apply(xs, xbs, self.kernel, self.attn_kernel_self, self.attn_kernel_neighs, self.bias,
                                  self.activation, h, self.precision, coef)
            return out.reshape(lead + out.shape[-2:])
        fin = xs.shape[-1] + (0 if xbs is None else xbs.shape[-1])
        if self.precision == 'bf16x3' and fin % 32 == 0 and self.channels % 16 == 0 and xs.shape[0] * xs.shape[1] >= 4096:
            # wide layers (d = 128, the reference's default): the linear part on the matrix cores (split-bf16 row GEMM, 64
            # columns per launch), the attention projections as two matrix-vector products, then the CSR aggregation kernel
            w2 = self.kernel.reshape(fin, self.channels)
            key = (self.kernel._version, self.attn_kernel_self._version, self.attn_kernel_neighs._version, self.kernel.data_ptr())
            if getattr(self, '_wide', None) is None or self._wide[0] != key:
                # per parameter version: the kernel's 64-column blocks and the two attention projections W a_self, W a_nbr
                # (s = hx a = z (W a): one more narrow row GEMM instead of a 600k-row matrix-vector product)
                cols = [_lib.rowgemm_pack(w2[:, c0:c0 + 64].contiguous()) for c0 in range(0, self.channels, 64)]
                wa = torch.stack([w2 @ self.attn_kernel_self.reshape(-1), w2 @ self.attn_kernel_neighs.reshape(-1)], dim=1)
                self._wide = (key, cols, _lib.rowgemm_pack(wa.contiguous()))
            _, cols, wa_packed = self._wide
            if not all(_lib.rowgemm_supported(fin, 32, min(64, self.channels - 64 * i)) for i in range(len(cols))):
                raise _lib.UdsError('GATConv: %d -> %d does not fit the matrix-core row GEMM' % (fin, self.channels))
            hx = torch.empty(xs.shape[:-1] + (self.channels,), device=xs.device, dtype=torch.float32)
            for i, pk in enumerate(cols):
                _lib.rowgemm_cat(xs, xbs, pk, None, min(64, self.channels - 64 * i), 'linear', out=hx, col0=64 * i)
            s2 = _lib.rowgemm_cat(xs, xbs, wa_packed, None, 2, 'linear')
            s_self, s_nbr = s2[..., 0].contiguous(), s2[..., 1].contiguous()
            out = _lib.gat_aggregate(h, hx, s_self, s_nbr, self.bias, self.activation)
            return out.reshape(lead + out.shape[-2:])
        out = _lib.gat_forward(h, xs, self.kernel, self.attn_kernel_self, self.attn_kernel_neighs, self.bias,
                               self.activation, xbs)
        return out.reshape(lead + out.shape[-2:])


MixedGAT = GATConv   # emulator.py:18-25: only re-types the (inactive) attention dropout


class GCNConv(nn.Module):
    """spektral.layers.GCNConv: act(a_hat @ (x @ kernel) + bias); a_hat = GCNConv.preprocess(adj)."""

    def __init__(self, channels, activation=None, use_bias=True, in_channels=None, generator=None):
        super().__init__()
        self.channels, self.activation, self.use_bias = int(channels), activation or 'linear', use_bias
        self.units, self.precision = self.channels, 'fp32'      # what autograd.DenseFn reads from its module (exact-fp32 kernels)
        self._gen = generator
        self._cache = {}
        self.kernel = self.bias = None
        if in_channels is not None:
            self.build(in_channels, 'cpu')

    @staticmethod
    def preprocess(adj):
        """gcn_filter: D^-1/2 (A + I) D^-1/2, row-sum degrees, inf -> 0 (`emulator.py:133`)."""
        a = np.asarray(adj, dtype=np.float64) + np.eye(len(adj))
        deg = a.sum(axis=1)
        with np.errstate(divide='ignore'):
            dinv = np.power(deg, -0.5)
        dinv[np.isinf(dinv)] = 0.0
        return dinv[:, None] * a * dinv[None, :]

    def build(self, in_channels, device):
        self.kernel = _param(_glorot_uniform((int(in_channels), self.channels), device, self._gen))
        self.bias = _param(torch.zeros(self.channels, device=device)) if self.use_bias else None

    def _filter(self, a, device):
        hit = self._cache.get(id(a))
        if hit is not None and hit[0] is a:
            return hit[1], hit[2]
        dense = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
        csr = csr_from_dense(dense, keep_values=True)
        h = _lib.CsrHandle(csr)
        val = torch.as_tensor(csr.val, dtype=torch.float32, device=device)
        self._cache[id(a)] = (a, h, val)
        return h, val

    def forward(self, inputs):
        x, a = inputs
        if self.kernel is None:
            self.build(x.shape[-1], x.device)
        h, val = self._filter(a, x.device)
        xs, lead = _flatten_snapshots(x)
        if _ag.grad_on(x, self.kernel, self.bias):
            # training: x @ kernel and the propagation a_hat @ (.) through their autograd Functions (the filter values are
            # constants: no gradient), bias + activation as tensor ops
            hx = _ag.DenseFn.apply(xs, self.kernel, None, self, 'linear')
            out = _ag.SpmmFn.apply(val, hx, h)
            if self.bias is not None:
                out = out + self.bias
            out = _ag.apply_activation(out, self.activation)
            return out.reshape(lead + out.shape[-2:])
        hx = _lib.dense_act(xs, self.kernel, None, 'linear')
        out = _lib.csr_spmm(h, val, hx, self.bias, self.activation)
        return out.reshape(lead + out.shape[-2:])


class DiffusionConv(nn.Module):
    """spektral.layers.DiffusionConv(channels, K=6, activation='tanh') as the reference runs it (`emulator.py:135-138,229`:
    dense mixed mode, `net(embed_size, activation=...)([x, filter])`): `channels` DiffuseFeatures filters with K + 1
    coefficients each -- `kernel` (channels, K + 1), row q = theta_q, highest power first, glorot_uniform -- and
        out[..., q] = act( reduce_sum( polyval(theta_q, a_hat) @ x, -1 ) ),     no bias,
    where tf.math.polyval is Horner's rule on the ENTRIES of a_hat (element-wise powers).  A zero entry of a_hat therefore
    takes the constant coefficient theta_q[K]; with r = x.sum(-1) the dense (N, N) product collapses to the support of a_hat
    plus a rank-one term (uds_diffusion_forward).  a_hat = DiffusionConv.preprocess(adj).  Inference only."""

    def __init__(self, channels, K=6, activation='tanh', in_channels=None, generator=None):
        super().__init__()
        self.channels, self.K, self.activation = int(channels), int(K) + 1, activation or 'linear'      # spektral: self.K = K + 1
        if self.channels % 4:
            raise ValueError('DiffusionConv: channels must be a multiple of 4, got %d' % self.channels)
        lim = math.sqrt(6.0 / (2 * self.K))         # glorot_uniform on shape (K + 1,): fan_in = fan_out = K + 1
        self.kernel = _param((torch.rand((self.channels, self.K), generator=generator) * 2 - 1) * lim)
        self._cache, self._vals = {}, None

    @staticmethod
    def preprocess(adj):
        """normalized_adjacency: D^-1/2 A D^-1/2 (no self loops added), row-sum degrees, inf -> 0 (`emulator.py:137-138`)."""
        a = np.asarray(adj, dtype=np.float64)
        deg = a.sum(axis=1)
        with np.errstate(divide='ignore'):
            dinv = np.power(deg, -0.5)
        dinv[np.isinf(dinv)] = 0.0
        return dinv[:, None] * a * dinv[None, :]

    def _filter(self, a, device):
        hit = self._cache.get(id(a))
        if hit is None or hit[0] is not a:
            dense = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
            csr = csr_from_dense(dense, keep_values=True)
            hit = self._cache[id(a)] = (a, _lib.CsrHandle(csr), torch.as_tensor(csr.val, dtype=torch.float32, device=device))
            self._vals = None
        key = (self.kernel._version, self.kernel.data_ptr(), id(a))
        if self._vals is None or self._vals[0] != key:
            av = hit[2].double()[:, None]
            th = self.kernel.detach().double()
            v = th[:, 0].expand(av.shape[0], -1)
            for k in range(1, self.K):                       # Horner, as tf.math.polyval
                v = v * av + th[:, k]
            self._vals = (key, (v - th[:, -1]).float().contiguous(), th[:, -1].float().contiguous())
        return hit[1], self._vals[1], self._vals[2]

    def forward(self, inputs):
        x, a = inputs
        if _ag.grad_on(x, self.kernel):
            raise NotImplementedError('DiffusionConv is built for inference (no backward kernels)')
        h, vals, c0 = self._filter(a, x.device)
        xs, lead = _flatten_snapshots(x)
        r = xs.sum(dim=-1)
        out = _lib.diffusion_forward(h, vals, c0, r.contiguous(), r.sum(dim=-1).contiguous(), self.activation)
        return out.reshape(lead + out.shape[-2:])


class NodeEdge(nn.Module):
    """`NodeEdge(inci)` (`emulator.py:27-45`): out = (weight * inci + bias) @ x, inci (R, M).

    `weight`, `bias` keep the reference's dense (R, M) shape by default.  The product is evaluated on
    the incidence support with the CSR kernel; a trained `bias` that is non-zero OFF the support makes
    the matrix genuinely dense, and that remainder is added by the hand-written split-bf16 MFMA GEMM
    (`remainder`: uds_remainder_forward; the check is re-done whenever the parameters change).  For networks where an (R, M) parameter cannot exist
    (N = 50k: 13 GB each) pass sparse=True: parameters then have one entry per support element
    (columns ascending inside a row)."""

    def __init__(self, inci, sparse=False, device='cpu', generator=None):
        super().__init__()
        if isinstance(inci, CSR):
            csr = inci if inci.val is not None else CSR(inci.rowptr, inci.col, inci.n_rows, inci.n_cols,
                                                        np.ones(inci.nnz))
        else:
            dense = inci.detach().cpu().numpy() if isinstance(inci, torch.Tensor) else np.asarray(inci)
            csr = csr_from_dense(dense, keep_values=True)
        self.csr = csr
        self.sparse = bool(sparse)
        self.shape = (csr.n_rows, csr.n_cols)
        rows = torch.as_tensor(csr.rows(), dtype=torch.int64)
        cols = torch.as_tensor(csr.col.astype(np.int64))
        self.register_buffer('_flat', rows * csr.n_cols + cols, persistent=False)
        self.register_buffer('_ival', torch.as_tensor(csr.val, dtype=torch.float32), persistent=False)
        shape = (csr.nnz,) if self.sparse else self.shape
        w = torch.randn(shape, generator=generator, dtype=torch.float32) * 0.05      # 'random_normal'
        self.weight = _param(w.to(device))
        self.bias = _param(torch.zeros(shape, device=device))
        self._handle = None
        self._val_cache = None
        self._rest_packed = None
        self.precision = 'bf16x3'      # of the dense remainder GEMM; SpatialLayer sets it to its own

    def support_values(self):
        """weight*inci + bias on the support (nnz,), plus the off-support remainder of bias or None."""
        key = (self.weight._version, self.bias._version, self.weight.data_ptr(), self.bias.data_ptr())
        if self._val_cache is not None and self._val_cache[0] == key:
            return self._val_cache[1], self._val_cache[2]
        flat, ival = self._flat.to(self.weight.device), self._ival.to(self.weight.device)
        if self.sparse:
            val, rest = self.weight * ival + self.bias, None
        else:
            val = self.weight.reshape(-1)[flat] * ival + self.bias.reshape(-1)[flat]
            rest = self.bias.clone()
            rest.reshape(-1)[flat] = 0.0
            if not bool((rest != 0).any()):
                rest = None
        self._val_cache = (key, val.contiguous(), rest)
        self._rest_packed = None
        return self._val_cache[1], rest

    def remainder(self, xs, precision='bf16x3'):
        """`rest @ xs`, rest = the trained bias off the incidence support (emulator.py:36-39,44): the dense GEMM on the
        matrix cores (uds_remainder_forward; `rest` is split into bf16 hi/lo once per parameter update), exact fp32 through
        rocBLAS when precision='fp32'.  None when the bias is zero off the support."""
        rest = self.support_values()[1]
        if rest is None:
            return None
        if precision != 'bf16x3' or xs.shape[-1] % 4 or xs.shape[-1] > 64:
            return torch.matmul(rest, xs)
        if self._rest_packed is None:
            self._rest_packed = _lib.remainder_pack(rest)
        return _lib.remainder_forward(self._rest_packed, tuple(rest.shape), xs)

    def remainder_of_dense(self, es, dense):
        """`rest @ dense(es)` with the Dense layer's output written straight into the GEMM's operand planes (uds_remainder_forward_dense);
        None when the shapes are not the ones that entry takes -- the caller then materialises dense(es) and calls remainder()."""
        rest = self.support_values()[1]
        if rest is None or dense.precision != 'bf16x3' or es.shape[-1] not in (64, 128) or dense.units not in (32, 64) or not es.is_cuda \
                or rest.shape[1] < 64:      # (a 30-row network fills half of that kernel's 64-row blocks: the flattened row GEMM is the better Dense there)
            return None
        if self._rest_packed is None:
            self._rest_packed = _lib.remainder_pack(rest)
        return _lib.remainder_forward_dense(self._rest_packed, tuple(rest.shape), es.contiguous(), _packed_kernel(dense, dense.kernel), dense.bias,
                                            dense.activation, dense.units)

    def handle(self):
        if self._handle is None:
            self._handle = _lib.CsrHandle(self.csr)
        return self._handle

    def forward(self, x):
        if x.shape[-2] != self.shape[1]:
            raise _lib.UdsError('NodeEdge expects %d elements on axis -2, got %r' % (self.shape[1], tuple(x.shape)))
        xs, lead = _flatten_snapshots(x)
        if _ag.grad_on(xs, self.weight, self.bias):
            # training: the support values are re-derived under autograd (gradients reach `weight` and `bias`); with the
            # reference's dense (R, M) parameters the bias also trains OFF the support (emulator.py:36-39,44), which is
            # a dense GEMM and only feasible for small networks -- sparse=True keeps one trainable entry per support element
            flat, ival = self._flat.to(self.weight.device), self._ival.to(self.weight.device)
            if self.sparse:
                val = self.weight * ival + self.bias
                out = _ag.SpmmFn.apply(val, xs, self.handle())
            else:
                val = self.weight.reshape(-1)[flat] * ival + self.bias.reshape(-1)[flat]
                out = _ag.SpmmFn.apply(val, xs, self.handle())
                if self.bias.requires_grad:
                    off = torch.ones_like(self.bias)
                    off.reshape(-1)[flat] = 0.0
                    rest = self.bias * off
                    if self.precision == 'bf16x3' and xs.shape[-1] % 4 == 0 and xs.shape[-1] <= 64:
                        out = out + _ag.RemainderFn.apply(rest, xs)      # the N x E x (S h) products on the HIP MFMA GEMM
                    else:
                        out = out + torch.matmul(rest, xs)
            return out.reshape(lead + out.shape[-2:])
        val, rest = self.support_values()
        out = _lib.csr_spmm(self.handle(), val, xs)
        if rest is not None:
            out = out + self.remainder(xs, self.precision)
        return out.reshape(lead + out.shape[-2:])


class SpatialLayer(nn.Module):
    """One iteration of the spatial-block loop (`emulator.py:225-230`, `:278-283`), conv = GAT:

        x_e = Dense(d/2, act)(e);  e_x = Dense(d/2, act)(x)
        x = concat[x, NodeEdge(|node_edge|)(x_e)];  e = concat[e, NodeEdge(|node_edge|^T)(e_x)]
        x = GAT(d, act)([x, Adj]);  e = GAT(d, act)([e, Eadj])

    forward(x, e): x (..., N, Fx), e (..., E, Fe) -> (..., N, d), (..., E, d).  Uses the fused C-ABI
    entry uds_spatial_layer_forward (concats are never materialised)."""

    def __init__(self, graph, embed_size, activation='relu', fx=None, fe=None, sparse_params=None, net=None,
                 generator=None, precision='bf16x3', conv='GAT', filters=None):
        super().__init__()
        if conv not in ('GAT', 'GCN', 'Diffusion'):
            raise NotImplementedError('conv=%r: GAT, GCN and Diffusion are built' % (conv,))
        if conv != 'GAT' and filters is None:
            raise ValueError("conv=%r needs filters=(preprocess(adj), preprocess(edge_adj)) of its layer class" % (conv,))
        self.conv, self.filters = conv, filters
        if precision not in _lib.PRECISION_FLAGS:
            raise ValueError("precision must be 'bf16x3' (fused kernel, split-bf16 MFMA, fp32 accumulate) or 'fp32' "
                             "(exact-fp32 unfused kernels), got %r" % (precision,))
        self.precision = precision
        if not isinstance(graph, DrainageGraph):
            raise TypeError('graph must be a gnn_uds_amd.graph.DrainageGraph')
        self.graph, self.d, self.h, self.activation = graph, int(embed_size), int(embed_size) // 2, activation
        if self.d % 8:
            raise ValueError('embed_size must be a multiple of 8 (float4 rows of width d/2), got %d' % self.d)
        if sparse_params is None:
            sparse_params = graph.n_node * graph.n_edge > (1 << 24)
        fx = self.d if fx is None else int(fx)
        fe = self.d if fe is None else int(fe)
        g = generator
        self.dense_xe = Dense(self.h, activation, in_features=fe, generator=g, precision=precision)          # emulator.py:225
        self.dense_ex = Dense(self.h, activation, in_features=fx, generator=g, precision=precision)          # emulator.py:226
        abs_n = CSR(graph.inc_n.rowptr, graph.inc_n.col, graph.n_node, graph.n_edge, np.abs(graph.inc_n.val))
        abs_e = CSR(graph.inc_e.rowptr, graph.inc_e.col, graph.n_edge, graph.n_node, np.abs(graph.inc_e.val))
        self.node_edge_n = NodeEdge(abs_n, sparse=sparse_params, generator=g)           # emulator.py:227
        self.node_edge_e = NodeEdge(abs_e, sparse=sparse_params, generator=g)           # emulator.py:228
        self.node_edge_n.precision = self.node_edge_e.precision = precision
        if conv == 'GAT':
            self.gat_x = GATConv(self.d, activation=activation, in_channels=fx + self.h, generator=g)   # :229
            self.gat_e = GATConv(self.d, activation=activation, in_channels=fe + self.h, generator=g)   # :230
            self.gat_x.precision = self.gat_e.precision = precision
        elif conv == 'GCN':
            self.gcn_x = GCNConv(self.d, activation=activation, in_channels=fx + self.h, generator=g)
            self.gcn_e = GCNConv(self.d, activation=activation, in_channels=fe + self.h, generator=g)
        else:
            self.gcn_x = DiffusionConv(self.d, activation=activation, generator=g)        # emulator.py:229-230 with net = DiffusionConv
            self.gcn_e = DiffusionConv(self.d, activation=activation, generator=g)
        self._net = net
        self._packed = None       # (parameter versions, packed bf16 hi/lo fragments) of the four GEMM kernels
        self.last_path = None     # which kernels the last forward ran: 'fused', 'fused+remainder', 'unfused' (tests, bench)

    def _packed_weights(self, p, fx, fe, remainder=False):
        ks = (self.dense_xe.kernel, self.dense_ex.kernel, self.gat_x.kernel, self.gat_e.kernel)
        key = tuple((k._version, k.data_ptr()) for k in ks) + (fx, fe, remainder)
        if self._packed is None or self._packed[0] != key:
            if remainder:
                # a trained dense NodeEdge bias through the fused kernel's 96-wide split-input variant: the dense remainder
                # `rest @ x_e` (S, R, 32) rides in as the 32 extra input columns, multiplied by the SAME rows of the GAT kernel
                # as the CSR aggregate ([x | rem | agg] @ [Wx; Wagg; Wagg] = [x | agg + rem] @ [Wx; Wagg]); the in-kernel
                # secondary MLP must not see those columns: zero rows under its kernel
                z = torch.zeros((self.h, self.h), device=p['xe_k'].device)
                aug = dict(xe_k=torch.cat([p['xe_k'], z]), ex_k=torch.cat([p['ex_k'], z]),
                           gx_k=torch.cat([p['gx_k'], p['gx_k'][-self.h:]]), ge_k=torch.cat([p['ge_k'], p['ge_k'][-self.h:]]))
                p = dict(p, **aug)
            else:
                aug = {}
            self._packed = (key, _lib.spatial_pack_weights(p, fx, fe, self.h, self.d), aug)
        return self._packed[1], self._packed[2]

    def _d128_remainder_ok(self, xs, es, xbs, ebs):
        if not (self.precision == 'bf16x3' and self.h == 64 and self.d == 128 and xs.shape[-1] == 128 and es.shape[-1] in (64, 128)
                and xbs is None and ebs is None):
            return False
        net = self.network()
        net.prepare(128, es.shape[-1])
        return bool(net.plan_info()['fused'] & (8 if es.shape[-1] == 128 else 16))

    def network(self):
        if self._net is None:
            self._net = _lib.NetworkHandle(self.graph)
        return self._net

    PACK_ROWS = 128       # rows of a full tile of the fused kernels (csrc/tile_plan.hpp: p_limit)

    def pack_factor(self):
        """Snapshots of this network that share one tile of the fused kernel: a network of 30 nodes fills a quarter of a
        128-row tile, and a tile costs its workgroup the same time a quarter full or full, so k = 128 // rows
        snapshots are laid side by side as k disjoint copies of the network (DrainageGraph.replicated)."""
        rows = max(self.graph.n_node, self.graph.n_edge)
        return max(1, self.PACK_ROWS // rows) if rows <= self.PACK_ROWS // 2 else 1

    def _replica(self, k):
        if getattr(self, '_rep', None) is None or self._rep[0] != k:
            self._rep = (k, _lib.NetworkHandle(self.graph.replicated(k)))
        return self._rep[1]

    def export_params(self):
        """Parameters as CPU tensors under the key names of uds_spatial_params_t (NodeEdge as
        'ne_*_w'/'ne_*_b' when dense, 'ne_*_v' support values when sparse)."""
        c = lambda t: None if t is None else t.detach().cpu().clone()
        if self.conv != 'GAT':
            raise NotImplementedError('export_params covers the GAT layer')
        p = dict(xe_k=c(self.dense_xe.kernel), xe_b=c(self.dense_xe.bias), ex_k=c(self.dense_ex.kernel),
                 ex_b=c(self.dense_ex.bias),
                 gx_k=c(self.gat_x.kernel), gx_as=c(self.gat_x.attn_kernel_self), gx_an=c(self.gat_x.attn_kernel_neighs),
                 gx_b=c(self.gat_x.bias),
                 ge_k=c(self.gat_e.kernel), ge_as=c(self.gat_e.attn_kernel_self), ge_an=c(self.gat_e.attn_kernel_neighs),
                 ge_b=c(self.gat_e.bias))
        for tag, ne in (('n', self.node_edge_n), ('e', self.node_edge_e)):
            if ne.sparse:
                p['ne_%s_v' % tag] = c(ne.support_values()[0])
            else:
                p['ne_%s_w' % tag], p['ne_%s_b' % tag] = c(ne.weight), c(ne.bias)
        return p

    def forward(self, x, e, xb=None, eb=None, adj_mask=None, attn_dropout=None):
        """xb / eb: 32 extra columns appended to a 64-wide x / e (`concat([x, b])`, emulator.py:260-262) -- read in place
        by the fused kernel; every other path concatenates.  adj_mask (..., nnz of the node adjacency): the per-snapshot
        adjacency of `use_adj` (emulator.py:268-271,282) -- node side through the masked aggregation kernel.
        attn_dropout: a DropoutStream = Keras' training mode for the two GATConv layers (Spektral's attention dropout, rate 0.5)."""
        if adj_mask is not None:
            if attn_dropout is not None:
                raise NotImplementedError('use_adj in training mode (attention dropout) is not built')
            if self.conv != 'GAT':
                raise NotImplementedError('use_adj is built for conv=GAT (GCN / Diffusion would re-normalise the filter per snapshot)')
            if xb is not None:
                x = torch.cat([x, xb], dim=-1)
            if eb is not None:
                e = torch.cat([e, eb], dim=-1)
            xs, lead_x = _flatten_snapshots(x)
            es, lead_e = _flatten_snapshots(e)
            net = self.network()
            x_e, e_x = self.dense_xe(es), self.dense_ex(xs)
            ox = self.gat_x([xs, net.adj], xb=self.node_edge_n(x_e), edge_mask=adj_mask)
            oe = self.gat_e([es, net.edge_adj], xb=self.node_edge_e(e_x))
            self.last_path = 'unfused'
            return ox.reshape(lead_x + ox.shape[-2:]), oe.reshape(lead_e + oe.shape[-2:])
        fused_split = (xb is not None or eb is not None) and self.conv == 'GAT' and self.precision == 'bf16x3' and \
            self.h == 32 and self.d == 64 and x.shape[-1] + (0 if xb is None else xb.shape[-1]) in (64, 96) and \
            e.shape[-1] + (0 if eb is None else eb.shape[-1]) in (64, 96) and (xb is None or (x.shape[-1], xb.shape[-1]) == (64, 32)) and \
            (eb is None or (e.shape[-1], eb.shape[-1]) == (64, 32)) and not _ag.grad_on(x, e, xb, eb, *self.parameters()) and \
            self.node_edge_n.support_values()[1] is None and self.node_edge_e.support_values()[1] is None
        if not fused_split:
            if xb is not None:
                x = torch.cat([x, xb], dim=-1)
            if eb is not None:
                e = torch.cat([e, eb], dim=-1)
            xb = eb = None
        xs, lead_x = _flatten_snapshots(x)
        es, lead_e = _flatten_snapshots(e)
        xbs = None if xb is None else _flatten_snapshots(xb)[0]
        ebs = None if eb is None else _flatten_snapshots(eb)[0]
        if self.conv != 'GAT':     # GCN a_hat @ ([x | agg] W) + b, or Diffusion: unfused composition of the Dense / NodeEdge / sparse kernels
            x_e, e_x = self.dense_xe(es), self.dense_ex(xs)
            ox = self.gcn_x([torch.cat([xs, self.node_edge_n(x_e)], dim=-1), self.filters[0]])
            oe = self.gcn_e([torch.cat([es, self.node_edge_e(e_x)], dim=-1), self.filters[1]])
            return ox.reshape(lead_x + ox.shape[-2:]), oe.reshape(lead_e + oe.shape[-2:])
        if attn_dropout is not None or _ag.grad_on(xs, es, *self.parameters()):
            # training: the unfused chain, every operator with its own HIP backward (autograd.py)
            net = self.network()
            x_e, e_x = self.dense_xe(es), self.dense_ex(xs)
            ox = self.gat_x([xs, net.adj], xb=self.node_edge_n(x_e), attn_dropout=attn_dropout)
            oe = self.gat_e([es, net.edge_adj], xb=self.node_edge_e(e_x), attn_dropout=attn_dropout)
            return ox.reshape(lead_x + ox.shape[-2:]), oe.reshape(lead_e + oe.shape[-2:])
        vn, rest_n = self.node_edge_n.support_values()
        ve, rest_e = self.node_edge_e.support_values()
        p = dict(xe_k=self.dense_xe.kernel, xe_b=self.dense_xe.bias, ex_k=self.dense_ex.kernel, ex_b=self.dense_ex.bias,
                 ne_n_val=vn, ne_e_val=ve,
                 gx_k=self.gat_x.kernel, gx_as=self.gat_x.attn_kernel_self, gx_an=self.gat_x.attn_kernel_neighs,
                 gx_b=self.gat_x.bias,
                 ge_k=self.gat_e.kernel, ge_as=self.gat_e.attn_kernel_self, ge_an=self.gat_e.attn_kernel_neighs,
                 ge_b=self.gat_e.bias)
        self.last_path = 'unfused'
        if rest_n is not None or rest_e is not None:
            # trained dense NodeEdge bias (every checkpoint the reference trains, emulator.py:36-45): secondary MLP on the
            # row-GEMM kernel, the dense remainder on the MFMA GEMM, everything else in the fused kernel
            net = self.network()

            def remainders():
                # rest_n @ Dense_xe(es) and rest_e @ Dense_ex(xs): the Dense outputs exist only for these products (the fused kernel has
                # its own secondary MLP), so they go straight into the GEMM's operand planes (uds_remainder_forward_dense) when the
                # shapes allow; else they are materialised first
                S_, N_, E_ = xs.shape[0], xs.shape[1], es.shape[1]
                if rest_n is None:
                    rem_n = torch.zeros((S_, N_, self.h), device=xs.device, dtype=torch.float32)
                else:
                    rem_n = self.node_edge_n.remainder_of_dense(es, self.dense_xe) if not os.environ.get('UDS_REMAINDER_MATERIALISE') else None
                    if rem_n is None:
                        rem_n = self.node_edge_n.remainder(self.dense_xe(es))
                if rest_e is None:
                    rem_e = torch.zeros((S_, E_, self.h), device=xs.device, dtype=torch.float32)
                else:
                    rem_e = self.node_edge_e.remainder_of_dense(xs, self.dense_ex) if not os.environ.get('UDS_REMAINDER_MATERIALISE') else None
                    if rem_e is None:
                        rem_e = self.node_edge_e.remainder(self.dense_ex(xs))
                return rem_n, rem_e

            if self.precision == 'bf16x3' and self.h == 32 and self.d == 64 and xs.shape[-1] == 64 and es.shape[-1] == 64 \
                    and xbs is None and ebs is None:
                rem_n, rem_e = remainders()
                ox = None
                if getattr(self, '_ws_rem_ok', True):
                    # the wave-specialised kernel adds the remainder to its NodeEdge aggregate (uds_spatial_layer_forward_rem); it
                    # refuses plans whose tiles it cannot hold -- then the 96-wide split-input kernel below carries the remainder
                    try:
                        net.prepare(64, 64)
                        ox, oe = _lib.spatial_layer_forward(net, dict(p, packed=self._packed_weights(p, 64, 64)[0]), xs, es, self.h, self.d,
                                                            self.activation, _lib.PRECISION_FLAGS[self.precision], rem_x=rem_n, rem_e=rem_e)
                    except _lib.UdsError:
                        self._ws_rem_ok = False
                if ox is None:
                    packed, aug = self._packed_weights(p, 96, 96, remainder=True)
                    net.prepare(96, 96)
                    ox, oe = _lib.spatial_layer_forward(net, dict(p, packed=packed, **aug), xs, es, self.h, self.d, self.activation,
                                                        _lib.PRECISION_FLAGS[self.precision], xb=rem_n, eb=rem_e)
                self.last_path = 'fused+remainder'
            elif self._d128_remainder_ok(xs, es, xbs, ebs):
                # the reference's stock model after training (embed_size 128, dense bias): the remainder is added to the support
                # aggregate inside the column-split kernel (uds_spatial_layer_forward_rem)
                rem_n, rem_e = remainders()
                packed = self._packed_weights(p, 128, es.shape[-1])[0]
                ox, oe = _lib.spatial_layer_forward(net, dict(p, packed=packed), xs, es, self.h, self.d, self.activation,
                                                    _lib.PRECISION_FLAGS[self.precision], rem_x=rem_n, rem_e=rem_e)
                self.last_path = 'fused+remainder'
            else:
                x_e, e_x = self.dense_xe(es), self.dense_ex(xs)
                ox = self.gat_x([xs, net.adj], xb=self.node_edge_n(x_e))
                oe = self.gat_e([es, net.edge_adj], xb=self.node_edge_e(e_x))
        else:
            fx = xs.shape[-1] + (0 if xbs is None else xbs.shape[-1])
            fe = es.shape[-1] + (0 if ebs is None else ebs.shape[-1])
            if self.precision == 'bf16x3' and self.h == 32 and self.d == 64 and fx in (64, 96) and fe in (64, 96):
                p['packed'] = self._packed_weights(p, fx, fe)[0]    # split once per parameter update, not per call
                self.network().prepare(fx, fe)                      # tile plans of the 96-wide variants are built on first use
            elif self.precision == 'bf16x3' and self.h == 64 and self.d == 128 and fx == 128 and fe in (64, 128) and xbs is None and ebs is None:
                self.network().prepare(128, fe)                     # d = 128 (the reference default): the column-split fused kernel
                if self.network().plan_info()['fused'] & (8 if fe == 128 else 16):
                    p['packed'] = self._packed_weights(p, fx, fe)[0]
            if 'packed' not in p and self.precision == 'bf16x3' and fx % 32 == 0 and fe % 32 == 0 and self.h % 16 == 0 and self.d % 16 == 0 \
                    and xs.shape[0] * xs.shape[1] >= 4096:
                # no fused kernel for this shape (d = 128: the reference's default embed_size): unfused composition with
                # the dense parts on the matrix cores instead of the exact-fp32 FMA kernels
                net = self.network()
                x_e, e_x = self.dense_xe(es), self.dense_ex(xs)
                ox = self.gat_x([xs, net.adj], xb=self.node_edge_n(x_e))
                oe = self.gat_e([es, net.edge_adj], xb=self.node_edge_e(e_x))
                return ox.reshape(lead_x + ox.shape[-2:]), oe.reshape(lead_e + oe.shape[-2:])
            k, S = self.pack_factor(), xs.shape[0]
            if 'packed' in p and k > 1 and S >= 16 * k:      # (a handful of snapshots -- the rollout's windows -- stay one launch)
                # small network: k snapshots per tile.  (S, N, F) -> (S / k, k N, F) is a view; the NodeEdge values repeat per copy.
                N, E, Sm = self.graph.n_node, self.graph.n_edge, S // k * k
                rep = self._replica(k)
                if self.d == 64:
                    rep.prepare(fx, fe)
                else:
                    rep.prepare(128, fe)
                pk = dict(p, ne_n_val=vn.repeat(k), ne_e_val=ve.repeat(k))
                v = lambda t, R: None if t is None else t[:Sm].reshape(Sm // k, k * R, t.shape[-1])
                ox, oe = _lib.spatial_layer_forward(rep, pk, v(xs, N), v(es, E), self.h, self.d, self.activation,
                                                    _lib.PRECISION_FLAGS[self.precision], xb=v(xbs, N), eb=v(ebs, E))
                ox, oe = ox.reshape(Sm, N, self.d), oe.reshape(Sm, E, self.d)
                if Sm < S:     # the last S mod k snapshots one per tile
                    t = lambda a: None if a is None else a[Sm:].contiguous()
                    rx, re = _lib.spatial_layer_forward(self.network(), p, t(xs), t(es), self.h, self.d, self.activation,
                                                        _lib.PRECISION_FLAGS[self.precision], xb=t(xbs), eb=t(ebs))
                    ox, oe = torch.cat([ox, rx]), torch.cat([oe, re])
                self.last_path = 'fused'
            else:
                ox, oe = _lib.spatial_layer_forward(self.network(), p, xs, es, self.h, self.d, self.activation,
                                                    _lib.PRECISION_FLAGS[self.precision], xb=xbs, eb=ebs)
                self.last_path = 'fused' if 'packed' in p else 'unfused'
        return ox.reshape(lead_x + ox.shape[-2:]), oe.reshape(lead_e + oe.shape[-2:])


class GraphBaseBlock(nn.Module):
    """The spatial block of the `graph_base` variants (`emulator.py:220-223,273-276`): ONE graph over the N nodes and the
    E links (`get_node_based_adj` / `get_edge_based_adj`), one conv per layer over `concat([x, e], axis=-2)`, split back into
    node and link rows.  `filt`: the combined (N+E) x (N+E) pattern as a `graph.CSR` with self loops (GAT) or the dense
    normalised filter (GCN)."""

    def __init__(self, n_node, n_edge, filt, embed_size, n_sp_layer, activation='relu', f_in=None, generator=None, conv='GAT',
                 precision='bf16x3'):
        super().__init__()
        if conv not in ('GAT', 'GCN', 'Diffusion'):
            raise NotImplementedError('conv=%r: GAT, GCN and Diffusion are built' % (conv,))
        self.n_node, self.n_edge, self.filt, self.conv = int(n_node), int(n_edge), filt, conv
        f_in = int(embed_size) if f_in is None else int(f_in)
        mk = {'GAT': GATConv, 'GCN': GCNConv, 'Diffusion': DiffusionConv}[conv]
        self.layers = nn.ModuleList([mk(embed_size, activation=activation, in_channels=(f_in if i == 0 else embed_size), generator=generator)
                                     for i in range(n_sp_layer)])
        for ly in self.layers:
            if conv == 'GAT':
                ly.precision = precision

    def forward(self, x, e, xb=None, eb=None, adj_mask=None, dropout=None, attn_dropout=None):
        if xb is not None:
            x = torch.cat([x, xb], dim=-1)
        if eb is not None:
            e = torch.cat([e, eb], dim=-1)
        if x.shape[-1] != e.shape[-1]:
            raise _lib.UdsError('graph_base stacks node and link rows: widths %d and %d differ' % (x.shape[-1], e.shape[-1]))
        if adj_mask is not None and self.conv != 'GAT':
            raise NotImplementedError('use_adj is built for conv=GAT')
        z = torch.cat([x, e], dim=-2)
        for ly in self.layers:
            if attn_dropout is not None and self.conv == 'GAT':
                z = ly([z, self.filt], edge_mask=adj_mask, attn_dropout=attn_dropout)
            else:
                z = ly([z, self.filt], edge_mask=adj_mask) if adj_mask is not None else ly([z, self.filt])
            if dropout is not None:      # Dropout on the node rows and on the link rows (emulator.py:234-235): one elementwise mask
                z = dropout(z)
        return z[..., :self.n_node, :].contiguous(), z[..., self.n_node:, :].contiguous()


class Dropout(nn.Module):
    """`keras.layers.Dropout(rate)` as the emulator uses it (`emulator.py:199-213,234-235,287-288,314-318`): identity unless
    called with training=True (`self.model(inp, training=fit)`, :411,434), then inverted dropout on the HIP kernel (uds_dropout,
    counter-based Philox4x32-10: the mask is a function of (seed, offset + element index), recomputed in the backward pass).
    `seed` comes from the generator given at construction (or torch's default one); every call consumes x.numel() positions of
    the stream, so successive calls and successive layers sharing one DropoutStream draw disjoint masks."""

    def __init__(self, rate, stream=None, generator=None):
        super().__init__()
        if not 0.0 <= rate < 1.0:
            raise ValueError('dropout rate %r outside [0, 1)' % (rate,))
        self.rate = float(rate)
        self.stream = stream if stream is not None else DropoutStream(generator=generator)

    def forward(self, x, training=False):
        if not training or self.rate == 0.0:
            return x
        x = x.contiguous()
        offset = self.stream.take(x.numel())
        return _ag.DropoutFn.apply(x, self.rate, self.stream.seed, offset)


class DropoutStream:
    """(seed, running offset) of one counter-based random stream shared by the Dropout layers of a model."""

    def __init__(self, seed=None, generator=None):
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,), generator=generator).item())
        self.seed, self.offset = int(seed), 0

    def take(self, n):
        o = self.offset
        self.offset += (int(n) + 3) // 4 * 4      # whole counters: every call starts on the float4 path of the kernel
        return o

    def reseed(self, seed):
        self.seed, self.offset = int(seed), 0


class SpatialBlock(nn.Module):
    """`for _ in range(n_sp_layer)` (`emulator.py:219-235`): first layer takes (fx, fe) features."""

    def __init__(self, graph, embed_size, n_sp_layer, activation='relu', fx=None, fe=None, sparse_params=None,
                 generator=None, precision='bf16x3', conv='GAT', filters=None):
        super().__init__()
        layers = []
        for i in range(n_sp_layer):
            layers.append(SpatialLayer(graph, embed_size, activation, fx if i == 0 else None, fe if i == 0 else None,
                                       sparse_params, generator=generator, precision=precision, conv=conv, filters=filters))
        self.layers = nn.ModuleList(layers)
        self.graph = graph

    def graphed(self, x, e):
        """The block's inference forward on FIXED input buffers as one captured HIP graph: returns `replay()` -> (x', e'), which
        re-launches the L layer kernels back to back without the host work between them (argument packing, output allocation,
        launch latency: ~3 us per layer at the headline size, more than the layers themselves on small networks).  x, e are
        read in place at every replay -- write new snapshots into them; the outputs are overwritten by the next replay.
        Parameters must not change between capture and replay (packed weights are baked in): capture again after an update."""
        if not (x.is_cuda and e.is_cuda):
            raise _lib.UdsError('SpatialBlock.graphed needs device tensors')
        side = torch.cuda.Stream(device=x.device)
        side.wait_stream(torch.cuda.current_stream(x.device))
        with torch.cuda.stream(side), torch.no_grad():      # warm-up outside the capture: tile plans, packed weights, allocator
            for _ in range(2):
                self.forward(x, e)
        torch.cuda.current_stream(x.device).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph), torch.no_grad():
            out = self.forward(x, e)

        def replay():
            graph.replay()
            return out
        replay.graph = graph
        return replay

    def forward(self, x, e, xb=None, eb=None, adj_mask=None, dropout=None, attn_dropout=None):
        """xb / eb: extra input columns of the FIRST layer (`concat([x, b])` before block 2, emulator.py:260-262);
        adj_mask: the per-snapshot node adjacency of `use_adj`, seen by every layer of the block (emulator.py:268-282);
        dropout: callable applied to x and e after every layer (`Dropout(self.dropout)`, emulator.py:234-235,287-288), training only;
        attn_dropout: DropoutStream for Spektral's attention dropout inside the GATConv layers, training only."""
        net = self.layers[0].network() if self.layers[0].conv == 'GAT' else None
        for i, layer in enumerate(self.layers):
            layer._net = net
            kw = {'attn_dropout': attn_dropout} if attn_dropout is not None and layer.conv == 'GAT' else {}
            x, e = layer(x, e, xb if i == 0 else None, eb if i == 0 else None, adj_mask=adj_mask, **kw)
            if dropout is not None:
                x, e = dropout(x), dropout(e)
        return x, e
