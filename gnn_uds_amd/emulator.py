"""`Emulator`: the reference's surrogate model (`surrogate/emulator.py:47-852`) on the HIP engine.

Same constructor (`Emulator(conv, resnet, recurrent, args)`, attributes read from `args` with the reference's
names and defaults, emulator.py:48-127), same method names, positional order and tensor layouts
(`(B,T,N,C)` / `(B,T,E,C)`), re-expressed as a `torch.nn.Module`:

    forward / model          build_network                      emulator.py:166-341
    predict, predict_tf      raw states -> de-normalised (B,T,N,5), (B,T,E,3)        :566-641
    _model                   single chunk or `roll` autoregressive chunks            :400-438
    simulate                 sliding-window evaluation of an event                   :521-564
    post_proc_tf, constrain_tf, normalize, set_norm, get_edge_action, get_action     :364-398,680-810
    save / load              weights + norm_*.npy                                    :814-852

What runs where: every layer of the network (embedding / fusion / head Dense, NodeEdge, GAT / GCN / Diffusion, causal dilated
Conv1D, the resnet prefix sum) and the link->node flow balance of post_proc_tf are HIP kernels behind the C ABI.
The remaining post-processing is elementwise gating / clipping on tensors already in HBM and is written with torch
tensor ops (device plumbing).  Training (`fit_eval`, GradNorm), `graph_base` 1 / 2, GCN and DiffusionConv are built.
`use_adj` (per-time-step adjacency rewritten by the control action) is built for GAT as a mask over the CSR entries.
GRU / LSTM temporal nets run (inference and, at 64 units, training).  Training-time dropout: `layers.Dropout` on `uds_dropout`.  Not built, each raises: `use_adj` with GCN / Diffusion or under autograd,
GeneralConv (a sparse-mode-only Spektral layer the reference's dense call cannot run either).  conv = False -- the reference's non-graph
baseline, its shipped `*_nncat_*` models -- runs on the same Dense / temporal / cumsum kernels (`_forward_mlp`).
"""
import os

import numpy as np
import torch
from torch import nn

from . import _lib
from .graph import DrainageGraph, csr_from_dense
from . import autograd as _ag
from .graph import csr_from_dense, edge_based_adj_csr, node_based_adj_csr
from .layers import Dense, Dropout, DropoutStream, GraphBaseBlock, SpatialBlock, _glorot_uniform, _packed_kernel, _param


class Conv1D(nn.Module):
    """keras Conv1D(filters, kernel_size, padding='causal', dilation_rate, activation) along the time axis of
    x (B, T, R, F) -- the reference transposes to (B*R, T, F) first (emulator.py:244), this layout needs no transpose."""

    def __init__(self, filters, kernel_size, dilation_rate=1, activation=None, in_features=None, generator=None,
                 precision='fp32'):
        super().__init__()
        self.filters, self.kernel_size, self.dilation_rate = int(filters), int(kernel_size), int(dilation_rate)
        self.precision = precision
        self.activation = activation or 'linear'
        self.kernel = _param(_glorot_uniform((self.kernel_size, int(in_features), self.filters), 'cpu', generator))
        self.bias = _param(torch.zeros(self.filters))

    def forward(self, x):
        if _ag.grad_on(x, self.kernel, self.bias):
            return _ag.Conv1DFn.apply(x, self.kernel, self.bias, self)
        x = x.contiguous()
        k, f, h = self.kernel.shape
        if self.precision == 'bf16x3' and _lib.rowgemm_supported(k * f, f, h):     # matrix-core path (split-bf16, 3 products)
            return _lib.rowgemm_forward(x, _packed_kernel(self, self.kernel.reshape(k * f, h)), self.bias, h, self.activation,
                                        taps=k, dilation=self.dilation_rate)
        return _lib.conv1d_causal(x, self.kernel, self.bias, self.dilation_rate, self.activation)


class _DenseShim:
    """What DenseFn reads of its module, for a bare (kernel, bias) pair."""

    def __init__(self, precision, units):
        self.precision, self.units = precision, units


class _Recurrent(nn.Module):
    """keras GRU / LSTM(units, return_sequences=True) along the time axis of x (B, T, R, F) (emulator.py:158-161): the input
    projection of every time step is one Dense launch, the recurrence one streaming kernel (uds_recurrent_forward), exact
    fp32; a 64 -> 64 layer with precision='bf16x3' is ONE launch on the matrix cores (uds_recurrent_fused).  Parameters keep the Keras names and shapes -- `kernel` (F, G*units), `recurrent_kernel` (units, G*units), `bias`
    ((2, 3*units) for the TF2 GRU with reset_after=True: input and recurrent bias; (4*units,) for the LSTM) -- and the
    Keras initialisers (glorot_uniform, orthogonal, zeros with the LSTM's forget-gate bias at one).  Under autograd (training) the
    layer runs Dense + `RecurrentFn` (autograd.py): back-propagation through time is one HIP launch (uds_recurrent_backward)."""
    KIND, G = None, 0

    def __init__(self, units, return_sequences=True, in_features=None, generator=None, precision='fp32'):
        super().__init__()
        self.precision, self._packed = precision, None
        if not return_sequences:
            raise NotImplementedError('the emulator uses return_sequences=True (emulator.py:159,161)')
        self.units = int(units)
        gh = self.G * self.units
        self.kernel = _param(_glorot_uniform((int(in_features), gh), 'cpu', generator))
        q, _ = torch.linalg.qr(torch.randn(gh, self.units, generator=generator))            # orthogonal initialiser
        self.recurrent_kernel = _param(q.T.contiguous())
        if self.KIND == 'GRU':
            self.bias = _param(torch.zeros(2, gh))
        else:
            b = torch.zeros(gh)
            b[self.units:2 * self.units] = 1.0                                              # unit_forget_bias
            self.bias = _param(b)

    def forward(self, x):
        if _ag.grad_on(x, self.kernel, self.recurrent_kernel, self.bias):
            # training (fit_eval, emulator.py:457-484): input projection through the Dense operator, the recurrence through
            # RecurrentFn (exact-fp32 forward, back-propagation through time on the matrix cores)
            if self.units != 64:
                raise NotImplementedError('the %s backward kernel is built for 64 units (hidden_dim = 64, the reference default)' % self.KIND)
            b_in, b_rec = (self.bias[0], self.bias[1]) if self.KIND == 'GRU' else (self.bias, None)
            shim = _DenseShim(self.precision, self.G * self.units)
            xp = _ag.DenseFn.apply(x, self.kernel, b_in.contiguous(), shim, 'linear')
            return _ag.RecurrentFn.apply(xp, self.recurrent_kernel, None if b_rec is None else b_rec.contiguous(), self.KIND, self.precision)
        x = x.contiguous()
        b_in, b_rec = (self.bias[0].contiguous(), self.bias[1].contiguous()) if self.KIND == 'GRU' else (self.bias, None)
        F = x.shape[-1]
        if self.precision == 'bf16x3' and self.units == 64 and F % 32 == 0:
            key = (self.kernel._version, self.recurrent_kernel._version, self.bias._version, self.kernel.data_ptr(), self.recurrent_kernel.data_ptr())
            direct = _lib.recurrent_fused_supported(F, self.KIND)
            if self._packed is None or self._packed[0] != key:
                if direct:
                    self._packed = (key, _lib.recurrent_pack(self.kernel, self.recurrent_kernel))
                else:     # W + U do not fit the LDS together (LSTM at 128) or another width: the projection on the row-GEMM kernel
                    G = self.G
                    fold = b_in.clone()
                    if b_rec is not None:
                        fold[:2 * 64] += b_rec[:2 * 64]               # the z / r gates add both biases (the candidate's stays apart)
                    self._packed = (key, _lib.recurrent_pack(None, self.recurrent_kernel),
                                    [_lib.rowgemm_pack(self.kernel[:, 64 * g:64 * (g + 1)].contiguous()) for g in range(G)], fold)
            if direct:
                # the whole layer in one launch on the matrix cores (input projection never written out, state fed back in registers)
                return _lib.recurrent_fused(x, self._packed[1], b_in, b_rec, self.KIND)
            if _lib.rowgemm_supported(F, F, 64):
                _, packed_u, packed_w, fold = self._packed
                xp = torch.empty(x.shape[:-1] + (self.G * 64,), device=x.device, dtype=torch.float32)
                for g, pk in enumerate(packed_w):
                    _lib.rowgemm_cat(x, None, pk, fold[64 * g:64 * (g + 1)].contiguous(), 64, 'linear', out=xp, col0=64 * g)
                return _lib.recurrent_fused(xp, packed_u, None, b_rec, self.KIND, projected=True)
        xp = _lib.dense_act(x, self.kernel, b_in, 'linear')
        return _lib.recurrent_forward(xp, self.recurrent_kernel, b_rec, self.KIND)


class GRU(_Recurrent):
    KIND, G = 'GRU', 3


class LSTM(_Recurrent):
    KIND, G = 'LSTM', 4


class KerasAdam:
    """keras.optimizers.Adam(learning_rate, clipnorm) as TF 2.10 applies it (emulator.py:111): every gradient is clipped
    by ITS OWN norm (tf.clip_by_norm), then m, v are updated and var -= lr * sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v) + 1e-7)."""

    def __init__(self, params, learning_rate=1e-3, beta_1=0.9, beta_2=0.999, epsilon=1e-7, clipnorm=None):
        self.params = list(params)
        self.lr, self.b1, self.b2, self.eps, self.clipnorm = learning_rate, beta_1, beta_2, epsilon, clipnorm
        self.t = 0
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]

    def state_dict(self):
        """Iteration count + first / second moments (what `optimizer.get_weights()` holds in Keras, emulator.py:826)."""
        return {'t': self.t, 'm': [t.detach().cpu() for t in self.m], 'v': [t.detach().cpu() for t in self.v]}

    def load_state_dict(self, sd):
        if len(sd['m']) != len(self.params) or any(tuple(a.shape) != tuple(p.shape) for a, p in zip(sd['m'], self.params)):
            raise ValueError('optimizer state does not match the model parameters')
        self.t = int(sd['t'])
        for dst, src in zip(self.m, sd['m']):
            dst.copy_(src)
        for dst, src in zip(self.v, sd['v']):
            dst.copy_(src)

    @torch.no_grad()
    def step(self):
        self.t += 1
        lr_t = self.lr * (1 - self.b2 ** self.t) ** 0.5 / (1 - self.b1 ** self.t)
        idx = [i for i, p in enumerate(self.params) if p.grad is not None]
        if not idx:
            return
        ps, gs = [self.params[i] for i in idx], [self.params[i].grad for i in idx]
        ms, vs = [self.m[i] for i in idx], [self.v[i] for i in idx]
        norms = torch.stack(torch._foreach_norm(gs))                      # one launch group, one host sync below
        if not bool(torch.isfinite(norms).all()):
            raise FloatingPointError('grads contain NaN/Inf!')
        if self.clipnorm is not None:
            gs = torch._foreach_mul(gs, list((self.clipnorm / torch.clamp(norms, min=self.clipnorm)).unbind()))
        torch._foreach_mul_(ms, self.b1)
        torch._foreach_add_(ms, gs, alpha=1 - self.b1)
        torch._foreach_mul_(vs, self.b2)
        torch._foreach_addcmul_(vs, gs, gs, value=1 - self.b2)
        den = torch._foreach_sqrt(vs)
        torch._foreach_add_(den, self.eps)
        torch._foreach_addcdiv_(ps, ms, den, value=-lr_t)


def _numeric_suffix(name):
    tail = name.rsplit('_', 1)[-1]
    return int(tail) if tail.isdigit() else 0


def _find_keras_weight(weights, lname, kname):
    """The array Keras saved for weight `kname` of layer `lname`, or None.  Tried in turn: the exact names (`<layer>/<w>:0`,
    the HDF5 group path `<layer>/<layer>/<w>:0`, `<layer>/<w>`); for a weight nested in a recurrent cell (`<cell>/<w>`) ANY
    single `*_cell*` sub-group of the layer (the cell's auto-number depends on how many cells the Keras session had created
    before, which the checkpoint does not say); for a DiffusionConv kernel the `diffuse_features*` sub-layers' (K + 1,)
    kernels stacked in numeric order into the (channels, K + 1) array the module keeps."""
    for key in ('%s/%s:0' % (lname, kname), '%s/%s/%s:0' % (lname, lname, kname), '%s/%s' % (lname, kname)):
        if key in weights:
            return weights[key]
    mine = [k for k in weights if isinstance(k, str) and k.split('/')[0] == lname]
    if '/' in kname:                                   # '<cell>/<weight>': match the cell sub-group by suffix
        tail = '/' + kname.split('/', 1)[1]
        hits = [k for k in mine if '_cell' in k and (k.endswith(tail + ':0') or k.endswith(tail))]
        if len(hits) == 1:
            return weights[hits[0]]
        return None
    if kname == 'kernel':
        subs = [k for k in mine if 'diffuse_features' in k and (k.endswith('/kernel:0') or k.endswith('/kernel'))]
        if subs:
            subs.sort(key=lambda k: _numeric_suffix([p for p in k.split('/') if p.startswith('diffuse_features')][-1]))
            return np.stack([np.asarray(weights[k]).reshape(-1) for k in subs])
    return None


class Emulator(nn.Module):
    def __init__(self, conv=None, resnet=False, recurrent=None, args=None, precision='bf16x3', generator=None):
        super().__init__()
        g = lambda k, d=None: getattr(args, k, d)
        self.n_node, self.n_in = g('state_shape', (40, 4))
        self.tide = bool(g('tide', False))
        self.b_in = 2 if self.tide else 1
        act = g('act', False)
        self.act = bool(act and act != 'False')
        self.n_out = self.n_in - 1
        self.seq_in, self.seq_out = g('seq_in', 6), g('seq_out', 1)
        self.embed_size, self.hidden_dim = g('embed_size', 64), g('hidden_dim', 64)
        self.kernel_size, self.n_sp_layer, self.n_tp_layer = g('kernel_size', 3), g('n_sp_layer', 3), g('n_tp_layer', 2)
        self.dropout = g('dropout', 0.0)
        self.activation = g('activation', 'relu')
        self.if_flood = int(g('if_flood', 0))
        if self.if_flood:
            self.n_in += 1
        self.epsilon = g('epsilon', -1.0)
        self.graph_base = g('graph_base', 0)
        self.edge_fusion = bool(g('edge_fusion', False))
        self.edges = np.asarray(g('edges'))
        self.n_edge, self.e_in = g('edge_state_shape', (40, 4))
        self.e_out = self.e_in - 1
        if self.edge_fusion:
            self.n_out -= 2
        self.use_adj = bool(g('use_adj', False)) if self.act else False
        self.act_edges = np.asarray(g('act_edges', np.zeros((0, 2), dtype=int))) if self.act else None
        self.roll = int(g('roll', 0))
        self.model_dir = g('model_dir')
        self.balance = bool(g('balance', False))
        self.gradnorm = bool(g('gradnorm', False))
        self.learning_rate = float(g('learning_rate', 1e-3))
        self._args, self._optimizer, self._lw = args, None, None
        self.resnet = bool(resnet)
        self.conv = False if conv in (None, 'None', 'False', 'NoneType', False) else conv
        self.recurrent = recurrent
        if not self.conv:
            # the reference's non-graph baseline (its shipped `*_nncat_*` models, `conv: 'False'`): node and link states flattened into
            # one vector per time step, Dense layers instead of graph convolutions (emulator.py:181-182,197-212,236-237); the graph
            # is still needed by the post-processing (link -> node flow balance)
            self.conv_kind = 'GAT'
        elif 'GAT' in self.conv:
            self.conv_kind = 'GAT'
        elif 'GCN' in self.conv:
            self.conv_kind = 'GCN'
        elif 'Diff' in self.conv:                      # emulator.py:135-138
            self.conv_kind = 'Diffusion'
        else:
            # GeneralConv (emulator.py:146-149) is a sparse-mode-only Spektral layer: the reference's dense mixed-mode call
            # does not run with it either
            raise NotImplementedError('conv=%r is not built (GAT, GCN and Diffusion are)' % (conv,))
        if self.use_adj and self.conv_kind != 'GAT':
            raise NotImplementedError('use_adj is built for conv=GAT (GCN / Diffusion re-normalise the rewritten adjacency per time step: '
                                      'emulator.py:355-358)')
        if self.graph_base not in (0, 1, 2):
            raise ValueError('graph_base must be 0, 1 (node-based) or 2 (edge-based), got %r' % (self.graph_base,))
        if recurrent not in ('Conv1D', 'GRU', 'LSTM', None, 'None', False):
            raise NotImplementedError('recurrent=%r: Conv1D, GRU, LSTM or none (emulator.py:154-163)' % (recurrent,))
        # keras Dropout layers (:199-213,234-235,287-288,314-318) and Spektral's attention dropout inside every GATConv (rate 0.5): active
        # only in `forward(..., training=True)` = `fit_eval(fit=True)` (`self.model(inp, training=fit)`, :411,434); one counter-based
        # stream for all of them (layers.Dropout, GATConv(attn_dropout=))
        self.dropout_stream = DropoutStream(generator=generator) if self.dropout else None
        self._drop02 = Dropout(0.2, self.dropout_stream) if self.dropout else None
        self._drop = Dropout(self.dropout, self.dropout_stream) if self.dropout else None

        graph = g('graph')
        self._base_filter = None
        if self.graph_base:
            # ONE graph over nodes and links (base.py:471-532); `args.adj` is then the combined (N+E) x (N+E) matrix
            # (base.py:320-323) -- or, with `args.graph`, the pattern is rebuilt in CSR from the link list
            self.graph = graph if isinstance(graph, DrainageGraph) else DrainageGraph.from_edges(self.edges, self.n_node)
            node_edge = None if isinstance(graph, DrainageGraph) else np.asarray(g('node_edge'), dtype=np.float64)
            if g('adj') is not None and not isinstance(graph, DrainageGraph):
                adj = np.asarray(g('adj'))
                if adj.shape != (self.n_node + self.n_edge,) * 2:
                    raise ValueError('graph_base needs the combined (N+E, N+E) adjacency, got %r' % (adj.shape,))
                if self.conv_kind != 'GAT':
                    from .layers import DiffusionConv, GCNConv
                    self._base_filter = (GCNConv if self.conv_kind == 'GCN' else DiffusionConv).preprocess(adj)
                else:
                    self._base_filter = csr_from_dense((adj > 0).astype(int), add_self_loops=True)
            else:
                if self.conv_kind != 'GAT':
                    raise NotImplementedError('graph_base with a CSR graph is built for conv=GAT')
                build = node_based_adj_csr if self.graph_base == 1 else edge_based_adj_csr
                self._base_filter = build(self.edges, self.n_node, bool(g('directed', False)), int(g('order', 1)), g('length', 0), g('lengths', None))
            self.filter = self.edge_filter = None
        elif isinstance(graph, DrainageGraph):
            # large networks: `args.graph` (CSR, e.g. DrainageGraph.from_edges) instead of the dense (N,N) / (E,E) / (N,E)
            # matrices of `args.adj`, `args.edge_adj`, `args.node_edge`, which cannot exist at N >= 50k
            if self.conv_kind != 'GAT':
                raise NotImplementedError('args.graph (CSR input) is built for conv=GAT')
            self.graph, self.filter, self.edge_filter, node_edge = graph, None, None, None
        else:
            adj = np.asarray(g('adj', np.eye(self.n_node)))
            edge_adj = np.asarray(g('edge_adj', np.eye(self.n_edge)))
            node_edge = np.asarray(g('node_edge'), dtype=np.float64)
            self.graph = DrainageGraph.from_dense(adj, edge_adj, node_edge, self.edges)
            if self.use_adj:
                # get_adj_action rewrites entries of the RAW adjacency and casts to int (emulator.py:343-361): with a weighted
                # adjacency (length > 0: Gaussian kernel values < 1) that cast removes every off-diagonal entry, actuated or
                # not -- kept as the reference has it: the raw values at the pattern's entries (the forced diagonal reads 0)
                ga = self.graph.adj
                rows = np.repeat(np.arange(ga.n_rows), np.diff(ga.rowptr))
                self._adj_raw = np.asarray(adj, dtype=np.float64)[rows, np.asarray(ga.col, dtype=np.int64)]
            if self.conv_kind != 'GAT':
                from .layers import DiffusionConv, GCNConv
                pre = (GCNConv if self.conv_kind == 'GCN' else DiffusionConv).preprocess
                self.filter, self.edge_filter = pre(adj), pre(edge_adj)                                   # emulator.py:133-134,137-138
            else:
                self.filter, self.edge_filter = (adj > 0).astype(int), (edge_adj > 0).astype(int)        # emulator.py:143-145

        # per-node / per-link physical constants (emulator.py:71-98), kept as float32 buffers
        vec = lambda k, n, d: torch.as_tensor(np.asarray(g(k, np.full(n, d)), dtype=np.float64), dtype=torch.float32)
        for name, n, d in (('is_outfall', self.n_node, 0.0), ('area', self.n_node, 0.0), ('pump_in', self.n_node, 0.0),
                           ('pump_out', self.n_node, 0.0), ('hmax', self.n_node, 1.5), ('hmin', self.n_node, 0.0),
                           ('ehmax', self.n_edge, 0.5), ('pump', self.n_edge, 0.0), ('offset', self.n_edge, 0.0)):
            self.register_buffer(name, vec(name, n, d), persistent=False)
        # dense signed incidence: only the offset / pump branches of post_proc_tf use it (small networks)
        self.register_buffer('node_edge', None if node_edge is None else torch.as_tensor(node_edge, dtype=torch.float32), persistent=False)
        self.register_buffer('_inc_sign', torch.as_tensor(self.graph.inc_n.val, dtype=torch.float32), persistent=False)
        self._inc_handle = None
        self._norms = {}
        # host-side facts about the constant buffers (the reference tests them per call, emulator.py:688,698,630): decided once,
        # so the forward path has no device -> host synchronisation (and can be captured into a HIP graph)
        self._has_offset = bool(float(self.offset.max()) > 0) if self.n_edge else False
        self._has_pump = bool(float(self.pump.min()) > 0) if self.n_edge else False             # `self.pump.min() > 0` (:698)
        self._has_link_pump = bool(float(self.pump.abs().max()) > 0) if self.n_edge else False   # any rated link pump (NumPy-mode override, :656-659)
        self._has_any_pump = bool(float(self.pump_in.sum() + self.pump_out.sum() + self.pump.sum()) > 0)
        self._idx_cache = {}
        self._norm_derived = {}
        self._graph = None

        d, h, H, L, gen = self.embed_size, self.embed_size // 2, self.hidden_dim, self.n_sp_layer, generator
        a = self.activation
        pr = precision
        if not self.conv:
            self._build_mlp(d, h, H, L, a, gen, pr)
            return
        self.embed_x = Dense(d, 'linear', in_features=self.n_in, generator=gen)                 # emulator.py:198
        self.embed_b = Dense(h, a, in_features=self.b_in, generator=gen)                        # :203
        self.embed_e = Dense(d, 'linear', in_features=self.e_in, generator=gen)                 # :206
        self.embed_ae = Dense(h, a, in_features=1, generator=gen) if self.act else None         # :212
        # NodeEdge parameters: the reference's dense (R, M) pair, or one (weight, bias) per support entry (`args.sparse_params`,
        # default: above 16M matrix entries) -- the latter cannot hold a bias trained off the support (load_keras_weights raises)
        sp = getattr(args, 'sparse_params', None)
        sp = self.n_node * self.n_edge > (1 << 24) if sp is None else bool(sp)
        if self.graph_base:
            self.block1 = GraphBaseBlock(self.n_node, self.n_edge, self._base_filter, d, L, a, generator=gen, conv=self.conv_kind,
                                         precision=precision)                                                          # :220-223
        else:
            self.block1 = SpatialBlock(self.graph, d, L, a, sparse_params=sp, generator=gen, precision=precision,
                                       conv=self.conv_kind, filters=(self.filter, self.edge_filter))                  # :219-235
        rec = self.recurrent if self.recurrent in ('Conv1D', 'GRU', 'LSTM') else None              # get_tem_nets, :154-163

        def tem(f):
            if rec == 'Conv1D':
                mods = [Conv1D(H, self.kernel_size, 2 ** i, a, in_features=(f if i == 0 else H), generator=gen, precision=pr)
                        for i in range(self.n_tp_layer)]
            elif rec:
                mods = [(GRU if rec == 'GRU' else LSTM)(H, in_features=(f if i == 0 else H), generator=gen, precision=pr) for i in range(self.n_tp_layer)]
            else:
                mods = []                                                                       # no temporal net: widths stay d
            return nn.ModuleList(mods)
        if not (rec and self.n_tp_layer):
            H = d
        self.tem1_x, self.tem1_e = tem(d), tem(d)                                               # :247,254
        fx2, fe2 = H + h, H + (h if self.act else 0)
        if self.graph_base:
            if fx2 != fe2:
                raise ValueError('graph_base stacks node and link rows in block 2: needs act=True (equal widths %d / %d)' % (fx2, fe2))
            self.block2 = GraphBaseBlock(self.n_node, self.n_edge, self._base_filter, d, L, a, f_in=fx2, generator=gen,
                                         conv=self.conv_kind, precision=precision)                                     # :273-276
        else:
            self.block2 = SpatialBlock(self.graph, d, L, a, fx=fx2, fe=fe2, sparse_params=sp, generator=gen, precision=precision,
                                       conv=self.conv_kind, filters=(self.filter, self.edge_filter))                  # :272-288
        self.tem2_x, self.tem2_e = tem(d), tem(d)                                               # :302,308
        self.res_x = Dense(d, 'linear' if self.resnet else a, in_features=H, generator=gen, precision=pr)     # :313 'dense_resx'
        self.res_e = Dense(d, 'linear' if self.resnet else a, in_features=H, generator=gen, precision=pr)     # :317
        self.out = Dense(self.n_out, 'hard_sigmoid', in_features=d, generator=gen, precision=pr)              # :324
        fl, fi = [], d
        for _ in range(self.if_flood):
            fl.append(Dense(h, a, in_features=fi, generator=gen, precision=pr))                               # :329
            fi = h
        self.flood = nn.ModuleList(fl)
        self.flood_out = Dense(1, 'sigmoid', in_features=fi, generator=gen, precision=pr) if self.if_flood else None      # :330
        self.e_out_layer = Dense(self.e_out, 'tanh', in_features=d, generator=gen, precision=pr)              # :336

    # ------------------------------------------------------------------ the non-graph baseline (conv False)
    def _build_mlp(self, d, h, H, L, a, gen, pr):
        """`build_network(conv=False)` (emulator.py:166-341 with `net = Dense`): every time step's node states (N * n_in values) and link
        states (E * e_in) are ONE row each; embeddings, `Dense(2 d)(concat([x, e]))` + split in place of the spatial layers, the same
        temporal nets / resnet head, and heads that emit all N (E) outputs of a step at once.  Same Dense / Conv1D / recurrent /
        cumsum kernels as the graph model, on (B, T, 1, F) rows."""
        if self.seq_in != self.seq_out:
            raise NotImplementedError('conv=False reshapes the boundary input with seq_in (emulator.py:202): it only runs with seq_in == seq_out')
        N, E = self.n_node, self.n_edge
        self.embed_x = Dense(d, 'linear', in_features=N * self.n_in, generator=gen)             # :197-198
        self.embed_b = Dense(h, a, in_features=N * self.b_in, generator=gen)                    # :202-203
        self.embed_e = Dense(d, 'linear', in_features=E * self.e_in, generator=gen)             # :205-206
        self.embed_ae = Dense(h, a, in_features=E, generator=gen) if self.act else None         # :211-212
        rec = self.recurrent if self.recurrent in ('Conv1D', 'GRU', 'LSTM') else None

        def tem(f):
            if rec == 'Conv1D':
                return nn.ModuleList([Conv1D(H, self.kernel_size, 2 ** i, a, in_features=(f if i == 0 else H), generator=gen, precision=pr)
                                      for i in range(self.n_tp_layer)])
            if rec:
                return nn.ModuleList([(GRU if rec == 'GRU' else LSTM)(H, in_features=(f if i == 0 else H), generator=gen, precision=pr)
                                      for i in range(self.n_tp_layer)])
            return nn.ModuleList([])
        if not (rec and self.n_tp_layer):
            H = d
        self.block1 = nn.ModuleList([Dense(2 * d, a, in_features=2 * d, generator=gen, precision=pr) for _ in range(L)])          # :236-237
        self.tem1_x, self.tem1_e = tem(d), tem(d)
        f2 = (H + h) + (H + (h if self.act else 0))
        self.block2 = nn.ModuleList([Dense(2 * d, a, in_features=(f2 if i == 0 else 2 * d), generator=gen, precision=pr) for i in range(L)])
        self.tem2_x, self.tem2_e = tem(d), tem(d)
        self.res_x = Dense(d, 'linear' if self.resnet else a, in_features=H, generator=gen, precision=pr)     # :313
        self.res_e = Dense(d, 'linear' if self.resnet else a, in_features=H, generator=gen, precision=pr)     # :317
        self.out = Dense(self.n_out * N, 'hard_sigmoid', in_features=d, generator=gen, precision=pr)          # :322-325
        fl, fi = [], d
        for _ in range(self.if_flood):
            fl.append(Dense(h, a, in_features=fi, generator=gen, precision=pr))
            fi = h
        self.flood = nn.ModuleList(fl)
        self.flood_out = Dense(N, 'sigmoid', in_features=fi, generator=gen, precision=pr) if self.if_flood else None      # :327-330
        self.e_out_layer = Dense(self.e_out * E, 'tanh', in_features=d, generator=gen, precision=pr)          # :335-337

    def _forward_mlp(self, X, B, E, AE=None, training=False):
        nb, T = X.shape[0], self.seq_out
        drop = bool(self.dropout) and training
        d02 = (lambda t: self._drop02(t, True)) if drop else (lambda t: t)
        dr = (lambda t: self._drop(t, True)) if drop else (lambda t: t)
        flat = lambda t: t.reshape(t.shape[0], t.shape[1], 1, -1).contiguous()        # (B, T, N, c) -> one row of N * c per step
        xl = d02(self.embed_x(flat(X), 'linear'))                                     # masks drawn in the reference's order x, b, e, ae
        b = d02(self.embed_b(flat(B)))
        el = d02(self.embed_e(flat(E), 'linear'))
        ae = d02(self.embed_ae(flat(AE))) if self.act else None
        x_lin_last, e_lin_last = xl[:, -1:].contiguous(), el[:, -1:].contiguous()
        if drop:      # the activation follows the dropout (:199-201)
            x, e = _ag.apply_activation(xl, self.activation), _ag.apply_activation(el, self.activation)
        else:
            x, e = self.embed_x(flat(X), self.activation), self.embed_e(flat(E), self.activation)

        def spatial(layers, x, e):
            for ly in layers:
                z = ly(torch.cat([x, e], dim=-1))
                x, e = dr(z[..., :z.shape[-1] // 2].contiguous()), dr(z[..., z.shape[-1] // 2:].contiguous())
            return x, e

        def chain(mods, t):
            for ly in mods:
                t = ly(t)
            return t

        x, e = spatial(self.block1, x, e)
        x, e = chain(self.tem1_x, x)[:, -T:], chain(self.tem1_e, e)[:, -T:]
        x = torch.cat([x, b], dim=-1)                                                 # :260
        if self.act:
            e = torch.cat([e, ae], dim=-1)                                            # :262
        x, e = spatial(self.block2, x, e)
        x, e = chain(self.tem2_x, x), chain(self.tem2_e, e)

        def res_head(layer, t, lin_last):
            y = dr(layer(t))                                                          # :314,318
            if not self.resnet:
                return y
            if _ag.grad_on(y, lin_last):
                return _ag.CumsumActFn.apply(y, lin_last, self.activation)
            return _lib.cumsum_act(y.contiguous(), lin_last.contiguous(), self.activation)

        x, e = res_head(self.res_x, x, x_lin_last), res_head(self.res_e, e, e_lin_last)
        out = self.out(x).reshape(nb, T, self.n_node, self.n_out)                     # :322-325
        if self.if_flood:
            flood = self.flood_out(chain(self.flood, x)).reshape(nb, T, self.n_node, 1)
            out = torch.cat([out, flood], dim=-1)
        return out, self.e_out_layer(e).reshape(nb, T, self.n_edge, self.e_out)

    # ------------------------------------------------------------------ network forward (build_network)
    def forward(self, X, B, E, AE=None, ADJ=None, training=False):
        """ADJ: the per-time-step node adjacency of `use_adj` (emulator.py:178-180,268-271; block 2 only): the edge mask
        (B, T_out, nnz) of `get_adj_action`, or the reference's dense (B, T_out, n, n) integer array (small networks).
        training: keras' `training=` flag -- switches the Dropout layers on when the model was built with `dropout` > 0."""
        if not self.conv:
            return self._forward_mlp(X, B, E, AE, training)
        drop = bool(self.dropout) and training
        d02 = (lambda t: self._drop02(t, True)) if drop else (lambda t: t)
        dr = (lambda t: self._drop(t, True)) if drop else None
        nb = X.shape[0]
        c = lambda t: t.contiguous()
        adj_mask = None
        if ADJ is not None:
            if not (self.act and self.use_adj):
                raise ValueError('ADJ given but the model was built without act + use_adj')
            adj_mask = self._adj_mask_from(ADJ, X.device).reshape(nb * self.seq_out, -1)
        # the embedding is linear, its last step is kept as the residual, then the activation is applied (:198-201):
        # two launches of the same GEMM (same per-row arithmetic), one with and one without the activation
        if drop:      # Dense(linear) -> Dropout(0.2) -> residual slice -> activation (:198-201,206-209); masks drawn in the reference's order x, b, e, ae
            xl = d02(self.embed_x(c(X), 'linear'))
            b = d02(self.embed_b(c(B)))
            el = d02(self.embed_e(c(E), 'linear'))
            ae = d02(self.embed_ae(c(AE))) if self.act else None
            x_lin_last, e_lin_last = c(xl[:, -1:]), c(el[:, -1:])
            x, e = _ag.apply_activation(xl, self.activation), _ag.apply_activation(el, self.activation)
        else:
            x_lin_last = self.embed_x(c(X[:, -1:]), 'linear')
            x = self.embed_x(c(X), self.activation)
            e_lin_last = self.embed_e(c(E[:, -1:]), 'linear')
            e = self.embed_e(c(E), self.activation)
            b = self.embed_b(c(B))
            ae = self.embed_ae(c(AE)) if self.act else None

        def spatial(block, x, e, xb=None, eb=None, mask=None):
            T = x.shape[1]
            r = lambda t, n: None if t is None else t.reshape(nb * T, n, -1)
            kw = {} if mask is None else {'adj_mask': mask}
            if dr is not None:
                kw['dropout'] = dr
                if self.conv_kind == 'GAT':      # training=True reaches the GATConv layers too: Spektral's attention dropout (rate 0.5)
                    kw['attn_dropout'] = self.dropout_stream
            xs, es = block(r(x, self.n_node), r(e, self.n_edge), r(xb, self.n_node), r(eb, self.n_edge), **kw)
            return xs.reshape(nb, T, self.n_node, -1), es.reshape(nb, T, self.n_edge, -1)

        def temporal(mods_x, mods_e, x, e):
            """The two temporal stacks (:244-257): layer by layer; a pair of small Conv1D layers of one shape goes out as ONE launch."""
            if len(mods_x) != len(mods_e):
                for ly in mods_x:
                    x = ly(x)
                for ly in mods_e:
                    e = ly(e)
                return x, e
            for lx, le in zip(mods_x, mods_e):
                kx = lx.kernel.shape if isinstance(lx, Conv1D) else None
                if (kx is not None and isinstance(le, Conv1D) and le.kernel.shape == kx and lx.dilation_rate == le.dilation_rate
                        and lx.activation == le.activation and lx.precision == le.precision == 'bf16x3' and x.shape[-1] == e.shape[-1] == kx[1]
                        and _lib.rowgemm_supported(kx[0] * kx[1], kx[1], kx[2]) and x.shape[0] * x.shape[1] * max(x.shape[2], e.shape[2]) <= 16384
                        and kx[0] * kx[1] // 32 in (2, 6) and not _ag.grad_on(x, e, lx.kernel, le.kernel, lx.bias, le.bias)):
                    x, e = _lib.rowgemm_forward_pair(c(x), _packed_kernel(lx, lx.kernel.reshape(kx[0] * kx[1], kx[2])), lx.bias,
                                                     c(e), _packed_kernel(le, le.kernel.reshape(kx[0] * kx[1], kx[2])), le.bias, kx[2],
                                                     lx.activation, taps=kx[0], dilation=lx.dilation_rate)
                else:
                    x, e = lx(x), le(e)
            return x, e

        x, e = spatial(self.block1, x, e)
        x, e = temporal(self.tem1_x, self.tem1_e, x, e)
        x, e = x[:, -self.seq_out:], e[:, -self.seq_out:]             # :249,256
        # concat([x, b]) / concat([e, ae]) (:260-262) are not materialised: the first layer of block 2 reads both pieces
        x, e = spatial(self.block2, c(x), c(e), b, ae if self.act else None, adj_mask)
        x, e = temporal(self.tem2_x, self.tem2_e, x, e)
        def res_head(layer, t, lin_last):                             # :313-320
            if drop:
                y = dr(layer(t))
                return _ag.CumsumActFn.apply(y, lin_last, self.activation) if self.resnet else y
            if not self.resnet:
                return layer(t)
            if (layer.precision == 'bf16x3' and layer.units == 64 and t.shape[-1] == 64 and t.shape[0] * t.shape[2] >= 4096
                    and not _ag.grad_on(t, lin_last, layer.kernel, layer.bias)):
                # Dense + cumsum over time + residual + activation in one streaming kernel
                return _lib.dense_cumsum(c(t), _packed_kernel(layer, layer.kernel), layer.bias, c(lin_last), self.activation)
            y = layer(t)
            if _ag.grad_on(y, lin_last):
                return _ag.CumsumActFn.apply(y, lin_last, self.activation)
            return _lib.cumsum_act(y, lin_last, self.activation)

        def fused_heads(layer, t, lin_last, head, hidden, head_f):
            """Dense + cumsum + residual + activation with the output heads as the epilogue of the same streaming kernel (the 64-wide
            resnet output never reaches memory); None when the shape or mode is not the one that kernel takes."""
            pk = lambda m: _packed_kernel(m, m.kernel)
            ok = (self.resnet and layer.precision == 'bf16x3' and layer.units == 64 and t.shape[-1] == 64 and head.units <= 4
                  and len(hidden) <= 5 and all(m.units == 32 for m in hidden) and (not hidden or hidden[0].kernel.shape[0] == 64)
                  and not drop and not _ag.grad_on(t, lin_last, *self.parameters()))
            if not ok:
                return None
            return _lib.dense_cumsum_heads(
                c(t), pk(layer), layer.bias, c(lin_last), self.activation, (pk(head), head.bias, head.units, head.activation),
                [(pk(m), m.bias) for m in hidden],
                (pk(head_f), head_f.bias, head_f.activation, hidden[0].activation) if hidden else None)

        out = fused_heads(self.res_x, x, x_lin_last, self.out, list(self.flood), self.flood_out)
        e_out = fused_heads(self.res_e, e, e_lin_last, self.e_out_layer, [], None)
        if out is None:
            x = res_head(self.res_x, x, x_lin_last)
            out = self.out(x)
            if self.if_flood:
                f = x
                for ly in self.flood:
                    f = ly(f)
                out = torch.cat([out, self.flood_out(f)], dim=-1)     # :330-333
        if e_out is None:
            e_out = self.e_out_layer(res_head(self.res_e, e, e_lin_last))
        return out, e_out

    model = forward

    # ------------------------------------------------------------------ normalisation (:794-810)
    def set_norm(self, norm_x, norm_b, norm_y, norm_r, norm_e):
        for k, v in (('x', norm_x), ('b', norm_b), ('y', norm_y), ('r', norm_r), ('e', norm_e)):
            if v is not None:
                self._norms[k] = torch.as_tensor(np.asarray(v), dtype=torch.float32)

    def _norm(self, item, device):
        t = self._norms[item]
        if t.device != device:
            t = self._norms[item] = t.to(device)
        return t

    def _norm_parts(self, item, dim, device):
        """(maxi - mini, mini) of the first `dim` channels, computed once per norm tensor (the rollout calls normalize a dozen
        times per step on constants)."""
        normal = self._norm(item, device)
        key = (item, dim, str(device))
        hit = self._norm_derived.get(key)
        if hit is None or hit[0] is not normal:
            maxi, mini = normal[0, ..., :dim], normal[1, ..., :dim]
            hit = self._norm_derived[key] = (normal, (maxi - mini).contiguous(), mini.contiguous())
        return hit[1], hit[2]

    def normalize(self, dat, item, inverse=False):
        span, mini = self._norm_parts(item, dat.shape[-1], dat.device)
        return dat * span + mini if inverse else (dat - mini) / span

    # ------------------------------------------------------------------ actions (:364-398)
    def _act_edge_index(self):
        hits = [np.where((self.edges == ae).all(1))[0] for ae in self.act_edges]
        flat = [int(i) for e in hits for i in e]
        return sorted(set(flat), key=flat.index)

    def _dev_index(self, name, arr, device):
        """An int64 index list on the device, uploaded once."""
        key = (name, str(device))
        if key not in self._idx_cache:
            self._idx_cache[key] = torch.as_tensor(arr, dtype=torch.int64, device=device)
        return self._idx_cache[key]

    def get_edge_action(self, a, g=True):
        out = np.zeros(self.n_edge, dtype=np.int64)
        out[self._act_edge_index()] = np.arange(1, a.shape[-1] + 1)
        table = torch.cat([torch.ones_like(a[..., :1]), a], dim=-1)
        return table[..., self._dev_index('edge_action', out, a.device)].unsqueeze(-1)

    def _adj_pattern(self):
        """(CSR of the adjacency the node-side conv of block 2 runs on, raw values at its entries, entry position of every
        actuated (from, to) pair or -1).  Two-graph form: the node adjacency; graph_base: the combined node + link graph."""
        if getattr(self, '_adj_pat', None) is None:
            csr = self._base_filter if self.graph_base else self.graph.adj
            raw = np.ones(csr.nnz) if getattr(self, '_adj_raw', None) is None else self._adj_raw
            rp, col = np.asarray(csr.rowptr, dtype=np.int64), np.asarray(csr.col, dtype=np.int64)
            pos = np.full(len(self.act_edges), -1, dtype=np.int64)
            for k, (u, v) in enumerate(np.asarray(self.act_edges, dtype=np.int64)):
                hit = np.nonzero(col[rp[u]:rp[u + 1]] == v)[0]
                if len(hit):
                    pos[k] = rp[u] + hit[0]
            self._adj_pat = (csr, raw, pos)
        return self._adj_pat

    def get_adj_action(self, a, g=True):
        """`get_adj_action` (emulator.py:343-362) for conv=GAT, in CSR form: a (B, T, n_act) settings -> edge mask
        (B, T, nnz) over the entries of the node adjacency.  The reference writes `adj[from_k, to_k] = a_k` (g=False) or
        multiplies that entry by a_k (g=True) -- the (from, to) entry only, not its transpose -- and casts the matrix to int:
        an entry survives iff trunc(value) != 0, so with a 0/1 adjacency any setting below 1 removes it."""
        csr, raw, pos = self._adj_pattern()
        dev = a.device
        base = torch.as_tensor(np.trunc(raw) != 0, dtype=torch.float32, device=dev)
        mask = base.expand(tuple(a.shape[:-1]) + (csr.nnz,)).clone()
        ok = pos >= 0
        if ok.any():
            w = torch.as_tensor(raw[pos[ok]], dtype=a.dtype, device=dev) if g else torch.ones(int(ok.sum()), dtype=a.dtype, device=dev)
            val = a[..., torch.as_tensor(np.nonzero(ok)[0], device=dev)] * w
            mask[..., torch.as_tensor(pos[ok], device=dev)] = (torch.trunc(val) != 0).to(torch.float32)
        return mask

    def _adj_mask_from(self, adj, device):
        """Edge mask from what `forward` is given: the mask itself (..., nnz) or the reference's dense (..., n, n) adjacency."""
        csr, _, _ = self._adj_pattern()
        adj = torch.as_tensor(adj) if not isinstance(adj, torch.Tensor) else adj
        n = csr.n_rows
        if adj.dim() >= 2 and tuple(adj.shape[-2:]) == (n, n) and adj.shape[-1] != csr.nnz:
            rows = torch.as_tensor(np.repeat(np.arange(n), np.diff(csr.rowptr)), device=adj.device)
            cols = torch.as_tensor(np.asarray(csr.col, dtype=np.int64), device=adj.device)
            adj = (torch.trunc(adj[..., rows, cols].to(torch.float32)) != 0)
        elif adj.shape[-1] != csr.nnz:
            raise ValueError('ADJ %r is neither an edge mask (..., %d) nor a dense (..., %d, %d) adjacency' % (tuple(adj.shape), csr.nnz, n, n))
        return adj.to(device=device, dtype=torch.float32)

    def get_action(self, a, g=True):
        out_o, out_i = np.zeros(self.n_node, dtype=np.int64), np.zeros(self.n_node, dtype=np.int64)
        out_o[self.act_edges[:, 0]] = np.arange(1, a.shape[-1] + 1)
        out_i[self.act_edges[:, 1]] = np.arange(1, a.shape[-1] + 1)
        table = torch.cat([torch.ones_like(a[..., :1]), a], dim=-1)
        return table[..., self._dev_index('act_out', out_o, a.device)], table[..., self._dev_index('act_in', out_i, a.device)]

    # ------------------------------------------------------------------ post-processing (:680-770)
    def _from_node_index(self, device):
        """(idx (E,) int64, ok (E,) float): the node whose incidence entry of link e is positive -- its from-node -- read off the
        signed incidence CSR (`get_node_edge`, base.py:432-439: +1 at the from-node, then -1 at the to-node; a link from a node to
        itself therefore has no positive entry: ok = 0, as in the dense `clip(node_edge, 0, 1)`)."""
        key = ('from_node', str(device))
        if key not in self._idx_cache:
            ie = self.graph.inc_e
            rows, cols, vals = ie.rows(), np.asarray(ie.col, dtype=np.int64), np.asarray(ie.val)
            idx, ok = np.zeros(self.n_edge, dtype=np.int64), np.zeros(self.n_edge, dtype=np.float32)
            pos = vals > 0
            idx[rows[pos]], ok[rows[pos]] = cols[pos], 1.0
            self._idx_cache[key] = (torch.as_tensor(idx, device=device), torch.as_tensor(ok, device=device))
        return self._idx_cache[key]

    def _at_from_node(self, v):
        """v (..., N) -> (..., E): v @ clip(node_edge, 0, 1) without the dense matrix."""
        idx, ok = self._from_node_index(v.device)
        return v[..., idx] * ok

    def _flow_balance(self, flow):
        """q_in, q_out (B,T,N,1) from de-normalised link flows (B,T,E,1): HIP kernel on the incidence CSR."""
        if self._inc_handle is None:
            self._inc_handle = _lib.CsrHandle(self.graph.inc_n)
        ny = self._norm('y', flow.device)
        hit = self._norm_derived.get(('flow_scale', str(flow.device)))
        if hit is None or hit[0] is not ny:
            hit = self._norm_derived[('flow_scale', str(flow.device))] = (
                ny, ((ny[0, :, 1] > 1e-3).float() / ny[0, :, 1]).contiguous(), ((ny[0, :, 2] > 1e-3).float() / ny[0, :, 2]).contiguous())
        s_in, s_out = hit[1], hit[2]
        lead = flow.shape[:-2]
        if _ag.grad_on(flow):
            edges = torch.as_tensor(self.edges, dtype=torch.int64, device=flow.device)
            q_in, q_out = _ag.FlowBalanceFn.apply(flow.reshape(-1, self.n_edge), self._inc_handle, self._inc_sign, s_in, s_out, edges)
        else:
            q_in, q_out = _lib.flow_balance(self._inc_handle, self._inc_sign, flow.reshape(-1, self.n_edge).contiguous(), s_in, s_out)
        return q_in.reshape(lead + (self.n_node, 1)), q_out.reshape(lead + (self.n_node, 1))

    def post_proc_tf(self, preds, a, b):
        """`post_proc_tf` (emulator.py:680-725): the graph-mode post-processing used by `predict_tf`, `_model`, MPC and MBRL."""
        return self._post_proc(preds, a, b, False)

    def post_proc(self, y, ey, a, b):
        """`post_proc` (emulator.py:643-678): the NumPy twin used by `predict` and `simulate`.  NOT equivalent to
        `post_proc_tf` where pumps are involved: the rated-pump override of the link flow is applied whenever the model has
        actions (the graph-mode form only when EVERY link is a pump, `self.pump.min() > 0`, :698), and the node pumps of the
        non-edge-fusion form open at depth > 0.01 instead of > 0 (:664-665 vs :710-711).  The offset gate is written without
        its `offset.max() > 0` guard (:649-653 vs :688), which changes nothing: with no offsets the expression is the flow."""
        return self._post_proc((y, ey), a, b, True)

    def _post_proc(self, preds, a, b, np_form):
        preds, edge_preds = preds
        pump_override = self._has_link_pump if np_form else self._has_pump      # :656-659 (always with actions) vs :698 (`pump.min() > 0`)
        depth_gate = 0.01 if np_form else 0.0                                      # :664-665 vs :710-711
        # `v @ clip(node_edge, 0, 1)` (:649,657,689,699): node_edge has one +1 per link, at its from-node (base.py:432-439), so the
        # dense (N, E) product is a gather of v at the from-nodes -- the same numbers from the incidence CSR, for any N
        if self.tide:
            h = preds[..., 0] * (1 - self.is_outfall) + b[..., -1]
            preds = torch.cat([h.unsqueeze(-1), preds[..., 1:]], dim=-1)
        if self._has_offset:
            inoff = self._at_from_node(self.normalize(preds, 'y', True)[..., 0] - self.hmin)
            flow, off = edge_preds[..., -1], self.offset
            flow = (flow * (flow > 0).float() * (off > 0).float() * (inoff > off).float() + flow * (flow <= 0).float() * (off > 0).float() +
                    flow * (off == 0).float()).unsqueeze(-1)
            edge_preds = torch.cat([edge_preds[..., :-1], flow], dim=-1)
        if self.act:
            ne_ = self._norm('e', preds.device)
            if pump_override:
                fl = self.pump * self._at_from_node((preds[..., 0] > 0.01).float())
                fl = fl * (ne_[0, :, 2] > 1e-3).float() / ne_[0, :, 2]
                flow = (edge_preds[..., -1] * (fl == 0).float() + fl).unsqueeze(-1)
            else:
                flow = edge_preds[..., -1:]
            edge_preds = torch.cat([edge_preds[..., :-1], flow * self.get_edge_action(a, True)], dim=-1)
            if not self.edge_fusion:
                ny = self._norm('y', preds.device)
                a_out, a_in = self.get_action(a[:, :self.seq_out], True)
                fli = self.pump_in * (preds[..., 0] > depth_gate).float() / ny[0, :, 1]
                flo = self.pump_out * (preds[..., 0] > depth_gate).float() / ny[0, :, 2]
                inflow = preds[..., 1] * (fli == 0).float() + fli
                outflow = preds[..., 2] * (flo == 0).float() + flo
                preds = torch.cat([torch.stack([preds[..., 0], inflow * a_in, outflow * a_out], dim=-1), preds[..., 3:]], dim=-1)
        if self.edge_fusion:
            span, mini = self._norm_parts('e', edge_preds.shape[-1], edge_preds.device)
            flow = edge_preds[..., -1:] * span[..., -1:] + mini[..., -1:]        # de-normalised flow channel only
            q_in, q_out = self._flow_balance(flow)
            preds = torch.cat([preds[..., :1], q_in, q_out, preds[..., 1:]], dim=-1)
        return preds, edge_preds

    def constrain_tf(self, y, r, h0=None):
        h, q_us, q_ds = y[..., 0], y[..., 1], y[..., 2]
        r = r.squeeze(-1)
        h = torch.minimum(torch.maximum(h, self.hmin), self.hmax)
        q_w = (q_us + r - q_ds).clamp(min=0) * (1 - self.is_outfall)
        if self.if_flood:
            f = (y[..., -1] > 0.5).float()
            h = self.hmax * f + h * (1 - f)
            y = torch.stack([h, q_us, q_ds, y[..., -1]], dim=-1)
        else:
            y = torch.stack([h, q_us, q_ds], dim=-1)
        if self.epsilon > 0:
            q_w = q_w * ((self.hmax - h) < self.epsilon).float()
        elif self.epsilon == 0:
            pass
        elif self.if_flood:
            q_w = q_w * f
        return q_w, y

    # ------------------------------------------------------------------ inference entry points
    def predict_tf(self, states, b, a=None, edge_state=None):
        """emulator.py:604-641 (graph mode: `post_proc_tf`)."""
        return self._predict(states, b, a, edge_state, False)

    def predict(self, states, b, a=None, edge_state=None):
        """emulator.py:566-602: the NumPy-mode entry point (`post_proc`, see there for how it differs from `predict_tf`);
        tensors stay on the device."""
        return self._predict(states, b, a, edge_state, True)

    def _predict(self, states, b, a, edge_state, np_form):
        x = states[:, -self.seq_in:]
        ex = edge_state[:, -self.seq_in:]
        assert b.shape[1] == self.seq_out
        # NumPy mode writes the setting INTO the adjacency entry (g=False, :572-574), graph mode multiplies the entry by it
        # (g=True, :611-613): the two differ on a weighted adjacency (length > 0), where entry * setting truncates to 0
        ae = self.get_edge_action(a, not np_form) if self.act else None
        adj = self.get_adj_action(a, not np_form) if self.act and self.use_adj else None
        nb_ = self.normalize(b, 'b')
        y, ey = self.forward(self.normalize(x, 'x'), nb_, self.normalize(ex, 'e'), ae, adj)
        y, ey = self._post_proc((y, ey), a, nb_, np_form)
        ey = self.normalize(ey, 'e', True)
        ey = torch.cat([torch.minimum(ey[..., 0].clamp(min=0), self.ehmax).unsqueeze(-1), ey[..., 1:]], dim=-1)
        y = self.normalize(y, 'y', True)
        if self._has_any_pump:       # pumped-storage depth (:630-638)
            idx, ok = self._from_node_index(y.device)          # clip(node_edge, 0, 1) @ pump: the pumps leaving every node
            ps = ((self.area * torch.zeros_like(self.area).index_add_(0, idx, self.pump * ok)) > 0).float()
            h, qin, qout = y[..., 0], y[..., 1], y[..., 2]
            de = []
            for t in range(self.seq_out):
                prev = x[:, -1, :, 0] if t == 0 else de[-1] + (qin - qout)[:, t] / (self.area + 1e-6)
                de.append(torch.minimum(torch.maximum(prev, self.hmin), self.hmax))
            y = torch.cat([(h * (1 - ps) + torch.stack(de, dim=1) * ps).unsqueeze(-1), y[..., 1:]], dim=-1)
        q_w, y = self.constrain_tf(y, b[..., :1], x[:, -1:, :, 0])
        return torch.cat([y, q_w.unsqueeze(-1)], dim=-1), ey

    def _model(self, x, a, b, ex, ae=None, adj=None, fit=False):
        """emulator.py:400-438 on normalised tensors; `roll` > 0 = autoregressive chunks of seq_out steps."""
        if self.roll:
            ys, eys = [], []
            for i in range(self.roll):
                sl = slice(i * self.seq_out, (i + 1) * self.seq_out)
                y, ey, x, ex = self._roll_step(x, ex, a[:, sl] if a is not None else None, b[:, sl], fit)
                ys.append(y)
                eys.append(ey)
            preds, edge_preds = torch.cat(ys, dim=1), torch.cat(eys, dim=1)
        else:
            if ae is None and self.act:
                ae = self.get_edge_action(a, True)
            if adj is None and self.act and self.use_adj:
                adj = self.get_adj_action(a, True)
            preds, edge_preds = self.post_proc_tf(self.forward(x, b, ex, ae, adj, training=fit), a, b)
        return preds.clamp(0, 1), edge_preds                          # :437

    def _roll_step(self, x, ex, a_i, b_i, fit=False):
        """One chunk of the autoregressive rollout (emulator.py:403-423): forward on the last seq_in steps, post-processing,
        then the window shifts by seq_out steps fed with the prediction (flood bit thresholded at 0.5)."""
        ae_i = self.get_edge_action(a_i, True) if self.act else None
        adj_i = self.get_adj_action(a_i, True) if self.act and self.use_adj else None       # :407
        y, ey = self.forward(x[:, -self.seq_in:], b_i, ex[:, -self.seq_in:], ae_i, adj_i, training=fit)
        y, ey = self.post_proc_tf((y, ey), a_i, b_i)
        if self.if_flood:                                   # flood bit fed back as a hard 0/1 (:417)
            x_new = torch.cat([y[..., :-1], (y[..., -1:] > 0.5).float(), b_i], dim=-1)
        else:
            x_new = torch.cat([y, b_i], dim=-1)
        keep = self.seq_in - self.seq_out
        x = torch.cat([x[:, -keep:], x_new], dim=1) if keep > 0 else x_new
        ae_new = ae_i if self.act else torch.ones(ey.shape[:-1] + (1,), device=ey.device)
        ex_new = torch.cat([ey, ae_new], dim=-1)
        ex = torch.cat([ex[:, -keep:], ex_new], dim=1) if keep > 0 else ex_new
        return y, ey, x, ex

    def _fused_roll_ok(self, x, b, ex):
        """The plain configuration whose post-forward part `uds_roll_update` covers: edge fusion, no control actions, no tide,
        no offset / pump gating, one runoff channel, state = [h, q_in, q_out, (flood)] + runoff, link state = 3 + setting."""
        cy = 1 + int(bool(self.if_flood))
        return (self.edge_fusion and not self.act and not self.tide and not self._has_offset and b.shape[-1] == 1 and
                x.shape[-1] == cy + 3 and ex.shape[-1] == 4 and self.seq_out <= self.seq_in)

    def _roll_step_fused(self, x_win, ex_win, b_i):
        """`_roll_step` on persistent windows: forward, then ONE kernel pair for post-processing + feedback (in place)."""
        if self._inc_handle is None:
            self._inc_handle = _lib.CsrHandle(self.graph.inc_n)
        y, ey = self.forward(x_win, b_i, ex_win, None)
        span, mini = self._norm_parts('e', ey.shape[-1], ey.device)
        dev = str(ey.device)
        hit = self._norm_derived.get(('roll_e', dev))
        if hit is None or hit[0] is not span:
            hit = self._norm_derived[('roll_e', dev)] = (span, span[..., -1].contiguous(), mini[..., -1].contiguous())
        ny = self._norm('y', ey.device)
        fs = self._norm_derived.get(('flow_scale', dev))
        if fs is None or fs[0] is not ny:
            fs = self._norm_derived[('flow_scale', dev)] = (
                ny, ((ny[0, :, 1] > 1e-3).float() / ny[0, :, 1]).contiguous(), ((ny[0, :, 2] > 1e-3).float() / ny[0, :, 2]).contiguous())
        preds = _lib.roll_update(self._inc_handle, self._inc_sign, hit[1], hit[2], fs[1], fs[2], y, ey, b_i, x_win, ex_win, self.if_flood)
        return preds, ey

    def rollout_graphed(self, x, a, b, ex):
        """`_model` with roll > 0 where every chunk replays ONE captured HIP graph (the step has ~60 small launches: with few
        snapshots per step the eager loop is bound by launch latency).  Same arguments and result as `_model(x, a, b, ex)`;
        the graph is re-captured when shapes change.  Parameters must not change between capture and replay (the packed
        weight buffers are baked in): call `drop_graph()` after loading or training."""
        if not self.roll:
            raise ValueError('rollout_graphed needs roll > 0')
        so = self.seq_out
        key = (tuple(x[:, -self.seq_in:].shape), tuple(b.shape[2:]), tuple(ex[:, -self.seq_in:].shape), None if a is None else tuple(a.shape[2:]),
               str(x.device))
        if self._graph is None or self._graph['key'] != key:
            G = dict(key=key, x=x[:, -self.seq_in:].clone(), ex=ex[:, -self.seq_in:].clone(), b=b[:, :so].clone(),
                     a=None if a is None else a[:, :so].clone())
            side = torch.cuda.Stream(device=x.device)
            side.wait_stream(torch.cuda.current_stream(x.device))
            fused = self._fused_roll_ok(G['x'], G['b'], G['ex'])     # post-processing + feedback as one kernel pair, in place
            with torch.cuda.stream(side), torch.no_grad():          # warm-up: caches, packed weights, tile plans
                for _ in range(2):
                    if fused:
                        self._roll_step_fused(G['x'].clone(), G['ex'].clone(), G['b'])
                    else:
                        self._roll_step(G['x'], G['ex'], G['a'], G['b'])
            torch.cuda.current_stream(x.device).wait_stream(side)
            G['graph'] = torch.cuda.CUDAGraph()
            with torch.cuda.graph(G['graph']), torch.no_grad():
                if fused:
                    y, ey = self._roll_step_fused(G['x'], G['ex'], G['b'])
                else:
                    y, ey, xn, exn = self._roll_step(G['x'], G['ex'], G['a'], G['b'])
                    G['x'].copy_(xn)
                    G['ex'].copy_(exn)
            G['y'], G['ey'] = y, ey
            self._graph = G
        G = self._graph
        with torch.no_grad():
            G['x'].copy_(x[:, -self.seq_in:])
            G['ex'].copy_(ex[:, -self.seq_in:])
            ys = torch.empty((x.shape[0], self.roll * so) + tuple(G['y'].shape[2:]), device=x.device)
            eys = torch.empty((x.shape[0], self.roll * so) + tuple(G['ey'].shape[2:]), device=x.device)
            for i in range(self.roll):
                sl = slice(i * so, (i + 1) * so)
                G['b'].copy_(b[:, sl])
                if a is not None:
                    G['a'].copy_(a[:, sl])
                G['graph'].replay()
                ys[:, sl].copy_(G['y'])
                eys[:, sl].copy_(G['ey'])
        return ys.clamp(0, 1), eys

    def drop_graph(self):
        self._graph = None

    def simulate(self, states, runoff, a=None, edge_states=None):
        """emulator.py:521-564, with every sliding window of the event batched into ONE forward (the reference loops
        over time steps with a host round trip each).  states (n,T_in,N,C), runoff (n,T_out,N,b_in) -> (n,T_out,N,5), (n,T_out,E,3)."""
        runoff = runoff[:, :self.seq_out]
        return self.predict(states, runoff, a, edge_states)      # (the reference's loop body is `predict` on one window: NumPy-mode `post_proc`)

    # ------------------------------------------------------------------ training step (:440-484)
    def _loss_setup(self, device):
        """nwei / ewei / poswei exactly as the constructor of the reference builds them (emulator.py:82,99-106,116)."""
        if getattr(self, '_lw', None) is not None and self._lw[0] == device:
            return self._lw[1]
        g = lambda k, d: np.asarray(getattr(self._args, k, d), dtype=np.float64)
        nwei = np.repeat(g('nwei', np.ones(self.n_node))[:, None], 3 + int(self.balance), axis=-1).astype(np.float32)
        hmax, hmin, outf = (self.hmax.cpu().numpy().astype(np.float64), self.hmin.cpu().numpy().astype(np.float64),
                            self.is_outfall.cpu().numpy().astype(np.float64))
        if hmin.max() > 0:
            wei = (hmax - hmin) * (1 - outf) + (hmax - hmin).mean() * outf
            wei = (hmax.max() - hmin.min()) / wei
            nwei = nwei * np.stack([wei] + (2 + int(self.balance)) * [np.ones_like(wei)], axis=-1)
        t = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float32), device=device)
        lw = dict(nwei=t(nwei), ewei=t(g('ewei', np.ones(self.n_edge))), poswei=t(g('poswei', np.ones(self.n_node))))
        self._lw = (device, lw)
        return lw

    @staticmethod
    def _mse(y_true, y_pred, sample_weight=None):
        """keras MeanSquaredError(): mean over the last axis, optional per-sample weights, then the mean over all samples."""
        per = ((y_pred - y_true) ** 2).mean(dim=-1)
        if sample_weight is not None:
            per = per * sample_weight
        return per.mean()

    @staticmethod
    def _bce(y_true, y_pred, sample_weight):
        """keras BinaryCrossentropy() on probabilities (clipped to [1e-7, 1 - 1e-7]), weighted, mean over all samples."""
        p = y_pred.clamp(1e-7, 1 - 1e-7)
        per = -(y_true * torch.log(p) + (1 - y_true) * torch.log(1 - p)).mean(dim=-1)
        return (per * sample_weight).mean()

    def get_node_loss(self, y, b, preds):
        lw = self._loss_setup(preds.device)
        if self.balance:
            q_w, pr = self.constrain_tf(self.normalize(preds, 'y', True), self.normalize(b, 'b', True)[..., :1])
            q_w = (q_w / self._norm('y', preds.device)[0, :, -1]).unsqueeze(-1)
            pr = self.normalize(pr, 'y').clamp(0, 1)
            return self._mse(torch.cat([y[..., :3], y[..., -1:]], dim=-1) * lw['nwei'], torch.cat([pr[..., :3], q_w], dim=-1) * lw['nwei'])
        return self._mse(y[..., :3] * lw['nwei'], preds[..., :3] * lw['nwei'])

    def get_flood_loss(self, y, preds):
        lw = self._loss_setup(preds.device)
        weight = lw['poswei'] * y[..., -2] + lw['nwei'][:, -1] * (1 - y[..., -2])
        return self._bce(y[..., -2:-1], preds[..., -1:], weight)

    def fit_eval(self, x, a, b, y, ex, ey, fit=True):
        """One training (fit=True) or evaluation step on NORMALISED tensors (emulator.py:457-484): forward through
        `_model`, node MSE (+ weighted flood BCE) + link MSE, reverse mode through the HIP operators (autograd.py), Adam with
        per-variable clipnorm=1.  Returns [node_loss, (flood_loss,) edge_loss] as 0-d tensors."""
        params = [p for p in self.parameters()]
        if fit:
            for p in params:
                p.requires_grad_(True)
        with torch.set_grad_enabled(bool(fit)):
            ae = self.get_edge_action(a, True) if self.act else None
            preds, edge_preds = self._model(x, a, b, ex, ae, None, fit)
            lw = self._loss_setup(preds.device)
            node_loss = self.get_node_loss(y, b, preds)
            fl_loss = self.get_flood_loss(y, preds) if self.if_flood and not self.balance else None
            edge_loss = self._mse(ey, edge_preds, lw['ewei'])
            if fit:
                loss = node_loss + edge_loss
                if self.gradnorm:                    # GradNorm task weights (emulator.py:470-473): constants for this step
                    alpha = self._alphas(preds.device)
                    loss = alpha[0].detach() * loss
                if self.if_flood:
                    if fl_loss is None:
                        raise NotImplementedError('if_flood with balance: the reference leaves fl_loss undefined here (emulator.py:466,473)')
                    loss = loss + (alpha[1].detach() * fl_loss if self.gradnorm else fl_loss)
                from .dist import all_ranks_finite, allreduce_gradients
                if not all_ranks_finite(loss):       # agreed over the data-parallel ranks: all raise or none does
                    raise FloatingPointError('Loss contains NaN or Inf values.')
                for p in params:
                    p.grad = None
                loss.backward()
                allreduce_gradients(params)          # data-parallel ranks: one bucketed all-reduce (no-op on one rank)
                if self._optimizer is None:
                    self._optimizer = KerasAdam(params, self.learning_rate, clipnorm=1.0)
                self._optimizer.step()
        out = [node_loss.detach()] + ([fl_loss.detach()] if self.if_flood and fl_loss is not None else []) + [edge_loss.detach()]
        return out

    # ------------------------------------------------------------------ GradNorm (:118-125,486-519)
    def _alphas(self, device):
        """[alpha_reg, alpha_cls]: the two trainable task weights (initially 1, kept at sum 2), with their own Adam(1e-4)."""
        if getattr(self, '_alpha', None) is None or self._alpha.device != device:
            self._alpha = torch.ones(2, device=device, dtype=torch.float32, requires_grad=True)
            self._alpha_optimizer = KerasAdam([self._alpha], 1e-4)
        return self._alpha

    def fit_grad_norm(self, x, a, b, y, ex, ey, ini_loss):
        """One GradNorm update of the task weights (`fit_grad_norm` / `_get_grad_norm`, emulator.py:486-519): the norms of
        alpha_task * d loss_task / d W at the shared layer W = `dense_resx` kernel are pulled (mean absolute error) towards
        mean(norms) * (relative inverse training rate) ** 0.5; one Adam(1e-4) step on the alphas, then they are rescaled to
        sum 2.  ini_loss = [node, flood, edge] losses of the first step.  Returns the alpha loss (0-d tensor)."""
        if not self.gradnorm:
            raise ValueError('fit_grad_norm needs args.gradnorm = True')
        if not self.if_flood:
            raise ValueError('GradNorm balances the regression and the flood-classification task: it needs if_flood')
        W = self.res_x.kernel
        flags = [(p, p.requires_grad) for p in self.parameters()]
        for p, _ in flags:
            p.requires_grad_(p is W)
        try:
            with torch.enable_grad():
                ae = self.get_edge_action(a, True) if self.act else None
                preds, edge_preds = self._model(x, a, b, ex, ae, None)
                lw = self._loss_setup(preds.device)
                reg_loss = self.get_node_loss(y, b, preds) + self._mse(ey, edge_preds, lw['ewei'])
                fl_loss = self.get_flood_loss(y, preds)
                g_reg, = torch.autograd.grad(reg_loss, W, retain_graph=True)
                g_cls, = torch.autograd.grad(fl_loss, W)
                alpha = self._alphas(preds.device)
                norms = torch.stack([(alpha[0] * g_reg.detach()).norm(), (alpha[1] * g_cls.detach()).norm()])
                ini = [float(v) for v in ini_loss]
                r = torch.stack([reg_loss.detach() / (ini[0] + ini[-1]), fl_loss.detach() / ini[1]])
                target = norms.detach().mean() * (r / r.mean()) ** 0.5
                alpha_loss = (target - norms).abs().mean()
                alpha.grad = None
                alpha_loss.backward()
            self._alpha_optimizer.step()
            with torch.no_grad():
                alpha.mul_(2.0 / alpha.sum())
        finally:
            for p, f in flags:
                p.requires_grad_(f)
        return alpha_loss.detach()

    # ------------------------------------------------------------------ Keras checkpoints (:814-852)
    def keras_layer_map(self):
        """[(keras layer name, module, [(keras weight name, parameter name), ...])] in the creation order of
        `build_network` (emulator.py:166-341): the auto-numbered names `save_weights('model.h5')` stores the weights
        under (`dense`, `dense_1`, ..., `node_edge_k`, `mixed_gat_k` / `gcn_conv_k`, `conv1d_k`, `dense_resx`)."""
        count = {}

        def name(base):
            k = count.get(base, 0)
            count[base] = k + 1
            return base if k == 0 else '%s_%d' % (base, k)

        dense_w = [('kernel', 'kernel'), ('bias', 'bias')]
        ne_w = [('weight', 'weight'), ('bias', 'bias')]
        gat_w = [('kernel', 'kernel'), ('attn_kernel_self', 'attn_kernel_self'), ('attn_kernel_neigh', 'attn_kernel_neighs'), ('bias', 'bias')]
        out = [(name('dense'), self.embed_x, dense_w), (name('dense'), self.embed_b, dense_w), (name('dense'), self.embed_e, dense_w)]
        if self.act:
            out.append((name('dense'), self.embed_ae, dense_w))

        # DiffusionConv keeps its coefficients in `channels` DiffuseFeatures sub-layers (one (K + 1,) kernel each): the importer
        # takes them stacked in creation order as ONE (channels, K + 1) array under '<layer>/kernel:0'
        conv_name = {'GAT': 'mixed_gat', 'GCN': 'gcn_conv', 'Diffusion': 'diffusion_conv'}[self.conv_kind]
        conv_w = {'GAT': gat_w, 'GCN': dense_w, 'Diffusion': [('kernel', 'kernel')]}[self.conv_kind]

        def spatial(block):
            if not self.conv:
                for ly in block:
                    out.append((name('dense'), ly, dense_w))
                return
            if self.graph_base:
                for ly in block.layers:
                    out.append((name(conv_name), ly, conv_w))
                return
            for ly in block.layers:
                out.append((name('dense'), ly.dense_xe, dense_w))
                out.append((name('dense'), ly.dense_ex, dense_w))
                out.append((name('node_edge'), ly.node_edge_n, ne_w))
                out.append((name('node_edge'), ly.node_edge_e, ne_w))
                if ly.conv == 'GAT':
                    out.append((name('mixed_gat'), ly.gat_x, gat_w))
                    out.append((name('mixed_gat'), ly.gat_e, gat_w))
                else:
                    out.append((name(conv_name), ly.gcn_x, conv_w))
                    out.append((name(conv_name), ly.gcn_e, conv_w))

        def temporal(mods):
            for m in mods:
                if isinstance(m, _Recurrent):       # keras nests the weights in the layer's cell: gru/gru_cell/kernel:0, ...
                    # the cell objects are auto-numbered by a counter of their own: the k-th GRU layer saves
                    # 'gru_k/gru_k/gru_cell_k/kernel:0' (Keras 2.10 creates the cell without a name)
                    cell = name(m.KIND.lower() + '_cell')
                    out.append((name(m.KIND.lower()), m, [(cell + '/kernel', 'kernel'), (cell + '/recurrent_kernel', 'recurrent_kernel'),
                                                         (cell + '/bias', 'bias')]))
                else:
                    out.append((name('conv1d'), m, dense_w))

        spatial(self.block1)
        temporal(self.tem1_x)
        temporal(self.tem1_e)
        spatial(self.block2)
        temporal(self.tem2_x)
        temporal(self.tem2_e)
        out.append(('dense_resx', self.res_x, dense_w))
        out.append((name('dense'), self.res_e, dense_w))
        out.append((name('dense'), self.out, dense_w))
        for m in self.flood:
            out.append((name('dense'), m, dense_w))
        if self.if_flood:
            out.append((name('dense'), self.flood_out, dense_w))
        out.append((name('dense'), self.e_out_layer, dense_w))
        return out

    def load_keras_weights(self, weights):
        """Weights of a trained reference model, keyed the way Keras stores them in `model.h5`:
        `weights['<layer>/<weight>:0']` or `weights['<layer>/<layer>/<weight>:0']` (the HDF5 group path) or
        `weights['<layer>'] = [arrays in the layer's own order]`, values array-like.  `Emulator.load('model.h5')` reads the file
        with the minimal HDF5 reader of `gnn_uds_amd/h5.py` and calls this; arrays dumped elsewhere with h5py work the same.  Shapes are
        checked; NodeEdge parameters created with sparse=True take the dense (R, M) arrays on their support and refuse a bias
        that is non-zero off it (ValueError: nothing is dropped silently)."""
        weights = dict(weights)
        missing = []
        with torch.no_grad():
            for lname, mod, pairs in self.keras_layer_map():
                for idx, (kname, pname) in enumerate(pairs):
                    arr = _find_keras_weight(weights, lname, kname)
                    if arr is None and lname in weights and not hasattr(weights[lname], 'shape'):
                        arr = weights[lname][idx]
                    if arr is None:
                        missing.append('%s/%s' % (lname, kname))
                        continue
                    p = getattr(mod, pname)
                    t = torch.as_tensor(np.asarray(arr), dtype=torch.float32)
                    if getattr(mod, 'sparse', False) and t.dim() == 2:        # NodeEdge with one parameter per support entry
                        flat = mod._flat.cpu()
                        if pname == 'bias':
                            # the reference trains `b` as a full (R, M) matrix (emulator.py:36-45): entries off the incidence
                            # support act on the output (`+ b @ x`) and a sparse=True module has nowhere to keep them
                            off = t.clone()
                            off.reshape(-1)[flat] = 0.0
                            if bool((off != 0).any()):
                                raise ValueError('%s/%s: the checkpoint bias has %d non-zero entries off the incidence support (max |b| = %.3g); '
                                                 'a model built with args.sparse_params=True would drop them and change the outputs -- build it '
                                                 'with args.sparse_params=False (the fused kernel takes the dense remainder)'
                                                 % (lname, kname, int((off != 0).sum()), float(off.abs().max())))
                        t = t.reshape(-1)[flat]
                    if pname == 'kernel' and t.dim() == 2 and p.dim() == 3:   # GCNConv (F, C) vs GATConv (F, 1, C)
                        t = t.reshape(p.shape)
                    if tuple(t.shape) != tuple(p.shape):
                        raise ValueError('%s/%s: checkpoint shape %r, model expects %r' % (lname, kname, tuple(t.shape), tuple(p.shape)))
                    p.copy_(t.to(p.device))
        if missing:
            raise KeyError('weights not found in the checkpoint: %s' % ', '.join(missing))
        return self

    def export_keras_weights(self):
        """{'<layer>/<weight>:0': ndarray} under the Keras names (inverse of load_keras_weights; dense NodeEdge only)."""
        out = {}
        for lname, mod, pairs in self.keras_layer_map():
            if getattr(mod, 'sparse', False):
                raise NotImplementedError('NodeEdge(sparse=True) has no dense (R, M) parameters to export')
            for kname, pname in pairs:
                out['%s/%s:0' % (lname, kname)] = getattr(mod, pname).detach().cpu().numpy().copy()
        return out

    # ------------------------------------------------------------------ checkpoints (:814-852)
    def save(self, model_dir=None):
        """emulator.py:814-831: weights, the five normalisers, the optimizer state (`optim.npy` there, `optim.pt` here) and,
        with GradNorm, the task weights and THEIR optimizer (`gradnorm.ckpt` there, `gradnorm.pt` here).  A path ending in
        `.pt` names the weight file itself, as `.h5` does in the reference."""
        model_dir = model_dir if model_dir is not None else self.model_dir
        if model_dir.endswith('.pt'):
            weights, model_dir = model_dir, os.path.dirname(model_dir)
        else:
            weights = os.path.join(model_dir, 'model.pt')
        os.makedirs(model_dir or '.', exist_ok=True)
        torch.save(self.state_dict(), weights)
        for item, t in self._norms.items():
            np.save(os.path.join(model_dir, 'norm_%s.npy' % item), t.cpu().numpy())
        if self._optimizer is not None:
            torch.save(self._optimizer.state_dict(), os.path.join(model_dir, 'optim.pt'))
        if self.gradnorm and getattr(self, '_alpha', None) is not None:
            torch.save({'alpha': self._alpha.detach().cpu(), 'optimizer': self._alpha_optimizer.state_dict()}, os.path.join(model_dir, 'gradnorm.pt'))

    def load(self, model_dir=None, retrain=False):
        """emulator.py:833-852; `retrain=True` also restores the optimizer moments / step count and the GradNorm state, so
        that training resumes where it stopped (`--load_model`, main.py:199-205)."""
        model_dir = model_dir if model_dir is not None else self.model_dir
        keras = None
        if model_dir.endswith('.h5'):                     # the reference's own layout: a Keras weight file (:835-838)
            keras, model_dir = model_dir, os.path.dirname(model_dir)
        elif model_dir.endswith('.pt'):
            weights, model_dir = model_dir, os.path.dirname(model_dir)
        else:
            weights = os.path.join(model_dir, 'model.pt')
            if not os.path.exists(weights) and os.path.exists(os.path.join(model_dir, 'model.h5')):
                keras = os.path.join(model_dir, 'model.h5')   # a model directory written by the reference
        if keras is not None:
            from .h5 import read_keras_weights            # minimal HDF5 reader (no h5py in this image): see h5.py for what it covers
            self.load_keras_weights(read_keras_weights(keras))
        else:
            self.load_state_dict(torch.load(weights, weights_only=True))
        for item in 'xbyre':
            path = os.path.join(model_dir, 'norm_%s.npy' % item)
            if os.path.exists(path):
                self._norms[item] = torch.as_tensor(np.load(path), dtype=torch.float32)
        dev = next(self.parameters()).device
        path = os.path.join(model_dir, 'optim.pt')
        if retrain and os.path.exists(path):
            if self._optimizer is None:
                self._optimizer = KerasAdam([p for p in self.parameters()], self.learning_rate, clipnorm=1.0)
            self._optimizer.load_state_dict(torch.load(path, weights_only=True))
        path = os.path.join(model_dir, 'gradnorm.pt')
        if retrain and self.gradnorm and os.path.exists(path):
            sd = torch.load(path, weights_only=True)
            alpha = self._alphas(dev)
            with torch.no_grad():
                alpha.copy_(sd['alpha'])
            self._alpha_optimizer.load_state_dict(sd['optimizer'])
