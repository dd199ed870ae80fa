"""CPU oracle of the whole surrogate forward, post-processing and rollout (TEST INFRASTRUCTURE ONLY;
PARITY UNPINNED, see oracle/__init__.py).

Restates, on torch CPU tensors (fp64 or fp32), `surrogate/emulator.py` of the reference:
  forward            build_network, conv = GAT / GCN / Diffusion, recurrent = Conv1D / GRU / LSTM / none      emulator.py:166-341
  normalize          min-max (de)normalisation                                 emulator.py:803-810
  get_edge_action / get_action   action -> per-link / per-node gates           emulator.py:364-398
  post_proc          post_proc_tf                                              emulator.py:680-725
  constrain          constrain_tf                                              emulator.py:750-770
  predict            predict_tf                                                emulator.py:604-641
  post_proc_np / predict_np / simulate   the NumPy-mode twins (`post_proc`, `predict`, `simulate`'s per-step loop)
                                                                               emulator.py:643-678, 566-602, 521-564
  model_rollout      _model with roll > 0 (curriculum / autoregressive chunks) emulator.py:400-438

`args` is any object with the reference's attribute names (`state_shape`, `edge_state_shape`, `seq_in`,
`embed_size`, `adj`, `edge_adj`, `node_edge`, ... -- `Emulator.__init__`, emulator.py:48-127); `params`
is the dict made by `init_params` (Keras creation order and initialisers).
Not restated (none of the reference's shipped model configurations uses them): use_adj for GCN / Diffusion (per-step re-normalised
filters), GeneralConv, training-time dropout.
"""
import math
from types import SimpleNamespace

import numpy as np
import torch

from . import spektral_dense as OD


def config(args):
    """Derived sizes, exactly as `Emulator.__init__` computes them (emulator.py:49-108)."""
    g = lambda k, d=None: getattr(args, k, d)
    c = SimpleNamespace()
    c.n_node, n_in0 = g('state_shape', (40, 4))
    c.tide = bool(g('tide', False))
    c.b_in = 2 if c.tide else 1
    act = g('act', False)
    c.act = bool(act and act != 'False')
    c.n_out = n_in0 - 1
    c.seq_in, c.seq_out = g('seq_in', 6), g('seq_out', 1)
    c.d, c.H = g('embed_size', 64), g('hidden_dim', 64)
    c.k, c.L, c.n_tp = g('kernel_size', 3), g('n_sp_layer', 3), g('n_tp_layer', 2)
    c.recurrent = g('recurrent', 'Conv1D')                # get_tem_nets (:154-163): 'Conv1D' | 'GRU' | 'LSTM' | anything else = none
    if c.recurrent not in ('Conv1D', 'GRU', 'LSTM') or not c.n_tp:
        c.recurrent, c.n_tp, c.H = None, 0, c.d           # no temporal net: the width stays embed_size
    c.activation = g('activation', 'relu')
    c.if_flood = int(g('if_flood', 0))
    c.n_in = n_in0 + (1 if c.if_flood else 0)
    c.is_outfall = np.asarray(g('is_outfall', np.zeros(c.n_node)), dtype=np.float64)
    c.epsilon = g('epsilon', -1.0)
    c.edge_fusion = bool(g('edge_fusion', False))
    c.edges = np.asarray(g('edges'))
    c.n_edge, c.e_in = g('edge_state_shape', (40, 4))
    c.e_out = c.e_in - 1
    c.ehmax = np.asarray(g('ehmax', np.full(c.n_edge, 0.5)), dtype=np.float64)
    c.pump = np.asarray(g('pump', np.zeros(c.n_edge)), dtype=np.float64)
    c.node_edge = np.asarray(g('node_edge'), dtype=np.float64)
    if c.edge_fusion:
        c.n_out -= 2
    c.adj = np.asarray(g('adj', np.eye(c.n_node)))
    c.edge_adj = np.asarray(g('edge_adj', np.eye(c.n_edge)))
    c.act_edges = np.asarray(g('act_edges', np.zeros((0, 2), dtype=int))) if c.act else None
    c.area = np.asarray(g('area', np.zeros(c.n_node)), dtype=np.float64)
    c.pump_in = np.asarray(g('pump_in', np.zeros(c.n_node)), dtype=np.float64)
    c.pump_out = np.asarray(g('pump_out', np.zeros(c.n_node)), dtype=np.float64)
    c.offset = np.asarray(g('offset', np.zeros(c.n_edge)), dtype=np.float64)
    c.hmax = np.asarray(g('hmax', np.full(c.n_node, 1.5)), dtype=np.float64)
    c.hmin = np.asarray(g('hmin', np.zeros(c.n_node)), dtype=np.float64)
    conv = g('conv', 'GAT')
    c.mlp = conv in (None, False, 'None', 'False', 'NoneType')          # `net = Dense`: the non-graph baseline (emulator.py:181-182)
    conv = 'GAT' if c.mlp else conv
    c.conv = 'GCN' if 'GCN' in conv else 'Diffusion' if 'Diff' in conv else 'GAT'     # emulator.py:131-142
    c.resnet = bool(g('resnet', False))
    c.roll = int(g('roll', 0))
    c.dropout = float(g('dropout', 0.0) or 0.0)           # rate of the Dropout layers behind the spatial layers / the resnet Dense (:65)
    c.graph_base = int(g('graph_base', 0))                # 0: node graph + line graph; 1 / 2: one graph over nodes AND links (:220-223)
    if c.conv == 'GAT':                                   # emulator.py:143-145
        c.filter = (c.adj > 0).astype(np.float64)
        c.edge_filter = (c.edge_adj > 0).astype(np.float64)
    else:                                                 # emulator.py:133-134 / :137-138
        pre = OD.gcn_preprocess if c.conv == 'GCN' else OD.diffusion_preprocess
        c.filter = pre(torch.from_numpy(c.adj.astype(np.float64))).numpy()
        c.edge_filter = pre(torch.from_numpy(c.edge_adj.astype(np.float64))).numpy()
    return c


def _glorot(gen, shape):
    if len(shape) == 1:
        fan_in = fan_out = shape[0]
    elif len(shape) == 2:
        fan_in, fan_out = shape
    else:
        rf = int(np.prod(shape[:-2]))
        fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return (torch.rand(shape, generator=gen, dtype=torch.float64) * 2 - 1) * lim


def init_params(args, seed=1, bias_scale=0.05):
    """Parameters in Keras creation order.  Kernels glorot_uniform, NodeEdge weights
    N(0, 0.05^2); biases get small random values instead of Keras' zeros so the tests exercise them."""
    c = config(args)
    g = torch.Generator().manual_seed(seed)
    d, h, H = c.d, c.d // 2, c.H
    bias = lambda n: torch.randn(n, generator=g, dtype=torch.float64) * bias_scale
    dense = lambda fi, fo: {'kernel': _glorot(g, (fi, fo)), 'bias': bias(fo)}

    def spatial(fx, fe):
        if c.graph_base:                                      # one conv over concat([x, e], axis=-2) (emulator.py:220-223,273-276)
            assert fx == fe, 'graph_base concatenates node and link rows: equal widths needed'
            if c.conv == 'Diffusion':
                return {'gat': {'theta': (torch.rand(d, 7, generator=g, dtype=torch.float64) * 2 - 1) * math.sqrt(6.0 / 14)}}
            return {'gat': {'kernel': _glorot(g, (fx, 1, d)), 'attn_kernel_self': _glorot(g, (d, 1, 1)),
                            'attn_kernel_neighs': _glorot(g, (d, 1, 1)), 'bias': bias(d)}}
        if c.conv == 'Diffusion':      # `d` DiffuseFeatures filters of K + 1 = 7 coefficients each (spektral default K = 6), no bias
            conv = lambda f: {'theta': (torch.rand(d, 7, generator=g, dtype=torch.float64) * 2 - 1) * math.sqrt(6.0 / 14)}
        else:
            conv = lambda f: {'kernel': _glorot(g, (f, 1, d)), 'attn_kernel_self': _glorot(g, (d, 1, 1)),
                              'attn_kernel_neighs': _glorot(g, (d, 1, 1)), 'bias': bias(d)}
        return {'dense_xe': dense(fe, h), 'dense_ex': dense(fx, h),
                'node_edge_n': {'weight': torch.randn(c.n_node, c.n_edge, generator=g, dtype=torch.float64) * 0.05,
                                'bias': torch.zeros(c.n_node, c.n_edge, dtype=torch.float64)},
                'node_edge_e': {'weight': torch.randn(c.n_edge, c.n_node, generator=g, dtype=torch.float64) * 0.05,
                                'bias': torch.zeros(c.n_edge, c.n_node, dtype=torch.float64)},
                'gat_x': conv(fx + h), 'gat_e': conv(fe + h)}

    def temporal(f):
        out, fi = [], f
        for _ in range(c.n_tp):
            if c.recurrent == 'Conv1D':
                out.append({'kernel': _glorot(g, (c.k, fi, H)), 'bias': bias(H)})
            else:      # keras GRU (reset_after=True: bias (2, 3H)) / LSTM (bias (4H,)): kernel (F, G H), recurrent_kernel (H, G H)
                G = 3 if c.recurrent == 'GRU' else 4
                out.append({'kernel': _glorot(g, (fi, G * H)), 'recurrent_kernel': _glorot(g, (H, G * H)),
                            'bias': bias(2 * G * H).reshape(2, G * H) if G == 3 else bias(G * H)})
            fi = H
        return out

    if c.mlp:            # build_network(conv=False): flattened rows, Dense(2 d) blocks, heads for all nodes / links at once
        N, E = c.n_node, c.n_edge
        p = {'embed_x': dense(N * c.n_in, d), 'embed_b': dense(N * c.b_in, h), 'embed_e': dense(E * c.e_in, d)}
        if c.act:
            p['embed_ae'] = dense(E, h)
        p['block1'] = [dense(2 * d, 2 * d) for _ in range(c.L)]
        p['tem1_x'], p['tem1_e'] = temporal(d), temporal(d)
        Ht = c.H
        f2 = (Ht + h) + (Ht + (h if c.act else 0))
        p['block2'] = [dense(f2 if i == 0 else 2 * d, 2 * d) for i in range(c.L)]
        p['tem2_x'], p['tem2_e'] = temporal(d), temporal(d)
        p['res_x'], p['res_e'] = dense(Ht, d), dense(Ht, d)
        p['out'] = dense(d, c.n_out * N)
        p['flood'] = []
        fi = d
        for _ in range(c.if_flood):
            p['flood'].append(dense(fi, h))
            fi = h
        if c.if_flood:
            p['flood_out'] = dense(fi, N)
        p['e_out'] = dense(d, c.e_out * E)
        return p
    p = {'embed_x': dense(c.n_in, d), 'embed_b': dense(c.b_in, h), 'embed_e': dense(c.e_in, d)}
    if c.act:
        p['embed_ae'] = dense(1, h)
    p['block1'] = [spatial(d, d) for _ in range(c.L)]
    p['tem1_x'], p['tem1_e'] = temporal(d), temporal(d)
    fx2, fe2 = H + h, H + (h if c.act else 0)
    p['block2'] = [spatial(fx2 if i == 0 else d, fe2 if i == 0 else d) for i in range(c.L)]
    p['tem2_x'], p['tem2_e'] = temporal(d), temporal(d)
    p['res_x'], p['res_e'] = dense(H, d), dense(H, d)
    p['out'] = dense(d, c.n_out)
    p['flood'] = []
    fi = d
    for _ in range(c.if_flood):
        p['flood'].append(dense(fi, h))
        fi = h
    if c.if_flood:
        p['flood_out'] = dense(fi, 1)
    p['e_out'] = dense(d, c.e_out)
    return p


def cast_params(p, dtype):
    if isinstance(p, dict):
        return {k: cast_params(v, dtype) for k, v in p.items()}
    if isinstance(p, list):
        return [cast_params(v, dtype) for v in p]
    return p.to(dtype)


def conv1d_causal(x, kernel, bias, dilation, act):
    """keras Conv1D(padding='causal', dilation_rate): x (M,T,F), kernel (k,F,H) -- emulator.py:155-157."""
    k = kernel.shape[0]
    T = x.shape[1]
    out = torch.zeros(x.shape[0], T, kernel.shape[2], dtype=x.dtype)
    for j in range(k):
        shift = (k - 1 - j) * dilation
        if shift < T:
            out[:, shift:] += x[:, :T - shift] @ kernel[j]
    return OD.activation(act)(out + bias)


def get_adj_action(c, a, g=True):
    """`get_adj_action` (emulator.py:343-362), GAT branch: a (B, T, n_act) -> (B, T, N, N) integer adjacency with the entry
    (from_k, to_k) of every actuated link multiplied by its setting (g=True: the graph form, `self.adj * gather(...)`,
    :345-348) or SET to its setting (g=False: the numpy form `adj[act_edges] = s`, :350-354, what `predict` / `simulate` use);
    the two agree on a 0/1 adjacency.  Then cast to int (truncation, :361)."""
    adj = torch.from_numpy(np.asarray(c.adj, dtype=np.float64))
    out = adj.expand(tuple(a.shape[:-1]) + adj.shape).clone()
    for k, (u, v) in enumerate(np.asarray(c.act_edges, dtype=np.int64)):
        out[..., u, v] = (adj[u, v] if g else 1.0) * a[..., k].to(torch.float64)
    return torch.trunc(out)


def gru_sequence(x, kernel, recurrent_kernel, bias):
    """keras.layers.GRU(H, return_sequences=True) as TF 2.10 builds it (emulator.py:159; defaults activation='tanh',
    recurrent_activation='sigmoid', reset_after=True, zero initial state): x (M, T, F), kernel (F, 3H) / recurrent_kernel (H, 3H)
    in gate order z, r, h, bias (2, 3H) = [input bias, recurrent bias].
        z = sig(x_z + h U_z), r = sig(x_r + h U_r), cand = tanh(x_h + r * (h U_h)), h' = z h + (1 - z) cand   (biases inside)."""
    H = recurrent_kernel.shape[0]
    xp = x @ kernel + bias[0]
    h = torch.zeros(x.shape[0], H, dtype=x.dtype)
    out = []
    for t in range(x.shape[1]):
        rp = h @ recurrent_kernel + bias[1]
        z = torch.sigmoid(xp[:, t, :H] + rp[:, :H])
        r = torch.sigmoid(xp[:, t, H:2 * H] + rp[:, H:2 * H])
        cand = torch.tanh(xp[:, t, 2 * H:] + r * rp[:, 2 * H:])
        h = z * h + (1 - z) * cand
        out.append(h)
    return torch.stack(out, dim=1)


def lstm_sequence(x, kernel, recurrent_kernel, bias):
    """keras.layers.LSTM(H, return_sequences=True) (emulator.py:161; tanh / sigmoid, zero initial state): gate order i, f, c, o,
    z = x W + h U + b;  c' = sig(z_f) c + sig(z_i) tanh(z_c);  h' = sig(z_o) tanh(c')."""
    H = recurrent_kernel.shape[0]
    xp = x @ kernel + bias
    h = torch.zeros(x.shape[0], H, dtype=x.dtype)
    cst = torch.zeros_like(h)
    out = []
    for t in range(x.shape[1]):
        z = xp[:, t] + h @ recurrent_kernel
        cst = torch.sigmoid(z[:, H:2 * H]) * cst + torch.sigmoid(z[:, :H]) * torch.tanh(z[:, 2 * H:3 * H])
        h = torch.sigmoid(z[:, 3 * H:]) * torch.tanh(cst)
        out.append(h)
    return torch.stack(out, dim=1)


# Large rollouts (tests/test_gpu_emulator.py::test_c2_rollout_100_fed_back_steps): route the GAT spatial layers through the
# SPARSE restatement (oracle/sparse_csr.py; tests/test_oracle_dual.py holds the two to 1e-12 of each other in fp64) -- the
# dense-masked form needs (S, E, E) logits per layer, minutes per step at E = 2 500.
SPARSE_SPATIAL = False
_csr_cache = {}


def _spatial_layer(x, e, p, c, dtype, adj=None):
    if SPARSE_SPATIAL and c.conv == 'GAT' and adj is None:
        from . import sparse_csr as OS
        key = (id(c.adj), id(c.edge_adj))          # `config` passes an ndarray `args.adj` through as the same object
        if key not in _csr_cache:
            _csr_cache.clear()                         # a new network: the per-layer entries below belong to the old one
            _csr_cache[key] = (OS.csr_from_dense(c.filter, True)[:2], OS.csr_from_dense(c.edge_filter, True)[:2],
                               torch.from_numpy(c.node_edge), c.adj, c.edge_adj)      # the arrays are kept alive with their ids
        a_csr, ea_csr, ne = _csr_cache[key][:3]
        pk = ('layer', id(p['node_edge_n']['weight']))
        if pk not in _csr_cache:                   # the support form of this layer's dense NodeEdge parameters, once (a bias off the
            inci = ne.abs()                        # support has no sparse form: the dense restatement must be used for it)
            rn, cn, vn, rest_n = OS.node_edge_support(inci, p['node_edge_n']['weight'], p['node_edge_n']['bias'])
            re_, ce, ve, rest_e = OS.node_edge_support(inci.T, p['node_edge_e']['weight'], p['node_edge_e']['bias'])
            assert rest_n is None and rest_e is None, 'SPARSE_SPATIAL: NodeEdge bias is non-zero off the incidence support'
            _csr_cache[pk] = ((rn, cn), (re_, ce), vn, ve, p['node_edge_n']['weight'])
        inc_n, inc_e, vn, ve = _csr_cache[pk][:4]
        q = {'xe_k': p['dense_xe']['kernel'], 'xe_b': p['dense_xe']['bias'], 'ex_k': p['dense_ex']['kernel'], 'ex_b': p['dense_ex']['bias'],
             'ne_n_v': vn.to(dtype), 'ne_e_v': ve.to(dtype), 'gx_k': p['gat_x']['kernel'], 'gx_as': p['gat_x']['attn_kernel_self'],
             'gx_an': p['gat_x']['attn_kernel_neighs'], 'gx_b': p['gat_x']['bias'], 'ge_k': p['gat_e']['kernel'],
             'ge_as': p['gat_e']['attn_kernel_self'], 'ge_an': p['gat_e']['attn_kernel_neighs'], 'ge_b': p['gat_e']['bias']}
        return OS.spatial_layer_csr(x, e, q, a_csr, ea_csr, inc_n, inc_e, act=c.activation)
    q = {'xe_k': p['dense_xe']['kernel'], 'xe_b': p['dense_xe']['bias'], 'ex_k': p['dense_ex']['kernel'],
         'ex_b': p['dense_ex']['bias'], 'ne_n_w': p['node_edge_n']['weight'], 'ne_n_b': p['node_edge_n']['bias'],
         'ne_e_w': p['node_edge_e']['weight'], 'ne_e_b': p['node_edge_e']['bias']}
    if c.conv == 'Diffusion':
        q.update(gx_theta=p['gat_x']['theta'], ge_theta=p['gat_e']['theta'])
        return OD.spatial_layer_dense(x, e, q, torch.from_numpy(c.filter).to(dtype), torch.from_numpy(c.edge_filter).to(dtype),
                                      torch.from_numpy(c.node_edge).to(dtype), c.activation, c.conv)
    q = {**q, 'gx_k': p['gat_x']['kernel'], 'gx_as': p['gat_x']['attn_kernel_self'], 'gx_an': p['gat_x']['attn_kernel_neighs'],
         'gx_b': p['gat_x']['bias'], 'ge_k': p['gat_e']['kernel'], 'ge_as': p['gat_e']['attn_kernel_self'],
         'ge_an': p['gat_e']['attn_kernel_neighs'], 'ge_b': p['gat_e']['bias']}
    return OD.spatial_layer_dense(x, e, q, torch.from_numpy(c.filter).to(dtype) if adj is None else adj.to(dtype),
                                  torch.from_numpy(c.edge_filter).to(dtype), torch.from_numpy(c.node_edge).to(dtype), c.activation, c.conv)


# Training-time dropout (emulator.py:199-213,234-235,287-288,314-318 with `training=fit`, :411,434): None = inference (identity);
# a test sets it to a callable (tensor, rate) -> tensor that applies the mask stream it wants to compare with (oracle/dropout_ref.py).
DROPOUT = None


def _drop(t, rate):
    return t if DROPOUT is None or not rate else DROPOUT(t, rate)


def _forward_mlp(c, params, X, B, E, AE, act, D):
    """`build_network(conv=False)` (emulator.py:166-341, `net = Dense`): the reference's non-graph baseline (`*_nncat_*` models)."""
    nb = X.shape[0]
    dr = c.dropout
    d02 = 0.2 if dr else 0.0
    flat = lambda t: t.reshape(t.shape[0], t.shape[1], -1)        # :197,202,205,211: (B, T, N, c) -> (B, T, N * c)
    x = _drop(D(flat(X), params['embed_x']), d02)
    res = x[:, -1:]
    x = act(x)
    b = _drop(D(flat(B), params['embed_b'], c.activation), d02)
    e = _drop(D(flat(E), params['embed_e']), d02)
    res_e = e[:, -1:]
    e = act(e)
    ae = _drop(D(flat(AE), params['embed_ae'], c.activation), d02) if c.act else None

    def spatial(x, e, layers):                                    # :236-237
        for p in layers:
            z = D(torch.cat([x, e], dim=-1), p, c.activation)
            x, e = _drop(z[..., :z.shape[-1] // 2], dr), _drop(z[..., z.shape[-1] // 2:], dr)
        return x, e

    def temporal(y, layers):                                      # rows are the batch elements: (B, T, F) as it stands
        for i, p in enumerate(layers):
            if c.recurrent == 'Conv1D':
                y = conv1d_causal(y, p['kernel'], p['bias'], 2 ** i, c.activation)
            else:
                y = (gru_sequence if c.recurrent == 'GRU' else lstm_sequence)(y, p['kernel'], p['recurrent_kernel'], p['bias'])
        return y

    x, e = spatial(x, e, params['block1'])
    x, e = temporal(x, params['tem1_x'])[:, -c.seq_out:], temporal(e, params['tem1_e'])[:, -c.seq_out:]
    x = torch.cat([x, b], dim=-1)
    if c.act:
        e = torch.cat([e, ae], dim=-1)
    x, e = spatial(x, e, params['block2'])
    x, e = temporal(x, params['tem2_x']), temporal(e, params['tem2_e'])
    x_out = _drop(D(x, params['res_x']), dr)
    e_out = _drop(D(e, params['res_e']), dr)
    if c.resnet:
        x, e = act(torch.cumsum(x_out, dim=1) + res), act(torch.cumsum(e_out, dim=1) + res_e)
    else:
        x, e = act(x_out), act(e_out)
    out = D(x, params['out'], 'hard_sigmoid').reshape(nb, c.seq_out, c.n_node, c.n_out)
    if c.if_flood:
        f = x
        for p in params['flood']:
            f = D(f, p, c.activation)
        out = torch.cat([out, D(f, params['flood_out'], 'sigmoid').reshape(nb, c.seq_out, c.n_node, 1)], dim=-1)
    return out, D(e, params['e_out'], 'tanh').reshape(nb, c.seq_out, c.n_edge, c.e_out)


def forward(args, params, X, B, E, AE=None, ADJ=None):
    """`Emulator.build_network` as a function: X (B,T_in,N,n_in), B (B,T_out,N,b_in), E (B,T_in,E,e_in),
    AE (B,T_out,E,1) when act -> out (B,T_out,N,n_out[+1]), e_out (B,T_out,E,e_out).  emulator.py:195-338."""
    c = config(args)
    dt = X.dtype
    act = OD.activation(c.activation)
    D = lambda t, p, a='linear': OD.dense(t, p['kernel'], p['bias'], a)
    if c.mlp:
        return _forward_mlp(c, params, X, B, E, AE, act, D)
    dr = c.dropout
    d02 = 0.2 if dr else 0.0                                      # `Dropout(0.2)(x) if self.dropout else x`
    x = _drop(D(X, params['embed_x']), d02)                       # :198-199
    res = x[:, -1:]                                               # :200
    x = act(x)
    b = _drop(D(B, params['embed_b'], c.activation), d02)         # :203-204
    e = _drop(D(E, params['embed_e']), d02)                       # :206-207
    res_e = e[:, -1:]
    e = act(e)
    ae = _drop(D(AE, params['embed_ae'], c.activation), d02) if c.act else None      # :212-213
    nb = X.shape[0]

    def spatial(x, e, layers, adj=None):                          # :217-235 / :265-288; adj: (B,T,n,n) of use_adj (:268-271)
        T = x.shape[1]
        xs, es = x.reshape((-1,) + tuple(x.shape[2:])), e.reshape((-1,) + tuple(e.shape[2:]))
        A = None if adj is None else adj.reshape((-1,) + tuple(adj.shape[-2:]))
        for p in layers:
            if c.graph_base:
                q = p['gat']
                if c.conv == 'Diffusion':
                    z = _drop(OD.diffusion_conv_dense(torch.cat([xs, es], dim=-2), torch.from_numpy(c.filter).to(dt), q['theta'], c.activation), dr)
                    xs, es = z[:, :c.n_node], z[:, c.n_node:]
                    continue
                z = OD.gat_conv_dense(torch.cat([xs, es], dim=-2), torch.from_numpy(c.filter).to(dt) if A is None else A.to(dt), q['kernel'], q['attn_kernel_self'],
                                      q['attn_kernel_neighs'], q['bias'], c.activation)
                z = _drop(z, dr)                                  # :234-235 (one mask over the stacked rows = one per half)
                xs, es = z[:, :c.n_node], z[:, c.n_node:]
            else:
                xs, es = _spatial_layer(xs, es, p, c, dt, A)
                xs, es = _drop(xs, dr), _drop(es, dr)             # :234-235,287-288
        return xs.reshape(nb, T, c.n_node, -1), es.reshape(nb, T, c.n_edge, -1)

    def temporal(x, layers, n):                                   # :244-257 / :299-310
        T = x.shape[1]
        y = x.permute(0, 2, 1, 3).reshape(-1, T, x.shape[-1])
        for i, p in enumerate(layers):
            if c.recurrent == 'Conv1D':
                y = conv1d_causal(y, p['kernel'], p['bias'], 2 ** i, c.activation)
            else:
                y = (gru_sequence if c.recurrent == 'GRU' else lstm_sequence)(y, p['kernel'], p['recurrent_kernel'], p['bias'])
        return y.reshape(nb, n, T, -1).permute(0, 2, 1, 3)

    x, e = spatial(x, e, params['block1'])
    x = temporal(x, params['tem1_x'], c.n_node)[:, -c.seq_out:]   # :249
    e = temporal(e, params['tem1_e'], c.n_edge)[:, -c.seq_out:]
    x = torch.cat([x, b], dim=-1)                                 # :260
    if c.act:
        e = torch.cat([e, ae], dim=-1)                            # :262
    x, e = spatial(x, e, params['block2'], ADJ)                       # :264-288 (A = A_in with use_adj)
    x = temporal(x, params['tem2_x'], c.n_node)
    e = temporal(e, params['tem2_e'], c.n_edge)
    x_out = _drop(D(x, params['res_x']), dr)                      # :313-314
    x = act(torch.cumsum(x_out, dim=1) + res) if c.resnet else act(x_out)   # :315-316
    e_o = _drop(D(e, params['res_e']), dr)                        # :317-318
    e = act(torch.cumsum(e_o, dim=1) + res_e) if c.resnet else act(e_o)     # :319-320
    out = D(x, params['out'], 'hard_sigmoid')                     # :324
    if c.if_flood:
        f = x
        for p in params['flood']:
            f = D(f, p, c.activation)                             # :329
        out = torch.cat([out, D(f, params['flood_out'], 'sigmoid')], dim=-1)    # :330-333
    return out, D(e, params['e_out'], 'tanh')                     # :336


def normalize(norms, dat, item, inverse=False):
    """emulator.py:803-810; norms[item] is (2, n, C) = [max, min]."""
    normal = norms[item]
    dim = dat.shape[-1]
    maxi, mini = normal[0, ..., :dim], normal[1, ..., :dim]
    return dat * (maxi - mini) + mini if inverse else (dat - mini) / (maxi - mini)


def _act_edge_index(c):
    hits = [np.where((c.edges == ae).all(1))[0] for ae in c.act_edges]     # emulator.py:386-388
    flat = [int(i) for e in hits for i in e]
    return sorted(set(flat), key=flat.index)


def get_edge_action(c, a):
    """emulator.py:385-392 (tensor branch): 1 on free links, the setting on actuated ones -> (B,T,E,1)."""
    out = np.zeros(c.n_edge, dtype=np.int64)
    idx = _act_edge_index(c)
    out[idx] = np.arange(1, a.shape[-1] + 1)
    table = torch.cat([torch.ones_like(a[..., :1]), a], dim=-1)
    return table[..., torch.from_numpy(out)].unsqueeze(-1)


def get_action(c, a):
    """emulator.py:364-371 (tensor branch): per-node outflow / inflow gates."""
    out_o, out_i = np.zeros(c.n_node, dtype=np.int64), np.zeros(c.n_node, dtype=np.int64)
    out_o[c.act_edges[:, 0]] = np.arange(1, a.shape[-1] + 1)
    out_i[c.act_edges[:, 1]] = np.arange(1, a.shape[-1] + 1)
    table = torch.cat([torch.ones_like(a[..., :1]), a], dim=-1)
    return table[..., torch.from_numpy(out_o)], table[..., torch.from_numpy(out_i)]


def post_proc(args, norms, preds, edge_preds, a, b):
    """`post_proc_tf` (emulator.py:680-725): tide, offset gate, pump rating, action gates, link->node flow balance."""
    c = config(args)
    dt = preds.dtype
    T = lambda v: torch.as_tensor(v, dtype=dt)
    ne = T(c.node_edge)
    pos = ne.clamp(0, 1)
    if c.tide:                                                    # :684-686
        h = preds[..., 0] * (1 - T(c.is_outfall)) + b[..., -1]
        preds = torch.cat([h.unsqueeze(-1), preds[..., 1:]], dim=-1)
    if c.offset.max() > 0:                                        # :688-694
        inoff = (normalize(norms, preds, 'y', True)[..., 0] - T(c.hmin)) @ pos
        flow = edge_preds[..., -1]
        off = T(c.offset)
        flow = (flow * (flow > 0).to(dt) * (off > 0).to(dt) * (inoff > off).to(dt) + flow * (flow <= 0).to(dt) * (off > 0).to(dt) +
                flow * (off == 0).to(dt)).unsqueeze(-1)
        edge_preds = torch.cat([edge_preds[..., :-1], flow], dim=-1)
    if c.act:                                                     # :696-715
        if c.pump.min() > 0:
            fl = T(c.pump) * ((preds[..., 0] > 0.01).to(dt) @ pos)
            fl = fl * (norms['e'][0, :, 2] > 1e-3).to(dt) / norms['e'][0, :, 2]
            flow = (edge_preds[..., -1] * (fl == 0).to(dt) + fl).unsqueeze(-1)
        else:
            flow = edge_preds[..., -1:]
        ae = get_edge_action(c, a)
        edge_preds = torch.cat([edge_preds[..., :-1], flow * ae], dim=-1)
        if not c.edge_fusion:
            a_out, a_in = get_action(c, a[:, :c.seq_out])
            fli = T(c.pump_in) * (preds[..., 0] > 0).to(dt) / norms['y'][0, :, 1]
            flo = T(c.pump_out) * (preds[..., 0] > 0).to(dt) / norms['y'][0, :, 2]
            inflow = preds[..., 1] * (fli == 0).to(dt) + fli
            outflow = preds[..., 2] * (flo == 0).to(dt) + flo
            preds = torch.cat([torch.stack([preds[..., 0], inflow * a_in, outflow * a_out], dim=-1), preds[..., 3:]], dim=-1)
    if c.edge_fusion:                                             # :718-724
        flow = normalize(norms, edge_preds, 'e', True)[..., -1:]
        neg = ne.clamp(-1, 0).abs()
        fp, fn = flow.clamp(min=0), -flow.clamp(max=0)
        node_out = pos @ fp + neg @ fn
        node_in = neg @ fp + pos @ fn
        ny = norms['y']
        node_out = node_out * (ny[0, :, 2:3] > 1e-3).to(dt) / ny[0, :, 2:3]
        node_in = node_in * (ny[0, :, 1:2] > 1e-3).to(dt) / ny[0, :, 1:2]
        preds = torch.cat([preds[..., :1], node_in, node_out, preds[..., 1:]], dim=-1)
    return preds, edge_preds


def constrain(args, y, r):
    """`constrain_tf` (emulator.py:750-770): depth clip, flooding volume, flood gating."""
    c = config(args)
    dt = y.dtype
    T = lambda v: torch.as_tensor(v, dtype=dt)
    h, q_us, q_ds = y[..., 0], y[..., 1], y[..., 2]
    r = r.squeeze(-1)
    h = torch.minimum(torch.maximum(h, T(c.hmin)), T(c.hmax))
    q_w = (q_us + r - q_ds).clamp(min=0) * (1 - T(c.is_outfall))
    if c.if_flood:
        f = (y[..., -1] > 0.5).to(dt)
        h = T(c.hmax) * f + h * (1 - f)
        y = torch.stack([h, q_us, q_ds, y[..., -1]], dim=-1)
    else:
        y = torch.stack([h, q_us, q_ds], dim=-1)
    if c.epsilon > 0:
        q_w = q_w * ((T(c.hmax) - h) < c.epsilon).to(dt)
    elif c.epsilon == 0:
        pass
    elif c.if_flood:
        q_w = q_w * f
    return q_w, y


def predict(args, params, norms, states, b, a=None, edge_state=None):
    """`predict_tf` (emulator.py:604-641): raw states in, de-normalised (B,T,N,5) / (B,T,E,3) out."""
    c = config(args)
    dt = states.dtype
    T = lambda v: torch.as_tensor(v, dtype=dt)
    x = states[:, -c.seq_in:]
    ex = edge_state[:, -c.seq_in:]
    assert b.shape[1] == c.seq_out
    ae = get_edge_action(c, a) if c.act else None
    y, ey = forward(args, params, normalize(norms, x, 'x'), normalize(norms, b, 'b'), normalize(norms, ex, 'e'), ae)
    y, ey = post_proc(args, norms, y, ey, a, normalize(norms, b, 'b'))
    ey = normalize(norms, ey, 'e', True)
    ey = torch.cat([torch.minimum(ey[..., 0].clamp(min=0), T(c.ehmax)).unsqueeze(-1), ey[..., 1:]], dim=-1)   # :626
    y = normalize(norms, y, 'y', True)
    if c.pump_in.sum() + c.pump_out.sum() + c.pump.sum() > 0:     # :630-638
        ps = ((T(c.area) * (T(c.node_edge).clamp(0, 1) @ T(c.pump))) > 0).to(dt)
        h, qin, qout = y[..., 0], y[..., 1], y[..., 2]
        de = []
        for t in range(c.seq_out):
            prev = x[:, -1, :, 0] if t == 0 else de[-1] + (qin - qout)[:, t] / (T(c.area) + 1e-6)
            de.append(torch.minimum(torch.maximum(prev, T(c.hmin)), T(c.hmax)))
        de = torch.stack(de, dim=1)
        y = torch.cat([(h * (1 - ps) + de * ps).unsqueeze(-1), y[..., 1:]], dim=-1)
    q_w, y = constrain(args, y, b[..., :1])
    return torch.cat([y, q_w.unsqueeze(-1)], dim=-1), ey


def post_proc_np(args, norms, y, ey, a, b):
    """`post_proc` (emulator.py:643-678), the NumPy twin of `post_proc_tf` used by `predict` and `simulate`, restated in
    NumPy op for op.  It is NOT the same function: the offset gate has no `offset.max() > 0` guard (:649-653), the rated-
    pump override of the link flow is applied whenever the model has actions -- no `pump.min() > 0` guard (:656-659 vs
    :698) -- and the node pumps of the non-edge-fusion form open at depth > 0.01 instead of > 0 (:664-665 vs :710-711)."""
    c = config(args)
    y, ey = np.array(y, dtype=np.float64), np.array(ey, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    N = {k: np.asarray(v, dtype=np.float64) for k, v in norms.items()}
    nz = lambda dat, item, inv=False: (dat * (N[item][0, ..., :dat.shape[-1]] - N[item][1, ..., :dat.shape[-1]]) + N[item][1, ..., :dat.shape[-1]]) if inv \
        else (dat - N[item][1, ..., :dat.shape[-1]]) / (N[item][0, ..., :dat.shape[-1]] - N[item][1, ..., :dat.shape[-1]])
    ne = np.asarray(c.node_edge, dtype=np.float64)
    if c.tide:                                                    # :645-647
        h = y[..., 0] * (1 - c.is_outfall) + b[..., -1]
        y = np.concatenate([np.expand_dims(h, -1), y[..., 1:]], axis=-1)
    inoff = np.matmul(nz(y, 'y', True)[..., 0] - c.hmin, np.clip(ne, 0, 1))      # :649
    flow = np.expand_dims(ey[..., -1] * (ey[..., -1] > 0) * (c.offset > 0) * (inoff > c.offset) +
                          ey[..., -1] * (ey[..., -1] <= 0) * (c.offset > 0) + ey[..., -1] * (c.offset == 0), axis=-1)      # :650-652
    ey = np.concatenate([ey[..., :-1], flow], axis=-1)
    if c.act:                                                     # :654-671
        at = torch.as_tensor(np.asarray(a), dtype=torch.float64)
        ae = get_edge_action(c, at).numpy()
        fl = c.pump * np.matmul((y[..., 0] > 0.01).astype(np.float64), np.clip(ne, 0, 1))      # :657
        fl = fl * (N['e'][0, :, 2] > 1e-3) / N['e'][0, :, 2]
        ey[..., -1] = ey[..., -1] * (fl == 0) + fl
        ey[..., -1:] = ey[..., -1:] * ae
        if not c.edge_fusion:
            a_out, a_in = (t.numpy() for t in get_action(c, at[:, :c.seq_out]))
            fli = c.pump_in * (y[..., 0] > 0.01) / N['y'][0, :, 1]
            flo = c.pump_out * (y[..., 0] > 0.01) / N['y'][0, :, 2]
            y[..., 1] = y[..., 1] * (fli == 0) + fli
            y[..., 2] = y[..., 2] * (flo == 0) + flo
            y[..., 2] = y[..., 2] * a_out
            y[..., 1] = y[..., 1] * a_in
    if c.edge_fusion:                                             # :672-677
        efl = nz(ey, 'e', True)[..., -1:]
        node_out = np.matmul(np.clip(ne, 0, 1), np.clip(efl, 0, np.inf)) + np.matmul(np.abs(np.clip(ne, -1, 0)), -np.clip(efl, -np.inf, 0))
        node_in = np.matmul(np.abs(np.clip(ne, -1, 0)), np.clip(efl, 0, np.inf)) + np.matmul(np.clip(ne, 0, 1), -np.clip(efl, -np.inf, 0))
        node_out = node_out * (N['y'][0, :, 2:3] > 1e-3) / N['y'][0, :, 2:3]
        node_in = node_in * (N['y'][0, :, 1:2] > 1e-3) / N['y'][0, :, 1:2]
        y = np.concatenate([y[..., :1], node_in, node_out, y[..., 1:]], axis=-1)
    return y, ey


def predict_np(args, params, norms, states, b, a=None, edge_state=None):
    """`predict` (emulator.py:566-602): as `predict_tf` but through `post_proc` (NumPy mode) and the NumPy `constrain`."""
    c = config(args)
    dt = states.dtype
    T = lambda v: torch.as_tensor(np.asarray(v), dtype=dt)
    x = states[:, -c.seq_in:]
    ex = edge_state[:, -c.seq_in:]
    assert b.shape[1] == c.seq_out
    ae = get_edge_action(c, a) if c.act else None
    y, ey = forward(args, params, normalize(norms, x, 'x'), normalize(norms, b, 'b'), normalize(norms, ex, 'e'), ae)
    y, ey = post_proc_np(args, norms, y.numpy(), ey.numpy(), None if a is None else a.numpy(), normalize(norms, b, 'b').numpy())
    y, ey = T(y), T(ey)
    y = normalize(norms, y, 'y', True)
    ey = normalize(norms, ey, 'e', True)
    ey = torch.cat([torch.minimum(ey[..., 0].clamp(min=0), T(c.ehmax)).unsqueeze(-1), ey[..., 1:]], dim=-1)   # :587
    if c.pump_in.sum() + c.pump_out.sum() + c.pump.sum() > 0:     # :590-598
        ps = ((T(c.area) * (T(c.node_edge).clamp(0, 1) @ T(c.pump))) > 0).to(dt)
        h, qin, qout = y[..., 0], y[..., 1], y[..., 2]
        de = []
        for t in range(c.seq_out):
            prev = x[:, -1, :, 0] if t == 0 else de[-1] + (qin - qout)[:, t] / (T(c.area) + 1e-6)
            de.append(torch.minimum(torch.maximum(prev, T(c.hmin)), T(c.hmax)))
        y = torch.cat([(h * (1 - ps) + torch.stack(de, dim=1) * ps).unsqueeze(-1), y[..., 1:]], dim=-1)
    q_w, y = constrain(args, y, b[..., :1])                       # (`constrain` and `constrain_tf` agree, :727-770)
    return torch.cat([y, q_w.unsqueeze(-1)], dim=-1), ey


def simulate(args, params, norms, states, runoff, a=None, edge_states=None):
    """`simulate` (emulator.py:521-564): the reference's per-time-step loop -- one un-batched forward per window,
    NumPy-mode post-processing -- restated as that loop (the product batches the windows into one forward)."""
    c = config(args)
    preds, edge_preds = [], []
    for idx in range(runoff.shape[0]):                            # :528
        y, ey = predict_np(args, params, norms, states[idx:idx + 1], runoff[idx:idx + 1, :c.seq_out],
                           None if a is None else a[idx:idx + 1], edge_states[idx:idx + 1])
        preds.append(y[0])
        edge_preds.append(ey[0])
    return torch.stack(preds), torch.stack(edge_preds)


def model_rollout(args, params, norms, x, a, b, ex):
    """`_model` with roll > 0 (emulator.py:401-425) on NORMALISED tensors: `roll` chunks of seq_out steps, each fed
    with the previous chunk's (post-processed) prediction; the flood bit is thresholded at 0.5."""
    c = config(args)
    dt = x.dtype
    ys, eys = [], []
    for i in range(c.roll):
        sl = slice(i * c.seq_out, (i + 1) * c.seq_out)
        ae = get_edge_action(c, a[:, sl]) if c.act else None
        y, ey = forward(args, params, x[:, -c.seq_in:], b[:, sl], ex[:, -c.seq_in:], ae)
        y, ey = post_proc(args, norms, y, ey, a[:, sl] if a is not None else None, b[:, sl])
        ys.append(y)
        eys.append(ey)
        if c.if_flood:
            x_new = torch.cat([y[..., :-1], (y[..., -1:] > 0.5).to(dt), b[:, sl]], dim=-1)      # :417
        else:
            x_new = torch.cat([y, b[:, sl]], dim=-1)
        x = torch.cat([x[:, -(c.seq_in - c.seq_out):], x_new], dim=1) if c.seq_in > c.seq_out else x_new
        ae_new = get_edge_action(c, a[:, sl]) if c.act else torch.ones(ey.shape[:-1] + (1,), dtype=dt)
        ex_new = torch.cat([ey, ae_new], dim=-1)                                                  # :422
        ex = torch.cat([ex[:, -(c.seq_in - c.seq_out):], ex_new], dim=1) if c.seq_in > c.seq_out else ex_new
    return torch.cat(ys, dim=1).clamp(0, 1), torch.cat(eys, dim=1)                               # :437


def mbrl_rollout(args, params, norms, policy, x, a, b, y, ex, n_step, r_step, node_attrs, link_attrs):
    """`rollout` of mbrl.py:304-347 for a graph agent, restated step by step: observation (time sum of the cumulative / volume
    channels, last step of the others, on normalised states) -> policy -> settings held r_step steps -> `predict` (predict_tf)
    -> feedback with the flood bit thresholded at 0.5.  `policy(obs_list) -> (B, n_act)` is any function of the observation."""
    c = config(args)
    obs_of = lambda dat, attrs: torch.stack([dat[..., i].sum(dim=1) if ('cum' in at or '_vol' in at) else dat[:, -1, :, i]
                                             for i, at in enumerate(attrs)], dim=-1)
    xs, exs, settings, perfs = [x], [ex], [a[:, :c.seq_in]], [y[:, :c.seq_in, :, -1:]]
    for i in range(n_step):
        bi = b[:, i * r_step:(i + 1) * r_step]
        obs = [obs_of(normalize(norms, x, 'x'), node_attrs), obs_of(normalize(norms, ex, 'e'), link_attrs)]
        setting = policy(obs).unsqueeze(1).expand(-1, r_step, -1)
        settings.append(setting)
        preds, edge_preds = predict(args, params, norms, x, bi, setting, ex)
        if c.if_flood:
            x = torch.cat([preds[..., :-2], (preds[..., -2:-1] > 0.5).to(preds.dtype), bi], dim=-1)
        else:
            x = torch.cat([preds[..., :-1], bi], dim=-1)
        ex = torch.cat([edge_preds, get_edge_action(c, setting)], dim=-1)
        xs.append(x)
        exs.append(ex)
        perfs.append(preds[..., -1:])
    return [torch.cat(t, dim=1) for t in (xs, exs, settings, perfs)]


def convnet_forward(args, params, X, E, B=None):
    """`ConvNet.build_network` of the RL agents (agent.py:65-99): Dense embeddings, the spatial block, then Spektral's
    GlobalAttnSumPool (softmax over the stacked node + link rows of x @ attn_kernel, weighted sum) -> (batch, conv_dim).
    params: {'embed_x', 'embed_e', 'block': [layers as in init_params], 'pool': {'attn_kernel' (F,1)}}."""
    c = config(args)
    dt = X.dtype
    a = getattr(args, 'activation', None) or 'linear'
    if getattr(args, 'use_pred', False):
        X = torch.cat([X, B], dim=-1)
    x = OD.dense(X, params['embed_x']['kernel'], params['embed_x']['bias'], a)
    e = OD.dense(E, params['embed_e']['kernel'], params['embed_e']['bias'], a)
    for p in params['block']:
        if c.graph_base:
            q = p['gat']
            z = OD.gat_conv_dense(torch.cat([x, e], dim=-2), torch.from_numpy(c.filter).to(dt), q['kernel'], q['attn_kernel_self'],
                                  q['attn_kernel_neighs'], q['bias'], a)
            x, e = z[:, :c.n_node], z[:, c.n_node:]
        else:
            x, e = _spatial_layer(x, e, p, c, dt)
    z = torch.cat([x, e], dim=-2)
    alpha = torch.softmax((z @ params['pool']['attn_kernel']).squeeze(-1), dim=-1)
    return (alpha.unsqueeze(-2) @ z).squeeze(-2)


def mpc_objective(args, params, norms, y, state, runoff, edge_state, n_step, n_act, r_step, targets, gamma=None):
    """Objective of `mpc_problem_gr.objective_fn` (mpc.py:551-598) with the astlingen objective (astlingen.py:75-99) for a
    population y (pop, n_step*n_act): settings held r_step steps, predict chunk by chunk, weighted flooding + treatment-plant
    inflow + inflow roughness, summed over the horizon."""
    c = config(args)
    pop = y.shape[0]
    s = y.reshape(-1, n_step, n_act).repeat_interleave(r_step, dim=1)
    T = runoff.shape[0]
    if s.shape[1] < T:
        s = torch.cat([s, s[:, -1:].expand(-1, T - s.shape[1], -1)], dim=1)
    s = s[:, :T]
    rep = lambda t: t.unsqueeze(0).expand((pop,) + tuple(t.shape))
    st0, ro, es = rep(state), rep(runoff), rep(edge_state)
    n_chunk = T // c.seq_out
    if n_chunk <= 1:
        preds, _ = predict(args, params, norms, st0, ro, s, es)
    else:
        st, ed, ys, perf = st0[:, -c.seq_in:], es[:, -c.seq_in:], [], None
        for i in range(n_chunk):
            sl = slice(i * c.seq_out, (i + 1) * c.seq_out)
            if c.if_flood and i > 0:
                st = torch.cat([st[..., :-1], (perf > 0).to(y.dtype), st[..., -1:]], dim=-1)
            yy, ey = predict(args, params, norms, st, ro[:, sl], s[:, sl], ed)
            st, perf = torch.cat([yy[..., :-2], ro[:, sl]], dim=-1), yy[..., -1:]
            ed = torch.cat([ey, get_edge_action(c, s[:, sl])], dim=-1)
            ys.append(yy)
        preds = torch.cat(ys, dim=1)
    q_w = preds[..., -1]
    q_in = torch.cat([st0[:, -1:, :, 1], preds[..., 1]], dim=1)
    obj = (q_w[..., targets['flood_idx']] * targets['flood_w']).sum(-1)
    if targets.get('outflow_idx') is not None:
        obj = obj + (q_in[:, 1:, targets['outflow_idx']] * targets['outflow_w']).sum(-1)
    if targets.get('smooth_idx') is not None:
        obj = obj + ((q_in[:, 1:, targets['smooth_idx']] - q_in[:, :-1, targets['smooth_idx']]).abs() * targets['smooth_w']).sum(-1)
    if gamma is not None:
        obj = obj * gamma
    return obj.sum(-1)
