"""Shared helpers for the tests: seeded parameters in the oracle's (Keras-shaped) format."""
import inspect
import os

import numpy as np
import torch

# (test name, observed max abs error, allowed) of every close() call of the session: tests/conftest.py prints the largest
# observed / allowed ratios when UDS_TOL_REPORT is set, so that tolerances can be kept a stated factor above what the
# kernels actually achieve instead of orders of magnitude above it
OBSERVED = []


def close(out, ref, tol, _depth=1):
    """max|out - ref| <= tol * max(1, max|ref|); returns the observed error."""
    out = out.detach().double().cpu()
    assert out.shape == ref.shape, (out.shape, ref.shape)
    err = float((out - ref).abs().max()) if ref.numel() else 0.0
    lim = tol * max(1.0, float(ref.abs().max()) if ref.numel() else 1.0)
    if os.environ.get('UDS_TOL_REPORT'):
        fr = inspect.stack()[_depth]
        OBSERVED.append((os.environ.get('PYTEST_CURRENT_TEST', fr.function).split(' ')[0], fr.lineno, err, lim))
    assert err <= lim, 'max abs err %.3e > %.3e' % (err, lim)
    return err


def glorot(gen, shape, dtype):
    if len(shape) == 2:
        fan_in, fan_out = shape
    else:
        rf = int(np.prod(shape[:-2]))
        fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
    lim = (6.0 / (fan_in + fan_out)) ** 0.5
    return ((torch.rand(shape, generator=gen, dtype=torch.float64) * 2 - 1) * lim).to(dtype)


def spatial_params(n_node, n_edge, fx, fe, d, seed=1, dtype=torch.float64, bias_scale=0.1, dense_ne=True,
                   nnz_n=None, nnz_e=None):
    """Random parameters of one spatial layer (`emulator.py:225-230`), Keras shapes.  Biases are
    non-zero (trained-like) so bias handling is exercised."""
    g = torch.Generator().manual_seed(seed)
    h = d // 2
    rn = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64).to(dtype)
    p = {
        'xe_k': glorot(g, (fe, h), dtype), 'xe_b': rn(h) * bias_scale,
        'ex_k': glorot(g, (fx, h), dtype), 'ex_b': rn(h) * bias_scale,
        'gx_k': glorot(g, (fx + h, 1, d), dtype), 'gx_as': glorot(g, (d, 1, 1), dtype),
        'gx_an': glorot(g, (d, 1, 1), dtype), 'gx_b': rn(d) * bias_scale,
        'ge_k': glorot(g, (fe + h, 1, d), dtype), 'ge_as': glorot(g, (d, 1, 1), dtype),
        'ge_an': glorot(g, (d, 1, 1), dtype), 'ge_b': rn(d) * bias_scale,
    }
    if dense_ne:
        p.update({'ne_n_w': rn(n_node, n_edge) * 0.05, 'ne_n_b': torch.zeros(n_node, n_edge, dtype=dtype),
                  'ne_e_w': rn(n_edge, n_node) * 0.05, 'ne_e_b': torch.zeros(n_edge, n_node, dtype=dtype)})
    else:
        p.update({'ne_n_v': rn(nnz_n) * 0.05 + 0.3, 'ne_e_v': rn(nnz_e) * 0.05 + 0.3})
    return p


def cast(p, dtype):
    return {k: (v.to(dtype) if isinstance(v, torch.Tensor) else v) for k, v in p.items()}


def load_spatial_layer(layer, p, device):
    """Copy oracle-format parameters into a gnn_uds_amd.layers.SpatialLayer."""
    f32 = lambda t: t.to(torch.float32).to(device).contiguous()
    layer.to(device)
    layer.dense_xe.kernel.data = f32(p['xe_k']); layer.dense_xe.bias.data = f32(p['xe_b'])
    layer.dense_ex.kernel.data = f32(p['ex_k']); layer.dense_ex.bias.data = f32(p['ex_b'])
    if 'ne_n_w' in p:
        layer.node_edge_n.weight.data = f32(p['ne_n_w']); layer.node_edge_n.bias.data = f32(p['ne_n_b'])
        layer.node_edge_e.weight.data = f32(p['ne_e_w']); layer.node_edge_e.bias.data = f32(p['ne_e_b'])
    else:
        layer.node_edge_n.weight.data = f32(p['ne_n_v']); layer.node_edge_n.bias.data = torch.zeros_like(layer.node_edge_n.weight.data)
        layer.node_edge_e.weight.data = f32(p['ne_e_v']); layer.node_edge_e.bias.data = torch.zeros_like(layer.node_edge_e.weight.data)
    layer.gat_x.kernel.data = f32(p['gx_k']); layer.gat_x.attn_kernel_self.data = f32(p['gx_as'])
    layer.gat_x.attn_kernel_neighs.data = f32(p['gx_an']); layer.gat_x.bias.data = f32(p['gx_b'])
    layer.gat_e.kernel.data = f32(p['ge_k']); layer.gat_e.attn_kernel_self.data = f32(p['ge_as'])
    layer.gat_e.attn_kernel_neighs.data = f32(p['ge_an']); layer.gat_e.bias.data = f32(p['ge_b'])
    return layer


# ---------------------------------------------------------------------------------------------------------------
# whole-emulator helpers
# ---------------------------------------------------------------------------------------------------------------
def emulator_args(edges, n_node, seed=0, **over):
    """A reference-style `args` namespace (attribute names of Emulator.__init__, emulator.py:48-127) for a link list."""
    from types import SimpleNamespace
    from oracle import graphs as OG
    edges = np.asarray(edges)
    rng = np.random.default_rng(seed)
    n_edge = len(edges)
    a = dict(state_shape=(n_node, 4), edge_state_shape=(n_edge, 4), seq_in=5, seq_out=5, embed_size=64, hidden_dim=64,
             kernel_size=3, n_sp_layer=2, n_tp_layer=2, activation='relu', if_flood=3, epsilon=-1.0, edge_fusion=True,
             edges=edges, adj=over['adj'] if 'adj' in over else OG.adjacency(edges),
             edge_adj=over['edge_adj'] if 'edge_adj' in over else OG.edge_adjacency(edges),      # (networkx refuses chaohu's parallel link)
             node_edge=OG.node_edge_incidence(n_node, edges),
             act=True, act_edges=edges[[1, 4]], conv='GAT', resnet=True, recurrent='Conv1D', roll=0, model_dir='/tmp/uds_model',
             is_outfall=(np.arange(n_node) == 0).astype(float), hmax=1.0 + rng.random(n_node), hmin=np.zeros(n_node),
             area=np.zeros(n_node), pump=np.zeros(n_edge), pump_in=np.zeros(n_node), pump_out=np.zeros(n_node),
             offset=np.zeros(n_edge), ehmax=0.3 + rng.random(n_edge), tide=False)
    a.update(over)
    if a.get('graph_base'):                  # the reference passes the combined (N+E)^2 adjacency as `adj` (base.py:320-323)
        build = OG.node_based_adjacency if a['graph_base'] == 1 else OG.edge_based_adjacency
        a['adj'] = build(edges, a.get('directed', False), a.get('order', 1))
    return SimpleNamespace(**a)


def emulator_norms(args, seed=0, dtype=torch.float64):
    """[max, min] per node / link and channel (dataloader.py:224-266 layout): min = 0, max random positive."""
    g = torch.Generator().manual_seed(seed)
    n, e = args.state_shape[0], args.edge_state_shape[0]
    n_in = args.state_shape[1] + (1 if args.if_flood else 0)
    mk = lambda rows, c: torch.stack([0.5 + torch.rand(rows, c, generator=g, dtype=torch.float64),
                                      torch.zeros(rows, c, dtype=torch.float64)]).to(dtype)
    return {'x': mk(n, n_in), 'b': mk(n, 2 if args.tide else 1), 'y': mk(n, 5), 'r': mk(n, 1), 'e': mk(e, 4)}


def load_emulator(emul, p, device):
    """Copy oracle.emulator_ref.init_params parameters into a gnn_uds_amd.Emulator."""
    f32 = lambda t: t.to(torch.float32).to(device).contiguous()
    emul.to(device)

    def dense(m, q):
        m.kernel.data, m.bias.data = f32(q['kernel']), f32(q['bias'])

    def spatial(block, layers):
        if not emul.conv:                           # conv=False: the blocks are lists of Dense(2 d) layers
            for layer, q in zip(block, layers):
                dense(layer, q)
            return
        for layer, q in zip(block.layers, layers):
            if 'gat' in q:                      # graph_base: one conv over the stacked node + link rows
                if 'theta' in q['gat']:
                    layer.kernel.data = f32(q['gat']['theta'])
                    continue
                layer.kernel.data = f32(q['gat']['kernel'])
                layer.attn_kernel_self.data, layer.attn_kernel_neighs.data = f32(q['gat']['attn_kernel_self']), f32(q['gat']['attn_kernel_neighs'])
                layer.bias.data = f32(q['gat']['bias'])
                continue
            dense(layer.dense_xe, q['dense_xe']); dense(layer.dense_ex, q['dense_ex'])
            for ne, key in ((layer.node_edge_n, 'node_edge_n'), (layer.node_edge_e, 'node_edge_e')):
                ne.weight.data, ne.bias.data = f32(q[key]['weight']), f32(q[key]['bias'])
            convs = (layer.gat_x, layer.gat_e) if layer.conv == 'GAT' else (layer.gcn_x, layer.gcn_e)
            for m, key in zip(convs, ('gat_x', 'gat_e')):
                if layer.conv == 'Diffusion':
                    m.kernel.data = f32(q[key]['theta'])
                    continue
                if layer.conv == 'GAT':
                    m.kernel.data = f32(q[key]['kernel'])
                    m.attn_kernel_self.data, m.attn_kernel_neighs.data = f32(q[key]['attn_kernel_self']), f32(q[key]['attn_kernel_neighs'])
                else:
                    m.kernel.data = f32(q[key]['kernel'][:, 0, :])
                m.bias.data = f32(q[key]['bias'])

    dense(emul.embed_x, p['embed_x']); dense(emul.embed_b, p['embed_b']); dense(emul.embed_e, p['embed_e'])
    if emul.act:
        dense(emul.embed_ae, p['embed_ae'])
    spatial(emul.block1, p['block1']); spatial(emul.block2, p['block2'])
    for mods, key in ((emul.tem1_x, 'tem1_x'), (emul.tem1_e, 'tem1_e'), (emul.tem2_x, 'tem2_x'), (emul.tem2_e, 'tem2_e')):
        assert len(mods) == len(p[key])
        for m, q in zip(mods, p[key]):
            dense(m, q)
            if 'recurrent_kernel' in q:
                m.recurrent_kernel.data = f32(q['recurrent_kernel'])
    dense(emul.res_x, p['res_x']); dense(emul.res_e, p['res_e']); dense(emul.out, p['out'])
    for m, q in zip(emul.flood, p['flood']):
        dense(m, q)
    if emul.if_flood:
        dense(emul.flood_out, p['flood_out'])
    dense(emul.e_out_layer, p['e_out'])
    return emul


def emulator_param_pairs(emul, flat):
    """(name, module parameter, tensor) for every entry of `flat` -- a name -> tensor dict keyed like
    oracle.train_ref.tree_leaves(oracle params) ('block1.0.gat_x.kernel', 'tem1_x.1.bias', 'e_out.kernel', ...)."""
    out = []
    for name, t in flat.items():
        parts = name.split('.')
        m = emul
        for i, part in enumerate(parts[:-1]):
            if part == 'e_out':
                m = m.e_out_layer
            elif part in ('block1', 'block2'):
                m = getattr(m, part).layers if emul.conv else getattr(m, part)      # conv = False: a plain list of Dense layers
            elif part.isdigit():
                m = m[int(part)]
            elif part == 'gat' and not hasattr(m, 'gat'):
                pass                              # graph_base: the layer IS the conv
            elif part in ('gat_x', 'gat_e') and not hasattr(m, part):
                m = getattr(m, part.replace('gat', 'gcn'))      # conv = GCN: the oracle keeps the key, the module is gcn_x / gcn_e
            else:
                m = getattr(m, part)
        if parts[-1].startswith('attn_kernel') and not hasattr(m, parts[-1]):
            continue                              # conv = GCN: the oracle's parameter tree keeps the (unused) attention vectors
        out.append((name, getattr(m, parts[-1]), t))
    return out
