"""Keras checkpoint name map (Emulator.keras_layer_map / load_keras_weights, SURVEY.md Appendix B + 8f rank 1): host
logic only, CPU parameters -- creation-order names, shape checks, round trip through the Keras-keyed dict."""
import json
import os

import numpy as np
import pytest
import torch

import gnn_uds_amd as U
from tests.util import emulator_args

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _emul(**over):
    with open(os.path.join(ROOT, 'tests', 'golden', 'networks.json')) as fh:
        net = json.load(fh)['astlingen']
    args = emulator_args(np.array(net['edges']), net['n_node'], **over)
    return U.Emulator(args.conv, args.resnet, args.recurrent, args, generator=torch.Generator().manual_seed(3))


def test_creation_order_names():
    names = [n for n, _, _ in _emul(n_sp_layer=2, n_tp_layer=2, if_flood=3).keras_layer_map()]
    assert names[:4] == ['dense', 'dense_1', 'dense_2', 'dense_3']                       # X, B, E, AE embeddings (emulator.py:198-212)
    assert names[4:10] == ['dense_4', 'dense_5', 'node_edge', 'node_edge_1', 'mixed_gat', 'mixed_gat_1']
    assert names[10:16] == ['dense_6', 'dense_7', 'node_edge_2', 'node_edge_3', 'mixed_gat_2', 'mixed_gat_3']
    assert names[16:20] == ['conv1d', 'conv1d_1', 'conv1d_2', 'conv1d_3']                # n_tp for x, then n_tp for e
    assert names[20:26] == ['dense_8', 'dense_9', 'node_edge_4', 'node_edge_5', 'mixed_gat_4', 'mixed_gat_5']
    assert names[32:36] == ['conv1d_4', 'conv1d_5', 'conv1d_6', 'conv1d_7']
    assert names[36:] == ['dense_resx', 'dense_12', 'dense_13', 'dense_14', 'dense_15', 'dense_16', 'dense_17', 'dense_18']
    no_act = [n for n, _, _ in _emul(act=False, if_flood=0, n_sp_layer=1, n_tp_layer=1).keras_layer_map()]
    # embeddings X, B, E (no AE), one spatial layer, 1+1 conv, spatial, 1+1 conv, dense_resx, e-res, out, e_out
    assert no_act == ['dense', 'dense_1', 'dense_2', 'dense_3', 'dense_4', 'node_edge', 'node_edge_1', 'mixed_gat', 'mixed_gat_1',
                      'conv1d', 'conv1d_1', 'dense_5', 'dense_6', 'node_edge_2', 'node_edge_3', 'mixed_gat_2', 'mixed_gat_3',
                      'conv1d_2', 'conv1d_3', 'dense_resx', 'dense_7', 'dense_8', 'dense_9']


def test_round_trip_and_shape_check():
    a, b = _emul(), _emul()
    for p in b.parameters():
        p.data.zero_()
    w = a.export_keras_weights()
    assert w['mixed_gat/kernel:0'].shape == (96, 1, 64) and w['node_edge/weight:0'].shape == (a.n_node, a.n_edge)
    assert w['conv1d/kernel:0'].shape == (3, 64, 64) and w['dense_resx/bias:0'].shape == (64,)
    b.load_keras_weights(w)
    for (n1, p1), (n2, p2) in zip(a.named_parameters(), b.named_parameters()):
        assert n1 == n2 and torch.equal(p1, p2)
    # HDF5 group-path keys and per-layer lists are accepted too
    h5 = {k.split('/')[0] + '/' + k: v for k, v in w.items()}
    c = _emul()
    c.load_keras_weights(h5)
    assert torch.equal(c.block2.layers[0].gat_e.attn_kernel_neighs, a.block2.layers[0].gat_e.attn_kernel_neighs)
    bad = dict(w)
    bad['dense_1/kernel:0'] = np.zeros((2, 32), dtype=np.float32)
    with pytest.raises(ValueError):
        _emul().load_keras_weights(bad)
    del bad['dense_1/kernel:0']
    with pytest.raises(KeyError):
        _emul().load_keras_weights(bad)


def test_dense_bias_is_not_dropped_silently():
    """The reference trains NodeEdge.b as a full (R, M) matrix (emulator.py:36-45): a model with one parameter per support
    entry takes such a checkpoint only when the bias is zero off the support, and refuses it otherwise."""
    a = _emul()
    w = a.export_keras_weights()
    sp = _emul(sparse_params=True)
    sp.load_keras_weights(w)                       # fresh model: bias is all zero -> fine
    ne = sp.block1.layers[0].node_edge_n
    assert ne.sparse and torch.equal(ne.weight, a.block1.layers[0].node_edge_n.weight.reshape(-1)[ne._flat])
    w = dict(w)
    b = np.array(w['node_edge/bias:0'])
    flat = np.asarray(ne._flat)
    off = np.setdiff1d(np.arange(b.size), flat)[0]
    b.reshape(-1)[flat] = 0.25                     # on the support: kept
    w['node_edge/bias:0'] = b
    sp.load_keras_weights(w)
    assert float(ne.bias.min()) == 0.25
    b = b.copy()
    b.reshape(-1)[off] = 1e-3                      # one trained entry off the support
    w['node_edge/bias:0'] = b
    with pytest.raises(ValueError, match='off the incidence support'):
        sp.load_keras_weights(w)
    _emul().load_keras_weights(w)                  # the dense model keeps it


def test_get_adj_action_matches_the_dense_rewrite_on_a_weighted_adjacency():
    """`get_adj_action` (emulator.py:343-362) as an edge mask over the CSR entries equals the reference's dense rewrite + int cast,
    also when the adjacency carries weights (entries below 1 vanish in the cast whether actuated or not -- a reference quirk
    that is kept).  Host logic only: no GPU needed."""
    from oracle import emulator_ref as OE
    with open(os.path.join(ROOT, 'tests', 'golden', 'networks.json')) as fh:
        net = json.load(fh)['astlingen']
    args = emulator_args(np.array(net['edges']), net['n_node'], use_adj=True)
    rng = np.random.default_rng(0)
    args.adj = np.array(args.adj, dtype=float) * (0.5 + 1.5 * rng.random(np.shape(args.adj)))
    em = U.Emulator(args.conv, args.resnet, args.recurrent, args)
    a = torch.rand(2, 3, len(args.act_edges), generator=torch.Generator().manual_seed(1)) * 2
    mask = em.get_adj_action(a)
    dense = OE.get_adj_action(OE.config(args), a)
    csr, _, pos = em._adj_pattern()
    rows, cols = np.repeat(np.arange(csr.n_rows), np.diff(csr.rowptr)), np.asarray(csr.col)
    off = torch.as_tensor(rows != cols)                          # the diagonal is forced to one afterwards (set_diag)
    want = (dense[..., rows, cols] != 0).float()
    assert torch.equal(mask[..., off], want[..., off]) and 0.2 < float(mask[..., off].mean()) < 0.9
    assert (pos >= 0).all()                                      # every actuated (from, to) pair is an entry of the pattern
    # the NumPy-mode form (g=False: the setting is WRITTEN into the entry, emulator.py:350-354; `predict` / `simulate`): on a
    # weighted adjacency an entry below 1 survives whenever the setting is >= 1, where the product form truncates it away
    mask_np = em.get_adj_action(a, False)
    want_np = (OE.get_adj_action(OE.config(args), a, g=False)[..., rows, cols] != 0).float()
    assert torch.equal(mask_np[..., off], want_np[..., off])
    assert not torch.equal(mask_np, mask)                        # the two forms do differ here


def test_keras_numbers_the_recurrent_cells_and_splits_diffusion_kernels(tmp_path):
    """Names a real Keras 2.10 `model.h5` uses and the builder's own exporter did not produce before: the k-th GRU layer's cell
    is `gru_cell_k` (`gru_1/gru_1/gru_cell_1/kernel:0`), whatever number the session had reached; DiffusionConv keeps one
    `diffuse_features_k/kernel:0` of (K + 1,) per channel.  Both go through the HDF5 writer / reader pair."""
    from tests import h5_writer as W
    a = _emul(recurrent='GRU', embed_size=32, hidden_dim=16, n_sp_layer=1, n_tp_layer=2, if_flood=0)
    names = [n for n, m, _ in a.keras_layer_map() if n.startswith('gru')]
    assert names == ['gru'] + ['gru_%d' % k for k in range(1, 8)]                     # 4 stacks x 2 layers
    w = a.export_keras_weights()
    assert 'gru/gru_cell/kernel:0' in w and 'gru_3/gru_cell_3/recurrent_kernel:0' in w and 'gru_7/gru_cell_7/bias:0' in w
    # a session that had created other cells before: every cell number shifted, the layers' own numbers not
    shifted = {}
    for k, v in w.items():
        parts = k.split('/')
        if len(parts) == 3 and parts[1].startswith('gru_cell'):
            n = int(parts[1].rsplit('_', 1)[1]) if parts[1] != 'gru_cell' else 0
            parts[1] = 'gru_cell_%d' % (n + 11)
        shifted['/'.join(parts)] = v
    W.write_tree(str(tmp_path / 'model.h5'), W.keras_tree(shifted), nodes=3)
    b = _emul(recurrent='GRU', embed_size=32, hidden_dim=16, n_sp_layer=1, n_tp_layer=2, if_flood=0)
    for q in b.parameters():
        q.data.zero_()
    b.load(str(tmp_path / 'model.h5'))
    for (n1, p1), (n2, p2) in zip(a.named_parameters(), b.named_parameters()):
        assert n1 == n2 and torch.equal(p1, p2), n1
    # DiffusionConv: per-channel sub-layer kernels, stacked in numeric (not lexicographic) order
    c = _emul(conv='Diffusion', embed_size=16, hidden_dim=16, n_sp_layer=1, n_tp_layer=1, if_flood=0, act=False)
    w = c.export_keras_weights()
    split = {}
    for k, v in w.items():
        if k.startswith('diffusion_conv') and k.endswith('/kernel:0'):
            for ch in range(v.shape[0]):
                split['%s/diffuse_features%s/kernel:0' % (k.split('/')[0], '' if ch == 0 else '_%d' % ch)] = np.asarray(v[ch])
        else:
            split[k] = v
    assert any('diffuse_features_11/' in k for k in split)                           # 16 channels: _10.. sort after _9, not after _1
    d = _emul(conv='Diffusion', embed_size=16, hidden_dim=16, n_sp_layer=1, n_tp_layer=1, if_flood=0, act=False)
    for q in d.parameters():
        q.data.zero_()
    d.load_keras_weights(split)
    for (n1, p1), (n2, p2) in zip(c.named_parameters(), d.named_parameters()):
        assert n1 == n2 and torch.equal(p1, p2), n1


def test_from_node_gather_equals_the_dense_incidence_product():
    """`v @ clip(node_edge, 0, 1)` (offset gate, rated pumps, pumped-storage depth: emulator.py:630-638,649,657,689,699) is a
    gather at the from-nodes: the same numbers from the incidence CSR on all five shipped networks (chaohu has a parallel
    link), so the gates also run for CSR-only networks, which have no dense (N, E) matrix to multiply by."""
    from oracle import graphs as OG
    with open(os.path.join(ROOT, 'tests', 'golden', 'networks.json')) as fh:
        nets = json.load(fh)
    for name, net in nets.items():
        edges, n = np.array(net['edges']), net['n_node']
        args = emulator_args(edges, n, n_sp_layer=1, n_tp_layer=1, edge_adj=np.eye(len(edges)))     # (the line graph is not needed here)
        em = U.Emulator(args.conv, args.resnet, args.recurrent, args)
        v = torch.rand(3, 2, n, generator=torch.Generator().manual_seed(1))
        pos = torch.as_tensor(OG.node_edge_incidence(n, edges), dtype=torch.float32).clamp(0, 1)
        assert torch.equal(em._at_from_node(v), torch.matmul(v, pos)), name
        idx, ok = em._from_node_index(v.device)
        pump = torch.rand(len(edges), generator=torch.Generator().manual_seed(2))
        assert torch.allclose(torch.zeros(n).index_add_(0, idx, pump * ok), torch.mv(pos, pump), atol=1e-6), name
