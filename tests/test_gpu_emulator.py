"""GPU parity of the whole surrogate (`Emulator`) against the fp64 CPU oracle: network forward, predict_tf with its
post-processing, autoregressive rollout, plus the three non-graph kernels it adds (causal Conv1D, resnet prefix sum,
link->node flow balance).  The oracle itself is "parity unpinned" (oracle/__init__.py).

Tolerances (relative to max(1, max|ref|); 3-6x above the largest error MEASURED over this file, UDS_TOL_REPORT=1):
  single exact-fp32 kernels 5e-6 (prefix sums over T steps: x sqrt(T)); single split-bf16 row-GEMM kernels 1e-4
  (measured <= 2.7e-5); whole forward 2e-5 with split-bf16 layers (measured <= 5e-6), 5e-6 exact fp32 (measured 6e-7)."""
TOL_FWD = {'fp32': 5e-6, 'bf16x3': 2e-5}      # whole Emulator forward
TOL_ROWGEMM = 1e-4
OUTLIERS_ALLOWED = 0       # entries beyond the tolerance in predict_tf / predict / simulate / short rollouts: none (round 2 allowed numel / 500)
ROLL100_TOL = 5e-5        # every one of 100 fed-back steps, split-bf16 layers (measured: worst step 1.2e-5, last step 6.2e-6, 9 states within 1e-5 of the threshold, no bit flipped)
import numpy as np
import pytest
import torch

import gnn_uds_amd as U
from gnn_uds_amd import _lib
from oracle import emulator_ref as OE
from oracle import spektral_dense as OD
from tests.util import close, emulator_args, emulator_norms, load_emulator

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda', 0)




def rnd(g, *shape):
    return torch.rand(*shape, generator=g, dtype=torch.float64)


@pytest.mark.parametrize('B,T,R,F,H,dil,act', [(2, 5, 7, 64, 64, 1, 'relu'), (1, 60, 3, 96, 64, 4, 'relu'), (3, 4, 5, 10, 6, 2, 'tanh')])
def test_conv1d_causal(dev, B, T, R, F, H, dil, act):
    g = torch.Generator().manual_seed(T)
    x, k, b = rnd(g, B, T, R, F) - 0.5, rnd(g, 3, F, H) - 0.5, rnd(g, H) - 0.5
    ref = OE.conv1d_causal(x.permute(0, 2, 1, 3).reshape(B * R, T, F), k, b, dil, act).reshape(B, R, T, H).permute(0, 2, 1, 3)
    f = lambda t: t.float().to(dev)
    close(_lib.conv1d_causal(f(x), f(k), f(b), dil, act), ref, 5e-6)


@pytest.mark.parametrize('B,T,R,F,H,taps,dil,act', [(2, 5, 7, 64, 64, 3, 1, 'relu'), (1, 60, 3, 96, 64, 3, 4, 'relu'), (3, 4, 50, 64, 32, 3, 2, 'tanh'),
                                                    (1, 1, 1000, 64, 64, 1, 1, 'linear'), (1, 1, 333, 32, 1, 1, 1, 'sigmoid'), (2, 3, 40, 64, 3, 1, 1, 'tanh'),
                                                    (1, 1, 97, 64, 32, 1, 1, 'relu'),
                                                    # slabs wide enough for the XCD-aware row mapping (R >= 4096), ragged XCD ranges
                                                    (1, 6, 4133, 64, 64, 3, 2, 'relu'), (2, 3, 4096, 32, 32, 3, 1, 'linear'),
                                                    # time-streaming Conv1D (taps 3, 64 -> 64, >= 256 sixteen-row streams): every dilation,
                                                    # T not a multiple of the 2D+1 ring, several time segments, a ragged last block
                                                    (1, 13, 4100, 64, 64, 3, 1, 'relu'), (1, 29, 4100, 64, 64, 3, 4, 'tanh'), (2, 60, 2050, 64, 64, 3, 2, 'relu'),
                                                    (1, 60, 4099, 64, 64, 3, 4, 'linear'), (1, 2, 4096, 64, 64, 3, 4, 'relu'),
                                                    # 3 x 128 -> 64 (first temporal layer of a d = 128 emulator): 96 KB of weights, ring of 2
                                                    (1, 7, 700, 128, 64, 3, 2, 'relu'), (2, 5, 33, 128, 32, 3, 1, 'tanh')])
def test_rowgemm_mfma_dense_and_conv(dev, B, T, R, F, H, taps, dil, act):
    """Matrix-core Dense / causal Conv1D (split-bf16, 3 products): tolerance 2e-4 * max(1, max|ref|)."""
    g = torch.Generator().manual_seed(T + R)
    x, k, b = rnd(g, B, T, R, F) - 0.5, rnd(g, taps, F, H) - 0.5, rnd(g, H) - 0.5
    if taps == 1:
        ref = OD.dense(x, k[0], b, act)
    else:
        ref = OE.conv1d_causal(x.permute(0, 2, 1, 3).reshape(B * R, T, F), k, b, dil, act).reshape(B, R, T, H).permute(0, 2, 1, 3)
    f = lambda t: t.float().to(dev)
    packed = _lib.rowgemm_pack(f(k).reshape(taps * F, H))
    close(_lib.rowgemm_forward(f(x), packed, f(b), H, act, taps=taps, dilation=dil), ref, TOL_ROWGEMM)
    close(_lib.rowgemm_forward(f(x), packed, None, H, 'linear', taps=taps, dilation=dil) if taps == 1 else
          _lib.rowgemm_forward(f(x), packed, None, H, 'linear', taps=taps, dilation=dil),
          OD.dense(x, k[0], None) if taps == 1 else
          OE.conv1d_causal(x.permute(0, 2, 1, 3).reshape(B * R, T, F), k, torch.zeros(H, dtype=torch.float64), dil, 'linear').reshape(B, R, T, H).permute(0, 2, 1, 3),
          2e-4)


@pytest.mark.parametrize('B,T,R,act,with_res', [(1, 7, 300, 'relu', True), (2, 13, 4100, 'relu', True), (1, 60, 33, 'tanh', False),
                                                (1, 1, 16, 'linear', True)])
def test_dense_cumsum_stream(dev, B, T, R, act, with_res):
    """Dense(64->64) + prefix sum over time + residual + activation in one kernel (split-bf16 MFMA): tolerance
    2e-4 * max(1, max|ref|) * sqrt(T) (the running sum accumulates T products in fp32)."""
    g = torch.Generator().manual_seed(T + R)
    x, k, b = rnd(g, B, T, R, 64) - 0.5, rnd(g, 64, 64) - 0.5, rnd(g, 64) - 0.5
    res = rnd(g, B, 1, R, 64) - 0.5 if with_res else None
    ref = torch.cumsum(x @ k + b, dim=1)
    ref = OD.activation(act)(ref + res if with_res else ref)
    f = lambda t: None if t is None else t.float().to(dev)
    out = _lib.dense_cumsum(f(x), _lib.rowgemm_pack(f(k)), f(b), f(res), act)
    close(out, ref, TOL_ROWGEMM * max(1.0, T ** 0.5))
    close(_lib.dense_cumsum(f(x), _lib.rowgemm_pack(f(k)), None, None, 'linear'), torch.cumsum(x @ k, dim=1), TOL_ROWGEMM * max(1.0, T ** 0.5))


def test_cumsum_act_and_flow_balance(dev, networks):
    g = torch.Generator().manual_seed(1)
    x, res = rnd(g, 2, 6, 9, 8) - 0.5, rnd(g, 2, 1, 9, 8) - 0.5
    close(_lib.cumsum_act(x.float().to(dev), res.float().to(dev), 'relu'), torch.relu(torch.cumsum(x, 1) + res), 1e-6)
    close(_lib.cumsum_act(x.float().to(dev), None, 'linear'), torch.cumsum(x, 1), 1e-6)
    net = networks['chaohu']
    gph = U.DrainageGraph.from_edges(np.array(net['edges']), net['n_node'])
    ne = torch.from_numpy(gph.inc_n.to_dense())
    flow = rnd(g, 4, gph.n_edge, 1) - 0.5
    s_in, s_out = rnd(g, gph.n_node), rnd(g, gph.n_node)
    pos, neg = ne.clamp(0, 1), ne.clamp(-1, 0).abs()
    fp, fn = flow.clamp(min=0), -flow.clamp(max=0)
    q_out, q_in = (pos @ fp + neg @ fn)[..., 0] * s_out, (neg @ fp + pos @ fn)[..., 0] * s_in
    h = _lib.CsrHandle(gph.inc_n)
    sign = torch.as_tensor(gph.inc_n.val, dtype=torch.float32, device=dev)
    gi, go = _lib.flow_balance(h, sign, flow[..., 0].float().to(dev).contiguous(), s_in.float().to(dev), s_out.float().to(dev))
    close(gi, q_in, 1e-6); close(go, q_out, 1e-6)


def _setup(networks, name, dev, precision='bf16x3', **over):
    net = networks[name]
    args = emulator_args(net['edges'], net['n_node'], **over)
    params = OE.init_params(args, seed=3)
    emul = load_emulator(U.Emulator(args.conv, args.resnet, args.recurrent, args, precision=precision), params, dev)
    norms = emulator_norms(args)
    emul.set_norm(*(norms[k].numpy() for k in 'xbyre'))
    return args, params, emul, norms


def _inputs(args, B, seed=5, n_act=2):
    g = torch.Generator().manual_seed(seed)
    N, E = args.state_shape[0], args.edge_state_shape[0]
    n_in = args.state_shape[1] + (1 if args.if_flood else 0)
    X = rnd(g, B, args.seq_in, N, n_in)
    Bd = rnd(g, B, args.seq_out * max(1, args.roll), N, 2 if args.tide else 1) * 0.1
    Ex = rnd(g, B, args.seq_in, E, 4)
    a = rnd(g, B, args.seq_out * max(1, args.roll), n_act)
    return X, Bd, Ex, a


@pytest.mark.parametrize('precision,tol', [('fp32', TOL_FWD['fp32']), ('bf16x3', TOL_FWD['bf16x3'])])
def test_network_forward(dev, networks, precision, tol):
    """build_network (emulator.py:166-341): GAT, edge fusion, actions, flood head, resnet; B*T = 10 snapshots."""
    args, params, emul, _ = _setup(networks, 'shunqing', dev, precision)
    X, Bd, Ex, a = _inputs(args, 2)
    c = OE.config(args)
    AE = OE.get_edge_action(c, a)
    ry, rey = OE.forward(args, params, X, Bd, Ex, AE)
    f = lambda t: t.float().to(dev)
    y, ey = emul(f(X), f(Bd), f(Ex), emul.get_edge_action(f(a)))
    assert ry.shape == (2, 5, args.state_shape[0], 2) and rey.shape == (2, 5, args.edge_state_shape[0], 3)
    close(y, ry, tol); close(ey, rey, tol)


def test_network_forward_gcn_and_plain_variants(dev, networks):
    """conv = GCN (emulator.py:131-134), no actions / flood / resnet / edge fusion, seq_in > seq_out."""
    args, params, emul, _ = _setup(networks, 'astlingen', dev, conv='GCN', act=False, if_flood=0, resnet=False, edge_fusion=False,
                                   seq_in=6, seq_out=2, embed_size=32, hidden_dim=16, n_sp_layer=1, n_tp_layer=3)
    X, Bd, Ex, _ = _inputs(args, 3)
    ry, rey = OE.forward(args, params, X, Bd, Ex)
    f = lambda t: t.float().to(dev)
    y, ey = emul(f(X), f(Bd), f(Ex))
    assert ry.shape == (3, 2, 30, 3)
    close(y, ry, TOL_FWD['fp32']); close(ey, rey, TOL_FWD['fp32'])


def test_network_forward_deep_shipped_configuration(dev, networks):
    """The deepest configuration the reference ships (`*_5lyrs` models: n_sp_layer = n_tp_layer = 5, if_flood = 5, act, edge fusion,
    resnet, Gaussian-kernel adjacency `length > 0`): Conv1D dilations up to 16 (beyond the streaming kernel's 1 / 2 / 4: the row-GEMM
    form), a flood chain of five hidden layers (the head epilogue's maximum: layers 2..5 read their fragments from LDS)."""
    from oracle import graphs as OG
    net = networks['astlingen']
    edges = np.array(net['edges'])
    rng = np.random.default_rng(4)
    lengths = 50.0 + 400.0 * rng.random(len(edges))
    over = dict(n_sp_layer=5, n_tp_layer=5, if_flood=5, seq_in=20, seq_out=20, adj=OG.adjacency(edges, length=300.0, lengths=lengths))
    args, params, emul, _ = _setup(networks, 'astlingen', dev, **over)
    X, Bd, Ex, a = _inputs(args, 1)
    AE = OE.get_edge_action(OE.config(args), a)
    ry, rey = OE.forward(args, params, X, Bd, Ex, AE)
    f = lambda t: t.float().to(dev)
    y, ey = emul(f(X), f(Bd), f(Ex), emul.get_edge_action(f(a)))
    assert len(emul.tem1_x) == 5 and emul.tem1_x[4].dilation_rate == 16 and len(emul.flood) == 5      # five hidden layers: still the fused head epilogue
    close(y, ry, TOL_FWD['bf16x3']); close(ey, rey, TOL_FWD['bf16x3'])      # 10 spatial + 10 temporal + 5 head layers deep: observed 1.0e-5


@pytest.mark.parametrize('over', [dict(), dict(act=False, if_flood=0, resnet=False, recurrent='GRU', embed_size=32, hidden_dim=64, n_tp_layer=1),
                                  dict(recurrent='None', edge_fusion=False, n_sp_layer=1)])
def test_network_forward_mlp_baseline(dev, networks, over):
    """conv = False: the reference's non-graph baseline (`net = Dense`, emulator.py:181-182,197-212,236-237 -- its shipped `*_nncat_*`
    models): flattened node / link rows, Dense(2 d) blocks, heads for all nodes at once; forward and `predict_tf` (the post-processing
    still balances link flows on the graph) against the oracle."""
    args, params, emul, norms = _setup(networks, 'astlingen', dev, conv='False', seq_in=5, seq_out=5, **over)
    X, Bd, Ex, a = _inputs(args, 3)
    c = OE.config(args)
    AE = OE.get_edge_action(c, a) if c.act else None
    ry, rey = OE.forward(args, params, X, Bd, Ex, AE)
    f = lambda t: t.float().to(dev)
    y, ey = emul(f(X), f(Bd), f(Ex), emul.get_edge_action(f(a)) if c.act else None)
    assert ry.shape == (3, 5, c.n_node, c.n_out + (1 if c.if_flood else 0)) and rey.shape == (3, 5, c.n_edge, c.e_out)
    close(y, ry, TOL_FWD['bf16x3']); close(ey, rey, TOL_FWD['bf16x3'])
    names = [n for n, _, _ in emul.keras_layer_map()]
    assert all(n.startswith(('dense', 'conv1d', 'gru', 'lstm')) for n in names) and 'dense_resx' in names
    with pytest.raises(NotImplementedError):
        U.Emulator('False', True, 'Conv1D', emulator_args(np.array(networks['astlingen']['edges']), networks['astlingen']['n_node'], conv='False',
                                                          seq_in=6, seq_out=2))


@pytest.mark.parametrize('precision,embed', [('fp32', 64), ('bf16x3', 64), ('bf16x3', 128)])
def test_network_forward_trained_node_edge_bias(dev, networks, precision, embed):
    """The whole network with reference-TRAINED NodeEdge layers (dense bias non-zero off the incidence support, emulator.py:36-45)
    in every spatial layer: 64-wide layers on the fused kernel with the remainder GEMM, the 96-wide first layer of block 2
    (`concat([x, b])`, :260-262) on the unfused chain with the same GEMM -- against the dense oracle, which simply multiplies
    the full (N, E) matrices."""
    # embed 128 / hidden 64 = the reference's default sizes (utils/config.yaml:39): the stock model after training
    args, params, emul, _ = _setup(networks, 'shunqing', dev, precision=precision, n_sp_layer=2, n_tp_layer=1, seq_in=4, seq_out=3,
                                   embed_size=embed, hidden_dim=64)
    g = torch.Generator().manual_seed(17)
    for blk in ('block1', 'block2'):
        for q in params[blk]:
            for key in ('node_edge_n', 'node_edge_e'):
                q[key]['bias'] = torch.randn(q[key]['bias'].shape, generator=g, dtype=torch.float64) * 0.02
    load_emulator(emul, params, dev)
    X, Bd, Ex, a = _inputs(args, 2)
    AE = OE.get_edge_action(OE.config(args), a)
    ry, rey = OE.forward(args, params, X, Bd, Ex, AE)
    f = lambda t: t.float().to(dev)
    y, ey = emul(f(X), f(Bd), f(Ex), emul.get_edge_action(f(a)))
    close(y, ry, TOL_FWD[precision]); close(ey, rey, TOL_FWD[precision])
    paths = [ly.last_path for ly in emul.block1.layers] + [ly.last_path for ly in emul.block2.layers]
    if precision == 'bf16x3':
        # d = 64: the 96-wide first layer of block 2 already uses the split-input slot the remainder rides in -> unfused; d = 128: that
        # layer is 128 wide like the others, and the column-split kernel takes the remainder beside its rows
        first_b2 = 'unfused' if embed == 64 else 'fused+remainder'
        assert paths == ['fused+remainder', 'fused+remainder', first_b2, 'fused+remainder'], paths
    # and the bias matters: the same inputs through the support-only model differ visibly
    for blk in ('block1', 'block2'):
        for q in params[blk]:
            for key in ('node_edge_n', 'node_edge_e'):
                q[key]['bias'] = torch.zeros_like(q[key]['bias'])
    r0y, _ = OE.forward(args, params, X, Bd, Ex, AE)
    assert float((ry - r0y).abs().max()) > 10 * TOL_FWD['bf16x3']


@pytest.mark.parametrize('graph_base', [0, 1])
def test_network_forward_diffusion(dev, networks, graph_base):
    """conv = Diffusion (emulator.py:135-138) through the whole network, two-graph and graph_base forms (parity unpinned)."""
    args, params, emul, _ = _setup(networks, 'astlingen', dev, conv='Diffusion', act=bool(graph_base), if_flood=0, seq_in=4, seq_out=4,
                                   embed_size=32, hidden_dim=32, n_sp_layer=2, n_tp_layer=1, graph_base=graph_base)
    # a diffusion filter sums over all features of all nodes (the constant coefficient reaches every zero entry of the filter):
    # with glorot-sized coefficients the heads saturate and the comparison would be vacuous -- shrink them
    for blk in ('block1', 'block2'):
        for q in params[blk]:
            for conv in (q['gat'],) if 'gat' in q else (q['gat_x'], q['gat_e']):
                conv['theta'] *= 0.002
    load_emulator(emul, params, dev)
    X, Bd, Ex, a = _inputs(args, 2)           # graph_base stacks node and link rows: equal widths need the action embedding
    AE = OE.get_edge_action(OE.config(args), a) if graph_base else None
    ry, rey = OE.forward(args, params, X, Bd, Ex, AE)
    f = lambda t: t.float().to(dev)
    y, ey = emul(f(X), f(Bd), f(Ex), emul.get_edge_action(f(a)) if graph_base else None)
    assert float(ry.std()) > 1e-3 and float(((ry > 0.02) & (ry < 0.98)).double().mean()) > 0.5      # not saturated: the comparison bites
    close(y, ry, TOL_FWD['bf16x3']); close(ey, rey, TOL_FWD['bf16x3'])


@pytest.mark.parametrize('precision', ['fp32', 'bf16x3'])
@pytest.mark.parametrize('F', [64, 128, 96])
@pytest.mark.parametrize('kind', ['GRU', 'LSTM'])
def test_recurrent_layer_64(dev, kind, F, precision):
    """One F -> 64 GRU / LSTM layer over (B, T, R, F) with a ragged row count against the step-by-step oracle: the one-launch
    matrix-core kernel on the input rows (uds_recurrent_fused: F = 64, and 128 -- the first temporal layer of a d = 128 model
    -- where W + U fit the LDS: the GRU), the same kernel on a row-GEMM input projection (LSTM at 128, F = 96), and the Dense +
    exact-fp32 recurrence pair (precision fp32)."""
    from gnn_uds_amd.emulator import GRU, LSTM
    g = torch.Generator().manual_seed(21)
    B, T, R = 2, 11, 37
    layer = (GRU if kind == 'GRU' else LSTM)(64, in_features=F, generator=g, precision=precision).to(dev)
    with torch.no_grad():
        layer.bias.add_(torch.rand(layer.bias.shape, generator=g).to(dev) * 0.2 - 0.1)
    assert _lib.recurrent_fused_supported(F, kind) == (F == 64 or (F == 128 and kind == 'GRU'))
    x = rnd(g, B, T, R, F) * 2 - 1
    f = OE.gru_sequence if kind == 'GRU' else OE.lstm_sequence
    xs = x.permute(0, 2, 1, 3).reshape(B * R, T, F)
    ref = f(xs, layer.kernel.double().cpu(), layer.recurrent_kernel.double().cpu(), layer.bias.double().cpu())
    ref = ref.reshape(B, R, T, 64).permute(0, 2, 1, 3)
    out = layer(x.float().to(dev))
    close(out, ref, TOL_FWD[precision])
    one = layer(x[:, :1].float().to(dev).contiguous())            # T = 1
    close(one, ref[:, :1], TOL_FWD[precision])


@pytest.mark.parametrize('recurrent', ['GRU', 'LSTM', 'None'])
def test_network_forward_recurrent_variants(dev, networks, recurrent):
    """`get_tem_nets` (emulator.py:154-163): GRU / LSTM temporal layers (uds_recurrent_forward after a Dense input projection)
    and the variant without temporal layers, against the step-by-step oracle (TF 2.10 gate conventions restated; unpinned)."""
    args, params, emul, _ = _setup(networks, 'astlingen', dev, recurrent=recurrent, seq_in=7, seq_out=4, embed_size=32, hidden_dim=16,
                                   n_sp_layer=1, n_tp_layer=2, if_flood=2)
    X, Bd, Ex, a = _inputs(args, 2)
    AE = OE.get_edge_action(OE.config(args), a)
    ry, rey = OE.forward(args, params, X, Bd, Ex, AE)
    f = lambda t: t.float().to(dev)
    y, ey = emul(f(X), f(Bd), f(Ex), emul.get_edge_action(f(a)))
    assert len(emul.tem1_x) == (0 if recurrent == 'None' else 2)
    close(y, ry, TOL_FWD['bf16x3']); close(ey, rey, TOL_FWD['bf16x3'])
    names = [n for n, _, _ in emul.keras_layer_map()]
    assert (recurrent.lower() + '_3' in names) == (recurrent != 'None') and 'conv1d' not in names


@pytest.mark.parametrize('graph_base', [0, 1])
def test_network_forward_use_adj(dev, networks, graph_base):
    """`use_adj` (emulator.py:178-180,268-271,343-362): the control action rewrites the adjacency entry of every actuated link
    per time step (GAT: cast to int, a setting < 1 removes it); block 2's node-side GAT sees a per-snapshot mask.  The HIP
    path takes the mask over the CSR entries, the oracle the dense (B, T, n, n) array; both forms of ADJ give the same."""
    act_edges = np.array(networks['astlingen']['edges'])[::2]     # half of the links actuated: the mask moves the outputs visibly
    args, params, emul, _ = _setup(networks, 'astlingen', dev, use_adj=True, if_flood=0, seq_in=4, seq_out=3, embed_size=32,
                                   hidden_dim=32, n_sp_layer=2, n_tp_layer=1, graph_base=graph_base, act_edges=act_edges)
    X, Bd, Ex, _ = _inputs(args, 2)
    g = torch.Generator().manual_seed(5)
    a = (torch.rand(2, 3, len(args.act_edges), generator=g) > 0.4).double() * (0.5 + torch.rand(2, 3, len(args.act_edges), generator=g))
    c = OE.config(args)
    AE, ADJ = OE.get_edge_action(c, a), OE.get_adj_action(c, a)
    assert 0 < float((ADJ != torch.from_numpy(np.asarray(c.adj, dtype=np.float64))).double().mean())       # some entries switched off
    ry, rey = OE.forward(args, params, X, Bd, Ex, AE, ADJ)
    r0y, _ = OE.forward(args, params, X, Bd, Ex, AE)
    assert float((ry - r0y).abs().max()) > 0                     # the outputs feel it (the layer-level test in test_gpu_parity.py bounds the effect from below)
    f = lambda t: t.float().to(dev)
    mask = emul.get_adj_action(f(a))
    assert mask.shape[:2] == (2, 3) and float(mask.min()) == 0.0
    y, ey = emul(f(X), f(Bd), f(Ex), emul.get_edge_action(f(a)), mask)
    close(y, ry, TOL_FWD['bf16x3']); close(ey, rey, TOL_FWD['bf16x3'])
    y2, ey2 = emul(f(X), f(Bd), f(Ex), emul.get_edge_action(f(a)), f(ADJ))           # the reference's dense layout
    assert torch.equal(y, y2) and torch.equal(ey, ey2)
    ym, _ = emul._model(f(X), f(a), f(Bd), f(Ex))                # _model builds both action tensors itself (:427-433)
    assert ym.shape[:3] == y.shape[:3] and bool(torch.isfinite(ym).all())


@pytest.mark.parametrize('variant', ['edge_fusion_act', 'pumps_offset_tide', 'plain', 'mlp_baseline'])
def test_predict_tf(dev, networks, variant):
    """predict_tf (emulator.py:604-641) incl. post_proc_tf / constrain_tf branches; raw states in, physical units out."""
    rng = np.random.default_rng(0)
    net = networks['astlingen']
    n, e = net['n_node'], len(net['edges'])
    over = dict(edge_fusion=True, act=True)
    if variant == 'pumps_offset_tide':
        over = dict(edge_fusion=False, act=True, tide=True, pump=0.1 + rng.random(e), pump_in=rng.random(n) * (rng.random(n) > 0.7),
                    pump_out=rng.random(n) * (rng.random(n) > 0.7), offset=rng.random(e) * (rng.random(e) > 0.5), area=rng.random(n),
                    epsilon=0.1)
    elif variant == 'plain':
        over = dict(edge_fusion=False, act=False, if_flood=0, epsilon=0.0)
    elif variant == 'mlp_baseline':      # conv = False (the `*_nncat_*` models) through the same post-processing
        over = dict(edge_fusion=True, act=True, conv='False')
    args, params, emul, norms = _setup(networks, 'astlingen', dev, 'fp32', **over)
    X, Bd, Ex, a = _inputs(args, 2)
    ry, rey = OE.predict(args, params, norms, X, Bd, a if args.act else None, Ex)
    f = lambda t: t.float().to(dev)
    y, ey = emul.predict_tf(f(X), f(Bd), f(a) if args.act else None, f(Ex))
    assert ry.shape == (2, 5, n, 5 if args.if_flood else 4) and rey.shape == (2, 5, e, 3)
    # hard thresholds (flood bit > 0.5, depth > 0.01, ...) can flip on values within rounding of the threshold:
    # compare the bulk tightly and allow a handful of flipped entries
    for out, ref in ((y, ry), (ey, rey)):
        d = (out.double().cpu() - ref).abs()
        bad = int((d > 2e-5 * max(1.0, float(ref.abs().max()))).sum())
        assert bad <= OUTLIERS_ALLOWED, (variant, bad, ref.numel(), float(d.max()))


@pytest.mark.parametrize('variant', ['mixed_pumps_edge_fusion', 'mixed_pumps_node_gates', 'no_pumps'])
def test_predict_and_simulate_numpy_mode(dev, networks, variant):
    """`predict` / `simulate` (emulator.py:566-602, 521-564) go through the NumPy `post_proc` (:643-678), which is NOT
    `post_proc_tf`: on a network where SOME links are pumps (the usual act = True case) the rated-pump override applies in
    `post_proc` and not in `post_proc_tf` (`pump.min() > 0` is false), and the node pumps open at depth > 0.01 instead of
    > 0.  Oracle: the reference's loop restated as a loop, one un-batched NumPy-mode forward per window
    (oracle/emulator_ref.py: simulate); the product batches the windows into one forward."""
    rng = np.random.default_rng(3)
    net = networks['astlingen']
    n, e = net['n_node'], len(net['edges'])
    some = (rng.random(e) > 0.6)
    if variant == 'mixed_pumps_edge_fusion':
        over = dict(edge_fusion=True, act=True, pump=(0.1 + rng.random(e)) * some, area=rng.random(n), offset=rng.random(e) * (rng.random(e) > 0.5))
    elif variant == 'mixed_pumps_node_gates':
        over = dict(edge_fusion=False, act=True, tide=True, pump=(0.1 + rng.random(e)) * some, pump_in=rng.random(n) * (rng.random(n) > 0.7),
                    pump_out=rng.random(n) * (rng.random(n) > 0.7), area=rng.random(n), epsilon=0.1)
    else:
        over = dict(edge_fusion=True, act=True)
    args, params, emul, norms = _setup(networks, 'astlingen', dev, 'fp32', **over)
    n_win = 4
    X, Bd, Ex, a = _inputs(args, n_win)
    f = lambda t: t.float().to(dev)
    ry, rey = OE.simulate(args, params, norms, X, Bd, a, Ex)                    # the per-window loop
    y, ey = emul.simulate(f(X), f(Bd), f(a), f(Ex))                             # all windows in one forward
    py, pey = emul.predict(f(X), f(Bd), f(a), f(Ex))
    assert torch.equal(y, py) and torch.equal(ey, pey)
    assert tuple(y.shape) == tuple(ry.shape) and tuple(ey.shape) == tuple(rey.shape)
    # hard thresholds (flood bit > 0.5, depth > 0.01, flow > 0 ...) can flip on values within rounding of the threshold:
    # compare the bulk tightly and allow a handful of flipped entries
    for out, ref in ((y, ry), (ey, rey)):
        d = (out.double().cpu() - ref).abs()
        bad = int((d > 2e-5 * max(1.0, float(ref.abs().max()))).sum())
        assert bad <= OUTLIERS_ALLOWED, (variant, bad, ref.numel(), float(d.max()))
    # the two modes really differ where the reference's do: link flows on the mixed-pump networks
    ty, tey = emul.predict_tf(f(X), f(Bd), f(a), f(Ex))
    differs = float((tey - ey).abs().max()) > 1e-3
    assert differs == (variant != 'no_pumps'), (variant, float((tey - ey).abs().max()))
    rty, rtey = OE.predict(args, params, norms, X, Bd, a, Ex)                   # and predict_tf still matches ITS restatement
    d = (tey.double().cpu() - rtey).abs()
    assert int((d > 2e-5 * max(1.0, float(rtey.abs().max()))).sum()) <= OUTLIERS_ALLOWED


def test_model_rollout(dev, networks):
    """_model with roll = 3 (emulator.py:401-425): autoregressive chunks, flood bit thresholded, window shifted."""
    args, params, emul, norms = _setup(networks, 'astlingen', dev, 'fp32', roll=3, seq_in=4, seq_out=2, n_sp_layer=1)
    X, Bd, Ex, a = _inputs(args, 2)
    ry, rey = OE.model_rollout(args, params, norms, X, a, Bd, Ex)
    f = lambda t: t.float().to(dev)
    y, ey = emul._model(f(X), f(a), f(Bd), f(Ex))
    assert ry.shape == (2, 6, 30, 4)
    d = (y.double().cpu() - ry).abs()
    assert int((d > 2e-5).sum()) <= OUTLIERS_ALLOWED, float(d.max())
    de = (ey.double().cpu() - rey).abs()
    assert int((de > 2e-5 * max(1.0, float(rey.abs().max()))).sum()) <= OUTLIERS_ALLOWED, float(de.max())


def test_c2_rollout_100_fed_back_steps(dev):
    """BASELINE.json config 2 at its stated size: N = 2 000 / E = 2 500, d = 64, 3 + 3 spatial layers, seq_in 6, seq_out 1, and
    100 autoregressive steps, every one fed with the previous step's post-processed prediction and its THRESHOLDED flood bit
    (emulator.py:400-425; mpc.py:565-582) -- eager `_model` and `rollout_graphed` against the fp64 oracle on the same inputs.

    A hard threshold fed back 100 times is where 1e-6 differences can flip a state: the oracle walks the same 100 steps in fp64
    and, at every step, compares its flood bits with the bits the GPU run fed back.  They must be identical except where the
    fp64 probability lies within 1e-5 of 0.5; at such a state (and only there) the oracle follows the GPU's bit, so that the
    two trajectories stay comparable to the last step.  The continuous outputs are bounded at every step and at step 100."""
    N, E, steps = 2000, 2500, 100
    edges = U.synthetic_drainage_network(N, E, 0)
    args = emulator_args(edges, N, seq_in=6, seq_out=1, roll=steps, n_sp_layer=3, n_tp_layer=2, if_flood=3, act=False, embed_size=64,
                         hidden_dim=64)
    params = OE.init_params(args, seed=3)
    norms = emulator_norms(args)
    c = OE.config(args)
    g = torch.Generator().manual_seed(11)
    X, Bd, Ex = rnd(g, 1, 6, N, c.n_in), rnd(g, 1, steps, N, 1) * 0.1, rnd(g, 1, 6, E, 4)
    def oracle_walk(n, follow=None):
        """n fed-back steps in fp64 (`model_rollout`, emulator.py:401-425, one step at a time); `follow` (1, n, N) bool: the bits
        to feed back instead of the oracle's own.  Yields (step, y, ey, flood probability)."""
        x, ex = X, Ex
        for i in range(n):
            sl = slice(i, i + 1)
            y, ey = OE.forward(args, params, x[:, -6:], Bd[:, sl], ex[:, -6:], None)
            y, ey = OE.post_proc(args, norms, y, ey, None, Bd[:, sl])
            prob = y[..., -1]
            yield i, y, ey, prob
            bit = (prob > 0.5) if follow is None else follow[:, sl]
            x_new = torch.cat([y[..., :-1], bit.unsqueeze(-1).double(), Bd[:, sl]], dim=-1)                 # :417
            x = torch.cat([x[:, 1:], x_new], dim=1)
            ex = torch.cat([ex[:, 1:], torch.cat([ey, torch.ones_like(ey[..., :1])], dim=-1)], dim=1)        # :422

    # a freshly initialised flood head answers 0.52 +- 0.01 everywhere: every bit on one side, nothing to flip.  Spread its logits
    # (kernel x 8) and centre them on the threshold where the fed-back trajectory settles (three rounds of 8 steps), so that
    # a third of the bits are on and a dozen states of the 200 000 land within 1e-5 of 0.5
    OE.SPARSE_SPATIAL = True
    try:
        params['flood_out']['kernel'] = params['flood_out']['kernel'] * 8.0
        for _ in range(3):
            for _, _, _, prob in oracle_walk(8):
                pass
            params['flood_out']['bias'] = params['flood_out']['bias'] - torch.logit(prob.median().clamp(1e-6, 1 - 1e-6))
    finally:
        OE.SPARSE_SPATIAL = False
    emul = load_emulator(U.Emulator(args.conv, args.resnet, args.recurrent, args), params, dev)
    emul.set_norm(*(norms[k].numpy() for k in 'xbyre'))
    f = lambda t: t.float().to(dev)
    y_e, ey_e = emul._model(f(X), None, f(Bd), f(Ex))
    y_g, ey_g = emul.rollout_graphed(f(X), None, f(Bd), f(Ex))
    assert torch.equal(y_e, y_g) and torch.equal(ey_e, ey_g)          # the captured graph replays the eager loop bit for bit
    emul.drop_graph()
    assert emul.block1.layers[0].last_path.startswith('fused')
    y_gpu, ey_gpu = y_e.double().cpu(), ey_e.double().cpu()
    bits_gpu = y_gpu[..., -1] > 0.5                                    # what the GPU run fed back (the value it thresholded)

    OE.SPARSE_SPATIAL = True
    try:
        err_y, err_e, near, flips = [], [], 0, []
        for i, y, ey, prob in oracle_walk(steps, follow=bits_gpu):
            sl = slice(i, i + 1)
            diff = (prob > 0.5) != bits_gpu[:, sl]
            near += int(((prob - 0.5).abs() <= 1e-5).sum())
            if bool(diff.any()):                                       # only at states on the threshold (the oracle then follows the GPU)
                assert float((prob - 0.5).abs()[diff].max()) <= 1e-5, (i, float((prob - 0.5).abs()[diff].max()))
                flips.append((i, int(diff.sum())))
            err_y.append(float((y_gpu[:, sl] - y.clamp(0, 1)).abs().max()))
            err_e.append(float((ey_gpu[:, sl] - ey).abs().max()) / max(1.0, float(ey.abs().max())))
    finally:
        OE.SPARSE_SPATIAL = False
    report = 'max err per step (nodes): first %.2e, worst %.2e at step %d, last %.2e; links worst %.2e; %d states within 1e-5 of ' \
             'the threshold, bit overrides %r' % (err_y[0], max(err_y), int(np.argmax(err_y)), err_y[-1], max(err_e), near, flips)
    print(report)
    assert float(bits_gpu.float().mean()) > 0.02 and float((~bits_gpu).float().mean()) > 0.02, 'flood bits all on one side: nothing is tested'
    assert len(flips) <= max(1, near), report
    assert max(err_y) <= ROLL100_TOL and max(err_e) <= ROLL100_TOL and err_y[-1] <= ROLL100_TOL, report


@pytest.mark.parametrize('graph_base', [1, 2])
def test_graph_base_variants(dev, networks, graph_base):
    """graph_base = 1 / 2 (`emulator.py:220-223,273-276`): one conv over the stacked node + link rows on the combined
    graph of `get_node_based_adj` / `get_edge_based_adj`; forward parity with the oracle from the
    dense `args.adj` AND from the CSR builders (`args.graph`)."""
    net = networks['astlingen']
    edges, n = np.array(net['edges']), net['n_node']
    args = emulator_args(edges, n, graph_base=graph_base, n_sp_layer=2)
    params = OE.init_params(args, seed=4)
    c = OE.config(args)
    g = torch.Generator().manual_seed(9)
    X, Bd, Ex = rnd(g, 2, c.seq_in, n, c.n_in), rnd(g, 2, c.seq_out, n, c.b_in), rnd(g, 2, c.seq_in, len(edges), c.e_in)
    AE = rnd(g, 2, c.seq_out, len(edges), 1)
    ry, re = OE.forward(args, params, X, Bd, Ex, AE)
    f = lambda t: t.float().to(dev)
    emul = load_emulator(U.Emulator(args.conv, args.resnet, args.recurrent, args), params, dev)
    y, ey = emul(f(X), f(Bd), f(Ex), f(AE))
    close(y, ry, TOL_FWD['bf16x3']); close(ey, re, TOL_FWD['bf16x3'])
    args2 = emulator_args(edges, n, graph_base=graph_base, n_sp_layer=2)
    args2.graph, args2.adj = U.DrainageGraph.from_edges(edges, n), None
    emul2 = load_emulator(U.Emulator(args2.conv, args2.resnet, args2.recurrent, args2), params, dev)
    y2, ey2 = emul2(f(X), f(Bd), f(Ex), f(AE))
    assert torch.equal(y, y2) and torch.equal(ey, ey2)
    names = [nm for nm, _, _ in emul.keras_layer_map()]
    assert names[4:6] == ['mixed_gat', 'mixed_gat_1'] and 'node_edge' not in names


@pytest.mark.parametrize('graph_base,use_pred', [(0, False), (0, True), (1, False)])
def test_rl_convnet_encoder(dev, networks, graph_base, use_pred):
    """`ConvNet` of the RL agents (agent.py:20-99): embeddings + the spatial block + GlobalAttnSumPool; batch of single
    snapshots.  Tolerance TOL_FWD['bf16x3'] (fused split-bf16 layers)."""
    from oracle import emulator_ref as ER
    net = networks['shunqing']
    edges, n = np.array(net['edges']), net['n_node']
    args = emulator_args(edges, n, graph_base=graph_base, n_sp_layer=2, conv_dim=64, use_pred=use_pred, if_flood=0, activation='relu')
    args.edge_state_shape = (len(edges), 3)
    gen = torch.Generator().manual_seed(7)
    d, h, n_in = 64, 32, 4 + (1 if use_pred else 0)
    gl = lambda *s: ER._glorot(gen, s)
    dense = lambda fi, fo: {'kernel': gl(fi, fo), 'bias': torch.randn(fo, generator=gen, dtype=torch.float64) * 0.05}
    conv = lambda f: {'kernel': gl(f, 1, d), 'attn_kernel_self': gl(d, 1, 1), 'attn_kernel_neighs': gl(d, 1, 1),
                      'bias': torch.randn(d, generator=gen, dtype=torch.float64) * 0.05}
    ne = lambda r, m: {'weight': torch.randn(r, m, generator=gen, dtype=torch.float64) * 0.05, 'bias': torch.zeros(r, m, dtype=torch.float64)}
    if graph_base:
        layers = [{'gat': conv(d)} for _ in range(2)]
    else:
        layers = [{'dense_xe': dense(d, h), 'dense_ex': dense(d, h), 'node_edge_n': ne(n, len(edges)), 'node_edge_e': ne(len(edges), n),
                   'gat_x': conv(d + h), 'gat_e': conv(d + h)} for _ in range(2)]
    params = {'embed_x': dense(n_in, d), 'embed_e': dense(3, d), 'block': layers, 'pool': {'attn_kernel': gl(d, 1)}}
    Bn = 6
    X, E, Bd = rnd(gen, Bn, n, 4), rnd(gen, Bn, len(edges), 3), rnd(gen, Bn, n, 1)
    ref = ER.convnet_forward(args, params, X, E, Bd if use_pred else None)
    m = U.ConvNet(args, 'GAT').to(dev)
    f32 = lambda t: t.float().to(dev).contiguous()
    m.embed_x.kernel.data, m.embed_x.bias.data = f32(params['embed_x']['kernel']), f32(params['embed_x']['bias'])
    m.embed_e.kernel.data, m.embed_e.bias.data = f32(params['embed_e']['kernel']), f32(params['embed_e']['bias'])
    m.pool.attn_kernel.data = f32(params['pool']['attn_kernel'])
    for ly, q in zip(m.block.layers, layers):
        if graph_base:
            ly.kernel.data, ly.bias.data = f32(q['gat']['kernel']), f32(q['gat']['bias'])
            ly.attn_kernel_self.data, ly.attn_kernel_neighs.data = f32(q['gat']['attn_kernel_self']), f32(q['gat']['attn_kernel_neighs'])
        else:
            for mod, key in ((ly.dense_xe, 'dense_xe'), (ly.dense_ex, 'dense_ex')):
                mod.kernel.data, mod.bias.data = f32(q[key]['kernel']), f32(q[key]['bias'])
            for mod, key in ((ly.node_edge_n, 'node_edge_n'), (ly.node_edge_e, 'node_edge_e')):
                mod.weight.data, mod.bias.data = f32(q[key]['weight']), f32(q[key]['bias'])
            for mod, key in ((ly.gat_x, 'gat_x'), (ly.gat_e, 'gat_e')):
                mod.kernel.data, mod.bias.data = f32(q[key]['kernel']), f32(q[key]['bias'])
                mod.attn_kernel_self.data, mod.attn_kernel_neighs.data = f32(q[key]['attn_kernel_self']), f32(q[key]['attn_kernel_neighs'])
    out = m(f32(X), f32(E), f32(Bd) if use_pred else None)
    assert tuple(out.shape) == (Bn, 64)
    close(out, ref, TOL_FWD['bf16x3'])


def test_emulator_at_the_reference_default_sizes(dev, networks):
    """embed_size 128, hidden_dim 64, n_sp_layer 2, n_tp_layer 2, seq_in 6, seq_out 1 (utils/config.yaml): the d = 128 fused
    kernel (block 1, later layers of block 2), the 64-column-block row GEMM for 128-wide Dense outputs, the 3 x 128 -> 64
    Conv1D.  Whole-forward tolerance as for d = 64; a batch large enough (rows >= 4096) to take
    the matrix-core paths."""
    net = networks['RedChicoSur']
    edges, n = np.array(net['edges']), net['n_node']
    args = emulator_args(edges, n, embed_size=128, hidden_dim=64, n_sp_layer=2, n_tp_layer=2, seq_in=6, seq_out=1)
    params = OE.init_params(args, seed=8)
    c = OE.config(args)
    g = torch.Generator().manual_seed(3)
    Bn = 12
    X, Bd, Ex = rnd(g, Bn, c.seq_in, n, c.n_in), rnd(g, Bn, c.seq_out, n, c.b_in), rnd(g, Bn, c.seq_in, len(edges), c.e_in)
    AE = rnd(g, Bn, c.seq_out, len(edges), 1)
    ry, re = OE.forward(args, params, X, Bd, Ex, AE)
    f = lambda t: t.float().to(dev)
    emul = load_emulator(U.Emulator(args.conv, args.resnet, args.recurrent, args), params, dev)
    y, ey = emul(f(X), f(Bd), f(Ex), f(AE))
    close(y, ry, TOL_FWD['bf16x3']); close(ey, re, TOL_FWD['bf16x3'])
    assert emul.block1.layers[0].network().plan_info()['fused'] & 8


def test_graph_captured_rollout_equals_the_eager_loop(dev, networks):
    """`rollout_graphed`: every autoregressive chunk replays one captured HIP graph; results are bit-identical to the eager
    `_model` loop (same kernels, same order), also when called again with new inputs and after a shape change."""
    net = networks['astlingen']
    edges, n = np.array(net['edges']), net['n_node']
    # the 2nd and 3rd configurations (no control actions) take the fused post-processing + feedback kernels (uds_roll_update):
    # without / with the flood channel, seq_out 2 / 1, window shift by 2 / 1 and seq_out == seq_in (no rows kept)
    for over in (dict(roll=4, seq_in=5, seq_out=1, n_sp_layer=1), dict(roll=3, seq_in=4, seq_out=2, n_sp_layer=1, act=False, if_flood=0),
                 dict(roll=4, seq_in=5, seq_out=1, n_sp_layer=1, act=False), dict(roll=2, seq_in=3, seq_out=3, n_sp_layer=1, act=False),
                 dict(roll=3, seq_in=5, seq_out=1, n_sp_layer=1, act=False, recurrent='GRU'),       # the reference's default temporal net
                 dict(roll=3, seq_in=4, seq_out=2, n_sp_layer=1, recurrent='LSTM')):
        args = emulator_args(edges, n, **over)
        norms = emulator_norms(args)
        emul = load_emulator(U.Emulator(args.conv, args.resnet, args.recurrent, args), OE.init_params(args, seed=2), dev)
        emul.set_norm(*(norms[k].numpy() for k in 'xbyre'))
        c = OE.config(args)
        for seed, B in ((1, 2), (2, 2), (3, 3)):
            g = torch.Generator().manual_seed(seed)
            f = lambda t: t.float().to(dev)
            x, b, ex = f(rnd(g, B, c.seq_in, n, c.n_in)), f(rnd(g, B, c.seq_out * c.roll, n, c.b_in) * 0.1), f(rnd(g, B, c.seq_in, len(edges), c.e_in))
            a = f(rnd(g, B, c.seq_out * c.roll, len(args.act_edges))) if c.act else None
            y0, e0 = emul._model(x, a, b, ex)
            y1, e1 = emul.rollout_graphed(x, a, b, ex)
            assert torch.equal(y0, y1) and torch.equal(e0, e1)
        emul.drop_graph()


@pytest.mark.parametrize('B,R,F', [(5, 59, 128), (3, 1000, 64), (2, 7, 8), (1, 257, 256)])
def test_attn_sum_pool_kernel(dev, B, R, F):
    """uds_attn_sum_pool (spektral GlobalAttnSumPool, agent.py:93-94) against the two-pass formula in fp64; repeatable."""
    g = torch.Generator().manual_seed(B + R)
    x, k = rnd(g, B, R, F) * 4 - 2, rnd(g, F, 1) - 0.5
    alpha = torch.softmax((x @ k).squeeze(-1), dim=-1)
    ref = (alpha.unsqueeze(-2) @ x).squeeze(-2)
    out = _lib.attn_sum_pool(x.float().to(dev), k.float().to(dev))
    close(out, ref, 5e-6)
    assert torch.equal(out, _lib.attn_sum_pool(x.float().to(dev), k.float().to(dev)))


def test_mbrl_agent_in_the_loop_rollout(dev, networks):
    """The model-based-RL virtual rollout (mbrl.py:304-347): policy -> settings -> predict_tf -> feedback, four control steps of
    two simulation steps, against its fp64 restatement (oracle.emulator_ref.mbrl_rollout).  The policy is a fixed smooth
    function of the graph observation (a stand-in for ConvNet + actor head, whose parity is test_rl_convnet_encoder's job):
    sigmoid of a random projection of the mean node / link observation."""
    from gnn_uds_amd import mbrl as MB
    args, params, emul, norms = _setup(networks, 'astlingen', dev, seq_in=4, seq_out=2, n_sp_layer=1, n_tp_layer=1)
    c = OE.config(args)
    n_step, r_step, B = 4, 2, 3
    g = torch.Generator().manual_seed(21)
    N, E, n_act = c.n_node, c.n_edge, len(args.act_edges)
    x, ex = rnd(g, B, c.seq_in, N, c.n_in), rnd(g, B, c.seq_in, E, c.e_in)
    x[..., 3] = (x[..., 3] > 0.7).double()
    a, b, y = rnd(g, B, c.seq_in, n_act), rnd(g, B, n_step * r_step, N, 1) * 0.1, rnd(g, B, c.seq_in, N, 5)
    node_attrs = ['depthN', 'cuminflow', 'outflow_vol', 'flood', 'lateral_infow_vol']      # which channels are summed over the window
    link_attrs = ['depthL', 'volumeL', 'flow_vol', 'setting']
    wn, we = rnd(g, c.n_in, n_act) - 0.5, rnd(g, c.e_in, n_act) - 0.5

    def policy64(obs):
        return torch.sigmoid(obs[0].mean(1) @ wn + obs[-1].mean(1) @ we)

    ref = OE.mbrl_rollout(args, params, norms, policy64, x, a, b, y, ex, n_step, r_step, node_attrs, link_attrs)
    f = lambda t: t.float().to(dev)
    wnd, wed = f(wn), f(we)
    got = MB.rollout(emul, lambda obs: torch.sigmoid(obs[0].mean(1) @ wnd + obs[-1].mean(1) @ wed), f(x), f(a), f(b), f(y), f(ex), n_step, r_step,
                     node_attrs, link_attrs)
    assert len(got) == 4 and tuple(got[0].shape) == (B, c.seq_in + n_step * r_step, N, c.n_in)
    for o, r in zip(got, ref):
        assert tuple(o.shape) == tuple(r.shape)
        d = (o.double().cpu() - r).abs()
        assert int((d > 3e-5 * max(1.0, float(r.abs().max()))).sum()) <= OUTLIERS_ALLOWED, float(d.max())


def test_save_load_and_not_built(dev, networks, tmp_path):
    args, params, emul, norms = _setup(networks, 'astlingen', dev, n_sp_layer=1)
    X, Bd, Ex, a = _inputs(args, 1)
    f = lambda t: t.float().to(dev)
    y0, e0 = emul.predict_tf(f(X), f(Bd), f(a), f(Ex))
    emul.save(str(tmp_path))
    other = U.Emulator(args.conv, args.resnet, args.recurrent, args).to(dev)
    other.load(str(tmp_path))
    y1, e1 = other.predict_tf(f(X), f(Bd), f(a), f(Ex))
    assert torch.equal(y0, y1) and torch.equal(e0, e1)
    for bad in (dict(conv='General'), dict(recurrent='Transformer'), dict(conv='GCN', use_adj=True)):
        from types import SimpleNamespace
        a2 = SimpleNamespace(**{**vars(args), **bad})
        with pytest.raises(NotImplementedError):
            U.Emulator(a2.conv, False, a2.recurrent, a2)


@pytest.mark.parametrize('recurrent', ['Conv1D', 'GRU'])
def test_large_forward_fused_tail_equals_unfused(dev, recurrent):
    """At a size the oracle does not reach (N=6000, T=8): the forward with the one-launch tail kernels (resnet head + output heads,
    one-launch recurrent layer, streaming Conv1D) against the same model run with precision='fp32' (exact-fp32 unfused kernels for
    every layer, each checked against the oracle at small sizes): the two HIP paths share no kernel and agree to the split-bf16
    tolerance of the whole forward."""
    N, E, T = 6000, 7200, 8
    edges = U.synthetic_drainage_network(N, E, 0)
    g = U.DrainageGraph.from_edges(edges)
    from types import SimpleNamespace
    a = SimpleNamespace(state_shape=(N, 4), edge_state_shape=(E, 4), seq_in=T, seq_out=T, embed_size=64, hidden_dim=64, kernel_size=3,
                        n_sp_layer=1, n_tp_layer=2, activation='relu', if_flood=3, edge_fusion=True, edges=edges, act=False, graph=g,
                        model_dir=None)
    fused = U.Emulator('GAT', True, recurrent, a, precision='bf16x3', generator=torch.Generator().manual_seed(1)).to(dev)
    exact = U.Emulator('GAT', True, recurrent, a, precision='fp32', generator=torch.Generator().manual_seed(1)).to(dev)
    with torch.no_grad():
        for p, q in zip(fused.parameters(), exact.parameters()):
            if float(p.abs().sum()) == 0:                       # biases: not all zero
                p.add_(torch.rand(p.shape, generator=torch.Generator().manual_seed(p.numel())).to(dev) * 0.1)
            q.copy_(p)
    gen = torch.Generator().manual_seed(2)
    X, B, Ex = torch.rand(1, T, N, 5, generator=gen).to(dev), torch.rand(1, T, N, 1, generator=gen).to(dev) * 0.1, torch.rand(1, T, E, 4, generator=gen).to(dev)
    y, ey = fused(X, B, Ex)
    ry, rey = exact(X, B, Ex)
    assert y.shape == (1, T, N, 2) and ey.shape == (1, T, E, 3)
    close(y, ry.double().cpu(), TOL_FWD['bf16x3']); close(ey, rey.double().cpu(), TOL_FWD['bf16x3'])
    assert float(ry.std()) > 1e-3                                # the heads are not saturated
