"""GPU parity: the HIP path (through the C ABI) vs the CPU oracle on identical seeded inputs.

Tolerances (stated; relative to max(1, max|ref|), every output element, against the fp64 oracle).  They sit 3-6x above
the largest error the kernels were MEASURED at over this whole file (UDS_TOL_REPORT=1 prints observed / allowed per
call), not orders of magnitude above it, so that an indexing or accumulation slip of 1e-5 fails:
  exact-fp32 kernels            5e-6   (measured <= 1.5e-6: fp32 rounding of K <= 192 term sums, plus expf / tanhf)
  fused split-bf16 kernels      1e-5   (measured <= 1.6e-6: ~2^-16 per product, fp32 accumulation)
  row-GEMM split-bf16 kernels   1e-4   (measured <= 2.7e-5: K up to 384, products summed in MFMA order)
Index bookkeeping (row schedules) is compared bit for bit.  The reference itself holds no
fixtures for this path: the oracle is "parity unpinned" (oracle/__init__.py)."""
import numpy as np
import pytest
import torch

import gnn_uds_amd as U
from gnn_uds_amd import _lib
from oracle import sparse_csr as OS
from oracle import spektral_dense as OD
from tests import util
from tests.util import cast, load_spatial_layer, spatial_params

pytestmark = pytest.mark.gpu
TOL = 5e-6          # exact-fp32 kernels
TOL_BF16X3 = 1e-5   # fused kernels: GEMM operands split into bf16 hi+lo, 3 products, fp32 accumulate (~2^-16/product)
TOL_ROWGEMM = 1e-4  # Dense / Conv1D on the matrix-core row GEMM (split-bf16)
PREC_TOL = {'fp32': TOL, 'bf16x3': TOL_BF16X3}


def close(out, ref, tol=TOL):
    return util.close(out, ref, tol, _depth=2)


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need the MI355X'
    _lib.load()
    return torch.device('cuda', 0)




def rnd(gen, *shape):
    return torch.rand(*shape, generator=gen, dtype=torch.float64)


@pytest.mark.parametrize('rows,fa,fb,fo,act', [
    (1, 8, 0, 4, 'relu'), (63, 64, 0, 32, 'relu'), (257, 64, 32, 64, 'linear'), (1000, 128, 64, 128, 'tanh'),
    (130, 5, 0, 3, 'hard_sigmoid'), (77, 32, 0, 1, 'sigmoid'), (64, 96, 0, 64, 'relu'), (300, 192, 0, 256, 'relu'),
    (40, 64, 0, 7, 'linear'),
    # narrow inputs, many rows: the streaming embedding kernel (emulator.py:198-212)
    (5000, 5, 0, 64, 'relu'), (4100, 1, 0, 32, 'relu'), (6000, 4, 0, 64, 'linear'), (4097, 8, 0, 96, 'tanh'), (4096, 2, 0, 4, 'sigmoid')])
def test_dense_act(dev, rows, fa, fb, fo, act):
    g = torch.Generator().manual_seed(rows + fo)
    xa, k, b = rnd(g, rows, fa) - 0.5, rnd(g, fa + fb, fo) - 0.5, rnd(g, fo) - 0.5
    xb = rnd(g, rows, fb) - 0.5 if fb else None
    ref = OD.dense(xa if xb is None else torch.cat([xa, xb], -1), k, b, act)
    f = lambda t: None if t is None else t.float().to(dev)
    close(_lib.dense_act(f(xa), f(k), f(b), act, f(xb)), ref)


def test_dense_act_attention_scalars(dev):
    g = torch.Generator().manual_seed(0)
    x, k = rnd(g, 3, 50, 40) - 0.5, rnd(g, 40, 64) - 0.5
    a_s, a_n = rnd(g, 64) - 0.5, rnd(g, 64) - 0.5
    f = lambda t: t.float().to(dev)
    out, s_self, s_nbr = _lib.dense_act(f(x), f(k), None, 'linear', attn=(f(a_s), f(a_n)))
    hx = x @ k
    close(out, hx); close(s_self, hx @ a_s); close(s_nbr, hx @ a_n)


@pytest.mark.parametrize('name', ['astlingen', 'shunqing', 'RedChicoSur'])
def test_csr_spmm_and_row_order(dev, networks, name):
    net = networks[name]
    gph = U.DrainageGraph.from_edges(np.array(net['edges']), net['n_node'])
    g = torch.Generator().manual_seed(1)
    for csr in (gph.inc_n, gph.inc_e, gph.adj):
        h = _lib.CsrHandle(csr)
        assert np.array_equal(h.row_order(), csr.degree_sorted_rows())          # bit-exact bookkeeping
        assert h.max_degree() == int(csr.degrees().max())
        x, val, b = rnd(g, 3, csr.n_cols, 32) - 0.5, rnd(g, csr.nnz) - 0.5, rnd(g, 32) - 0.5
        ref = OS.incidence_aggregate_csr(x, csr.rowptr, csr.col, val, csr.n_rows)
        close(_lib.csr_spmm(h, val.float().to(dev), x.float().to(dev)), ref)
        close(_lib.csr_spmm(h, None, x.float().to(dev), b.float().to(dev), 'relu'),
              torch.relu(OS.incidence_aggregate_csr(x, csr.rowptr, csr.col, torch.ones(csr.nnz, dtype=torch.float64), csr.n_rows) + b))


@pytest.mark.parametrize('name,d', [('astlingen', 8), ('hague', 64), ('chaohu', 128)])
def test_gat_forward_vs_dense_masked_oracle(dev, networks, name, d):
    net = networks[name]
    gph = U.DrainageGraph.from_edges(np.array(net['edges']), net['n_node'])
    g = torch.Generator().manual_seed(2)
    S, F = 3, d + d // 2
    x = rnd(g, S, gph.n_node, F)
    k, a_s, a_n, b = rnd(g, F, 1, d) - 0.5, rnd(g, d, 1, 1) - 0.5, rnd(g, d, 1, 1) - 0.5, rnd(g, d) - 0.5
    filt = torch.from_numpy(gph.adj.to_dense())
    filt[torch.arange(gph.n_node), torch.arange(gph.n_node)] = 0          # the layer must force the diagonal itself
    ref = OD.gat_conv_dense(x, filt, k, a_s, a_n, b, 'relu')
    layer = U.GATConv(d, activation='relu')
    out = layer([x.float().to(dev), filt.numpy()])                         # lazy build, dense (N,N) filter input
    f = lambda t: t.float().to(dev)
    layer.kernel.data, layer.attn_kernel_self.data, layer.attn_kernel_neighs.data, layer.bias.data = f(k), f(a_s), f(a_n), f(b)
    close(layer([f(x), filt.numpy()]), ref)
    assert out.shape == (S, gph.n_node, d)
    # (B,T,N,F) leading dims are preserved
    x4 = f(x).reshape(1, S, gph.n_node, F)
    assert layer([x4, filt.numpy()]).shape == (1, S, gph.n_node, d)


def test_gat_isolated_and_empty(dev):
    a = np.zeros((5, 5)); a[0, 1] = a[1, 0] = 1                            # nodes 2..4 isolated
    g = torch.Generator().manual_seed(3)
    x, k = rnd(g, 2, 5, 8), rnd(g, 8, 1, 8) - 0.5
    a_s, a_n = rnd(g, 8, 1, 1), rnd(g, 8, 1, 1)
    ref = OD.gat_conv_dense(x, torch.from_numpy(a), k, a_s, a_n, None, 'linear')
    layer = U.GATConv(8, activation='linear', use_bias=False, in_channels=8).to(dev)
    f = lambda t: t.float().to(dev)
    layer.kernel.data, layer.attn_kernel_self.data, layer.attn_kernel_neighs.data = f(k), f(a_s), f(a_n)
    close(layer([f(x), a]), ref)
    assert layer([f(x)[:0], a]).shape == (0, 5, 8)                         # S = 0


def test_gcn_conv(dev, networks):
    e = np.array(networks['astlingen']['edges'])
    adj = U.graph.get_adj(e)
    ah = U.GCNConv.preprocess(adj)
    assert np.allclose(ah, OD.gcn_preprocess(torch.from_numpy(adj)).numpy(), atol=1e-15)
    g = torch.Generator().manual_seed(4)
    x, k, b = rnd(g, 2, 30, 16), rnd(g, 16, 8) - 0.5, rnd(g, 8) - 0.5
    ref = OD.gcn_conv_dense(x, torch.from_numpy(ah), k, b, 'relu')
    layer = U.GCNConv(8, activation='relu', in_channels=16).to(dev)
    layer.kernel.data, layer.bias.data = k.float().to(dev), b.float().to(dev)
    close(layer([x.float().to(dev), ah]), ref)


@pytest.mark.parametrize('name,act', [('astlingen', 'tanh'), ('shunqing', 'relu')])
def test_diffusion_conv(dev, networks, name, act):
    """DiffusionConv (emulator.py:135-138): uds_diffusion_forward on the support of the normalised adjacency against the
    dense element-wise-polyval oracle (parity unpinned: Spektral 1.3.1 restated from memory, see the oracle's docstring)."""
    e = np.array(networks[name]['edges'])
    adj = U.graph.get_adj(e)
    ah = U.DiffusionConv.preprocess(adj)
    assert np.allclose(ah, OD.diffusion_preprocess(torch.from_numpy(adj)).numpy(), atol=1e-15)
    g = torch.Generator().manual_seed(4)
    n = adj.shape[0]
    x = rnd(g, 3, n, 24) - 0.3
    layer = U.DiffusionConv(16, activation=act, generator=g).to(dev)
    assert tuple(layer.kernel.shape) == (16, 7)                  # K = 6 -> 7 coefficients per output channel
    ref = OD.diffusion_conv_dense(x, torch.from_numpy(ah), layer.kernel.detach().double().cpu(), act)
    close(layer([x.float().to(dev), ah]), ref)
    out4 = layer([x.float().to(dev).reshape(1, 3, n, 24), ah])     # leading dims kept
    assert out4.shape == (1, 3, n, 16)
    with torch.no_grad():
        layer.kernel.mul_(0.5)                                     # the prepared polynomial values follow the parameter
    ref = OD.diffusion_conv_dense(x, torch.from_numpy(ah), layer.kernel.detach().double().cpu(), act)
    close(layer([x.float().to(dev), ah]), ref)


@pytest.mark.parametrize('trained_bias', [False, True])
def test_node_edge_dense_parameters(dev, networks, trained_bias):
    net = networks['shunqing']
    gph = U.DrainageGraph.from_edges(np.array(net['edges']), net['n_node'])
    inci = np.abs(gph.inc_n.to_dense())
    g = torch.Generator().manual_seed(5)
    w = torch.randn(inci.shape, generator=g, dtype=torch.float64) * 0.05
    b = torch.randn(inci.shape, generator=g, dtype=torch.float64) * 0.01 if trained_bias else torch.zeros(inci.shape, dtype=torch.float64)
    x = rnd(g, 4, inci.shape[1], 32)
    ref = OD.node_edge_dense(x, torch.from_numpy(inci), w, b)
    layer = U.NodeEdge(inci).to(dev)
    assert tuple(layer.weight.shape) == inci.shape and tuple(layer.bias.shape) == inci.shape   # reference shapes
    layer.weight.data, layer.bias.data = w.float().to(dev), b.float().to(dev)
    close(layer(x.float().to(dev)), ref)
    assert (layer.support_values()[1] is not None) == trained_bias


@pytest.mark.parametrize('precision', ['fp32', 'bf16x3'])
@pytest.mark.parametrize('name,d,S', [('astlingen', 8, 1), ('shunqing', 64, 3), ('RedChicoSur', 64, 2), ('hague', 128, 2),
                                      ('chaohu', 64, 7), ('astlingen', 64, 1)])
def test_spatial_layer_vs_dense_masked_oracle(dev, networks, name, d, S, precision):
    tol = PREC_TOL[precision]
    net = networks[name]
    gph = U.DrainageGraph.from_edges(np.array(net['edges']), net['n_node'])
    p = spatial_params(gph.n_node, gph.n_edge, d, d, d, seed=7)
    g = torch.Generator().manual_seed(8)
    x, e = rnd(g, S, gph.n_node, d), rnd(g, S, gph.n_edge, d)
    ne = torch.from_numpy(gph.inc_n.to_dense())
    rx, re = OD.spatial_layer_dense(x, e, p, torch.from_numpy(gph.adj.to_dense()), torch.from_numpy(gph.edge_adj.to_dense()), ne)
    layer = load_spatial_layer(U.SpatialLayer(gph, d, 'relu', sparse_params=False, precision=precision), p, dev)
    ox, oe = layer(x.float().to(dev), e.float().to(dev))
    close(ox, rx, tol); close(oe, re, tol)
    if d == 64:
        info = layer.network().plan_info()
        assert info['fused'] & 1 and info['lds_bytes'] <= 160 * 1024
    # trained (dense) NodeEdge bias, the form every reference-trained checkpoint has (emulator.py:36-45): the dense remainder
    # on the MFMA GEMM rides into the fused kernel as 32 extra input columns; same answer as the dense oracle
    p['ne_n_b'] = torch.randn(p['ne_n_b'].shape, generator=g, dtype=torch.float64) * 0.01
    p['ne_e_b'] = torch.randn(p['ne_e_b'].shape, generator=g, dtype=torch.float64) * 0.01
    rx, re = OD.spatial_layer_dense(x, e, p, torch.from_numpy(gph.adj.to_dense()), torch.from_numpy(gph.edge_adj.to_dense()), ne)
    load_spatial_layer(layer, p, dev)
    ox, oe = layer(x.float().to(dev), e.float().to(dev))
    close(ox, rx, tol); close(oe, re, tol)
    # (d = 128, the reference's default width: the remainder is added to the aggregate inside the column-split kernel, uds_spatial_layer_forward_rem)
    assert layer.last_path == ('fused+remainder' if d in (64, 128) and precision == 'bf16x3' else 'unfused')
    if d == 64 and precision == 'bf16x3':
        # d = 64 has two carriers for the remainder: the wave-specialised kernel adds it to its aggregate (the call above); the
        # 96-wide split-input kernel takes it as 32 extra input columns (the fallback for tile plans the first one refuses)
        assert getattr(layer, '_ws_rem_ok', True)
        layer._ws_rem_ok = False
        ox2, oe2 = layer(x.float().to(dev), e.float().to(dev))
        close(ox2, rx, tol); close(oe2, re, tol)
        assert layer.last_path == 'fused+remainder'
        layer._ws_rem_ok = True
    p['ne_e_b'] = torch.zeros_like(p['ne_e_b'])        # trained on the node side only
    rx, re = OD.spatial_layer_dense(x, e, p, torch.from_numpy(gph.adj.to_dense()), torch.from_numpy(gph.edge_adj.to_dense()), ne)
    load_spatial_layer(layer, p, dev)
    ox, oe = layer(x.float().to(dev), e.float().to(dev))
    close(ox, rx, tol); close(oe, re, tol)


@pytest.mark.parametrize('precision', ['fp32', 'bf16x3'])
def test_spatial_layer_with_per_snapshot_adjacency(dev, networks, precision):
    """`use_adj` at the layer (emulator.py:268-271,282): node-side GAT with a different adjacency per snapshot -- entries of the
    static pattern switched off by a (S, nnz) mask (uds_gat_aggregate_masked) -- against the dense oracle fed the (S, N, N)
    arrays; the diagonal is kept whatever the mask says (set_diag), and the masked result differs visibly from the plain one."""
    tol = PREC_TOL[precision]
    net = networks['shunqing']
    gph = U.DrainageGraph.from_edges(np.array(net['edges']), net['n_node'])
    d, S = 64, 3
    p = spatial_params(gph.n_node, gph.n_edge, d, d, d, seed=7)
    g = torch.Generator().manual_seed(8)
    x, e = rnd(g, S, gph.n_node, d), rnd(g, S, gph.n_edge, d)
    ne = torch.from_numpy(gph.inc_n.to_dense())
    adj = torch.from_numpy(gph.adj.to_dense())
    mask = (torch.rand(S, gph.adj.nnz, generator=g) > 0.4).double()
    rows = np.repeat(np.arange(gph.n_node), np.diff(gph.adj.rowptr))
    cols = np.asarray(gph.adj.col, dtype=np.int64)
    A = torch.zeros(S, gph.n_node, gph.n_node, dtype=torch.float64)
    A[:, rows, cols] = mask                                       # the diagonal entries that got a 0 are restored by set_diag
    assert float(A[:, np.arange(gph.n_node), np.arange(gph.n_node)].min()) == 0
    rx, re = OD.spatial_layer_dense(x, e, p, A, torch.from_numpy(gph.edge_adj.to_dense()), ne)
    r0x, _ = OD.spatial_layer_dense(x, e, p, adj, torch.from_numpy(gph.edge_adj.to_dense()), ne)
    assert float((rx - r0x).abs().max()) > 1000 * tol
    layer = load_spatial_layer(U.SpatialLayer(gph, d, 'relu', sparse_params=False, precision=precision), p, dev)
    ox, oe = layer(x.float().to(dev), e.float().to(dev), adj_mask=mask.float().to(dev))
    close(ox, rx, tol); close(oe, re, tol)
    ones = torch.ones(S, gph.adj.nnz, device=dev)                 # an all-ones mask is the plain layer
    ox1, _ = layer(x.float().to(dev), e.float().to(dev), adj_mask=ones)
    close(ox1, r0x, tol)


@pytest.mark.parametrize('R,M,S,h', [(1, 1, 1, 4), (130, 77, 3, 32), (257, 300, 5, 32), (64, 1000, 2, 64), (300, 129, 7, 12),
                                     # the LDS-DMA staged form (k_remainder_gemm2; S * h >= 256 and R >= 128): 256 x 128 tiles with ragged
                                     # edges on both sides, two k-steps / one k-step / many, and a shape whose round count picks 256 x 256 tiles
                                     (130, 64, 8, 32), (1000, 777, 9, 32), (443, 444, 5, 64), (129, 31, 37, 12), (8192, 300, 64, 32),
                                     # 272 tiles: one whole round of 256 + 16 tiles cut along K into four pieces each, in one launch + the reduce
                                     (8704, 1000, 64, 32)])
def test_remainder_gemm_ragged_shapes(dev, R, M, S, h):
    """uds_remainder_forward (the dense off-support part of a trained NodeEdge, emulator.py:44) against the fp64 product:
    shapes that are not multiples of the 128 x 128 (or 256 x 128 / 256 x 256) tile, of the k-step of 32, or of one snapshot per tile
    column block."""
    g = torch.Generator().manual_seed(R * 7 + M)
    rest, x = rnd(g, R, M), rnd(g, S, M, h)
    ref = torch.matmul(rest, x)
    packed = _lib.remainder_pack(rest.float().to(dev))
    out = _lib.remainder_forward(packed, (R, M), x.float().to(dev))
    assert tuple(out.shape) == (S, R, h)
    close(out, ref, TOL_BF16X3)
    out4 = _lib.remainder_forward(packed, (R, M), x.float().to(dev).reshape(1, S, M, h))     # leading dims are kept
    assert torch.equal(out4[0], out)


def test_spatial_block_c1_wide_first_layer(dev):
    """Block 2 of the reference starts from H + d/2 features (`emulator.py:260-262`)."""
    gph = U.DrainageGraph.from_edges(U.synthetic_drainage_network(50, 60, 0))
    d, fx = 8, 12
    ps = [spatial_params(50, 60, fx, fx, d, seed=1), spatial_params(50, 60, d, d, d, seed=2)]
    g = torch.Generator().manual_seed(9)
    x, e = rnd(g, 2, 50, fx), rnd(g, 2, 60, fx)
    ne = torch.from_numpy(gph.inc_n.to_dense())
    rx, re = OD.spatial_block_dense(x, e, ps, torch.from_numpy(gph.adj.to_dense()), torch.from_numpy(gph.edge_adj.to_dense()), ne)
    block = U.SpatialBlock(gph, d, 2, 'relu', fx=fx, fe=fx, sparse_params=False)
    for layer, p in zip(block.layers, ps):
        load_spatial_layer(layer, p, dev)
    ox, oe = block(x.float().to(dev), e.float().to(dev))
    close(ox, rx); close(oe, re)


@pytest.mark.parametrize('precision', ['fp32', 'bf16x3'])
@pytest.mark.parametrize('fx,fe', [(64, 64), (96, 96), (96, 64), (64, 96)])
def test_spatial_layer_c2_size_vs_sparse_oracle(dev, precision, fx, fe):
    """C2 scale (2k nodes / 2.5k conduits, d=64) incl. the wide first layer of block 2 (`emulator.py:260-262`:
    H + d/2 = 96 input features on either side): all four fused-kernel instantiations."""
    gph = U.DrainageGraph.from_edges(U.synthetic_drainage_network(2000, 2500, 0))
    d, S = 64, 3
    p = spatial_params(2000, 2500, fx, fe, d, seed=11, dense_ne=False, nnz_n=gph.inc_n.nnz, nnz_e=gph.inc_e.nnz)
    g = torch.Generator().manual_seed(12)
    x, e = rnd(g, S, 2000, fx), rnd(g, S, 2500, fe)
    rx, re = OS.spatial_layer_csr(x, e, p, (gph.adj.rowptr, gph.adj.col), (gph.edge_adj.rowptr, gph.edge_adj.col),
                                  (gph.inc_n.rowptr, gph.inc_n.col), (gph.inc_e.rowptr, gph.inc_e.col))
    layer = load_spatial_layer(U.SpatialLayer(gph, d, 'relu', fx=fx, fe=fe, sparse_params=True, precision=precision), p, dev)
    ox, oe = layer(x.float().to(dev), e.float().to(dev))
    close(ox, rx, PREC_TOL[precision]); close(oe, re, PREC_TOL[precision])
    # 96-wide inputs given as [64 | 32] from two tensors (uds_spatial_layer_forward_split; `concat([x, b])` of
    # emulator.py:260-262 never materialised): bit-identical to the concatenated call
    xd, ed = x.float().to(dev), e.float().to(dev)
    xa, xb = (xd[..., :64].contiguous(), xd[..., 64:].contiguous()) if fx == 96 else (xd, None)
    ea, eb = (ed[..., :64].contiguous(), ed[..., 64:].contiguous()) if fe == 96 else (ed, None)
    if xb is not None or eb is not None:
        sx, se = layer(xa, ea, xb, eb)
        assert torch.equal(sx, ox) and torch.equal(se, oe)


def test_wide_layer_d128_matrix_core_unfused_path(dev):
    """d = 128 (the reference's default embed_size, utils/config.yaml): no fused kernel; with precision='bf16x3' the dense
    parts run on the matrix-core row GEMM + uds_gat_aggregate."""
    gph = U.DrainageGraph.from_edges(U.synthetic_drainage_network(2000, 2500, 0))
    d, S = 128, 3
    p = spatial_params(2000, 2500, d, d, d, seed=5, dense_ne=False, nnz_n=gph.inc_n.nnz, nnz_e=gph.inc_e.nnz)
    g = torch.Generator().manual_seed(6)
    x, e = rnd(g, S, 2000, d), rnd(g, S, 2500, d)
    rx, re = OS.spatial_layer_csr(x, e, p, (gph.adj.rowptr, gph.adj.col), (gph.edge_adj.rowptr, gph.edge_adj.col),
                                  (gph.inc_n.rowptr, gph.inc_n.col), (gph.inc_e.rowptr, gph.inc_e.col))
    for precision in ('bf16x3', 'fp32'):
        layer = load_spatial_layer(U.SpatialLayer(gph, d, 'relu', sparse_params=True, precision=precision), p, dev)
        ox, oe = layer(x.float().to(dev), e.float().to(dev))
        close(ox, rx, PREC_TOL[precision]); close(oe, re, PREC_TOL[precision])
    hx, ss, sn = rnd(g, S, 2000, d).float().to(dev), rnd(g, S, 2000).float().to(dev), rnd(g, S, 2000).float().to(dev)
    with pytest.raises(_lib.UdsError):
        _lib.gat_aggregate(_lib.CsrHandle(gph.adj), hx, ss[:, :5].contiguous(), sn)


def test_fused_d128_mixed_width_layer(dev):
    """First layer of block 2 at d = 128 without actions (emulator.py:260-262): 128-wide node rows, 64-wide link rows -> the
    <128,64> (node tiles) and <64,128> (link tiles) variants of k_fused128, one launch each."""
    gph = U.DrainageGraph.from_edges(U.synthetic_drainage_network(2000, 2500, 0))
    d, S, fx, fe = 128, 3, 128, 64
    p = spatial_params(2000, 2500, fx, fe, d, seed=13, dense_ne=False, nnz_n=gph.inc_n.nnz, nnz_e=gph.inc_e.nnz)
    g = torch.Generator().manual_seed(14)
    x, e = rnd(g, S, 2000, fx), rnd(g, S, 2500, fe)
    rx, re = OS.spatial_layer_csr(x, e, p, (gph.adj.rowptr, gph.adj.col), (gph.edge_adj.rowptr, gph.edge_adj.col),
                                  (gph.inc_n.rowptr, gph.inc_n.col), (gph.inc_e.rowptr, gph.inc_e.col))
    layer = load_spatial_layer(U.SpatialLayer(gph, d, 'relu', fx=fx, fe=fe, sparse_params=True, precision='bf16x3'), p, dev)
    ox, oe = layer(x.float().to(dev), e.float().to(dev))
    assert layer.network().plan_info()['fused'] & 16
    close(ox, rx, TOL_BF16X3); close(oe, re, TOL_BF16X3)


def test_dense_wide_outputs_column_blocks(dev):
    """Dense 64 -> 128 / 96 -> 80 with precision='bf16x3': 64-column blocks of the kernel through uds_rowgemm_forward_cat."""
    g = torch.Generator().manual_seed(2)
    for fi, fo, act in ((64, 128, 'relu'), (96, 80, 'tanh')):
        x, k, b = rnd(g, 3, 1500, fi) - 0.5, rnd(g, fi, fo) - 0.5, rnd(g, fo) - 0.5
        m = U.Dense(fo, act, in_features=fi, precision='bf16x3').to(dev)
        m.kernel.data, m.bias.data = k.float().to(dev), b.float().to(dev)
        close(m(x.float().to(dev)), OD.dense(x, k, b, act), TOL_ROWGEMM)


@pytest.mark.parametrize('case', ['hub', 'astlingen', 'chaohu', 'tanh', 'c3'])
def test_fused_d128_kernel_cases(dev, networks, case):
    """k_fused128 (d = 128, h = 64): rows with more than 16 neighbours (the per-lane walk), small real networks (a single
    tile, ragged blocks, isolated nodes / parallel links), a run-time activation, and a 50k-node network (many tiles,
    several snapshot chunks); run twice: bitwise identical (the per-wave score partials are added in a fixed order)."""
    d, S, act = 128, 3, 'relu'
    if case == 'hub':
        edges = np.array([[0, i] for i in range(1, 31)] + [[i, i + 1] for i in range(30, 45)])
        n = 46
    elif case == 'c3':
        n, S = 50000, 2
        edges = U.synthetic_drainage_network(n, 65000, 0)
    else:
        net = networks['astlingen' if case == 'tanh' else case]
        edges, n = np.array(net['edges']), net['n_node']
        act = 'tanh' if case == 'tanh' else 'relu'
    gph = U.DrainageGraph.from_edges(edges, n)
    p = spatial_params(gph.n_node, gph.n_edge, d, d, d, seed=3, dense_ne=False, nnz_n=gph.inc_n.nnz, nnz_e=gph.inc_e.nnz)
    g = torch.Generator().manual_seed(4)
    x, e = rnd(g, S, gph.n_node, d), rnd(g, S, gph.n_edge, d)
    rx, re = OS.spatial_layer_csr(x.float() if case == 'c3' else x, e.float() if case == 'c3' else e, cast(p, torch.float32) if case == 'c3' else p,
                                  (gph.adj.rowptr, gph.adj.col), (gph.edge_adj.rowptr, gph.edge_adj.col),
                                  (gph.inc_n.rowptr, gph.inc_n.col), (gph.inc_e.rowptr, gph.inc_e.col), act=act)
    layer = load_spatial_layer(U.SpatialLayer(gph, d, act, sparse_params=True, precision='bf16x3'), p, dev)
    ox, oe = layer(x.float().to(dev), e.float().to(dev))
    assert layer.network().plan_info()['fused'] & 8             # the d = 128 tile plan exists: the fused kernel ran
    tol = 2e-5 if case == 'c3' else TOL_BF16X3                  # c3: fp32 oracle (the fp64 one needs minutes at this size)
    close(ox, rx.double(), tol); close(oe, re.double(), tol)
    ox2, oe2 = layer(x.float().to(dev), e.float().to(dev))
    assert torch.equal(ox, ox2) and torch.equal(oe, oe2)


def test_split_input_needs_the_fused_kernel(dev):
    gph = U.DrainageGraph.from_edges(U.synthetic_drainage_network(300, 360, 0))
    layer = U.SpatialLayer(gph, 64, 'relu', fx=96, sparse_params=True, generator=torch.Generator().manual_seed(1)).to(dev)
    x, xb, e = torch.rand(2, 300, 64, device=dev), torch.rand(2, 300, 32, device=dev), torch.rand(2, 360, 64, device=dev)
    vn, _ = layer.node_edge_n.support_values(); ve, _ = layer.node_edge_e.support_values()
    p = {k: (v.to(dev) if v is not None else None) for k, v in layer.export_params().items() if not k.startswith('ne_')}
    p.update(ne_n_val=vn, ne_e_val=ve)
    with pytest.raises(_lib.UdsError):      # exact-fp32 (unfused) path + split rows: refused, not silently concatenated
        _lib.spatial_layer_forward(layer.network(), p, x, e, 32, 64, 'relu', _lib.PRECISION_FLAGS['fp32'], xb=xb)
    with pytest.raises(_lib.UdsError):      # a split must be 64 + 32 columns
        _lib.spatial_layer_forward(layer.network(), p, x, e, 32, 64, 'relu', 0, xb=torch.rand(2, 300, 16, device=dev))


def test_fused_kernel_is_required_and_used(dev):
    """UDS_FLAG_REQUIRE_FUSED: the call fails instead of silently falling back."""
    gph = U.DrainageGraph.from_edges(U.synthetic_drainage_network(300, 360, 0))
    layer = U.SpatialLayer(gph, 64, 'relu', sparse_params=True, generator=torch.Generator().manual_seed(1)).to(dev)
    x, e = torch.rand(2, 300, 64, device=dev), torch.rand(2, 360, 64, device=dev)
    vn, _ = layer.node_edge_n.support_values(); ve, _ = layer.node_edge_e.support_values()
    p = {k: v.to(dev) if v is not None else None for k, v in layer.export_params().items() if not k.startswith('ne_')}
    p.update(ne_n_val=vn, ne_e_val=ve)
    ox, _ = _lib.spatial_layer_forward(layer.network(), p, x, e, 32, 64, 'relu', _lib.FLAG_REQUIRE_FUSED)
    assert torch.equal(ox, layer(x, e)[0])
    with pytest.raises(_lib.UdsError, match='fused kernel unavailable'):
        _lib.spatial_layer_forward(layer.network(), p, x, e, 32, 64, 'relu', _lib.FLAG_REQUIRE_FUSED | _lib.FLAG_EXACT_FP32)


def test_headline_size_properties(dev):
    """Full BASELINE size (10k / 12k, d=64): size-independent properties instead of the slow oracle.
    (1) snapshots are independent: running one snapshot alone gives the same bits;
    (2) two runs are bitwise identical (no atomics anywhere);
    (3) TWO full snapshots -- one inside the first chunk of snapshots a workgroup handles, the last one of the launch --
        against the fp64 sparse oracle (all rows of both sides)."""
    gph = U.DrainageGraph.from_edges(U.synthetic_drainage_network(10000, 12000, 0))
    d, S = 64, 9
    layer = U.SpatialLayer(gph, d, 'relu', sparse_params=True, generator=torch.Generator().manual_seed(1)).to(dev)
    g = torch.Generator().manual_seed(2)
    x, e = torch.rand(S, 10000, d, generator=g).to(dev), torch.rand(S, 12000, d, generator=g).to(dev)
    ox, oe = layer(x, e)
    ox2, oe2 = layer(x, e)
    assert torch.equal(ox, ox2) and torch.equal(oe, oe2)
    o3x, o3e = layer(x[3:4].contiguous(), e[3:4].contiguous())
    assert torch.equal(o3x[0], ox[3]) and torch.equal(o3e[0], oe[3])
    p = cast(layer.export_params(), torch.float64)
    sel = [3, S - 1]
    rx, re = OS.spatial_layer_csr(x[sel].double().cpu(), e[sel].double().cpu(), p, (gph.adj.rowptr, gph.adj.col),
                                  (gph.edge_adj.rowptr, gph.edge_adj.col), (gph.inc_n.rowptr, gph.inc_n.col),
                                  (gph.inc_e.rowptr, gph.inc_e.col))
    close(ox[sel], rx, TOL_BF16X3); close(oe[sel], re, TOL_BF16X3)
    assert bool(torch.isfinite(ox).all()) and bool((ox >= 0).all())
    # (4) the fused split-bf16 kernel and the exact-fp32 unfused kernels agree on the FULL tensors
    exact = U.SpatialLayer(gph, d, 'relu', sparse_params=True, precision='fp32').to(dev)
    exact.load_state_dict(layer.state_dict())
    fx_, fe_ = exact(x, e)
    close(fx_[sel], rx); close(fe_[sel], re)
    assert float((fx_ - ox).abs().max()) <= TOL_BF16X3 * max(1.0, float(fx_.abs().max()))
    assert float((fe_ - oe).abs().max()) <= TOL_BF16X3 * max(1.0, float(fe_.abs().max()))


@pytest.mark.parametrize('precision', ['fp32', 'bf16x3'])
def test_hub_network_rows_with_more_than_16_neighbours(dev, precision):
    """A hub junction with 30 conduits plus a chain: node rows with 31 neighbours and link rows with 30+ line-graph
    neighbours take the fused kernel's long-list path (every lane walks the whole list) -- real drainage networks
    never do (max degree 11 in the shipped ones), so it needs its own case."""
    hub = [[0, i] for i in range(1, 31)]
    chain = [[i, i + 1] for i in range(30, 90)]
    edges = np.array(hub + chain)
    gph = U.DrainageGraph.from_edges(edges)
    assert gph.adj.degrees().max() == 31 and gph.edge_adj.degrees().max() >= 30
    d, S = 64, 3
    p = spatial_params(gph.n_node, gph.n_edge, d, d, d, seed=21)
    g = torch.Generator().manual_seed(22)
    x, e = rnd(g, S, gph.n_node, d), rnd(g, S, gph.n_edge, d)
    rx, re = OD.spatial_layer_dense(x, e, p, torch.from_numpy(gph.adj.to_dense()), torch.from_numpy(gph.edge_adj.to_dense()),
                                    torch.from_numpy(gph.inc_n.to_dense()))
    layer = load_spatial_layer(U.SpatialLayer(gph, d, 'relu', sparse_params=False, precision=precision), p, dev)
    ox, oe = layer(x.float().to(dev), e.float().to(dev))
    close(ox, rx, PREC_TOL[precision]); close(oe, re, PREC_TOL[precision])


def test_c3_size_properties(dev):
    """C3 scale (50k junctions / 65k conduits, d=64): same size-independent properties as the headline test."""
    gph = U.DrainageGraph.from_edges(U.synthetic_drainage_network(50000, 65000, 0))
    d, S = 64, 4
    layer = U.SpatialLayer(gph, d, 'relu', sparse_params=True, generator=torch.Generator().manual_seed(1)).to(dev)
    g = torch.Generator().manual_seed(2)
    x, e = torch.rand(S, 50000, d, generator=g).to(dev), torch.rand(S, 65000, d, generator=g).to(dev)
    ox, oe = layer(x, e)
    ox2, oe2 = layer(x, e)
    assert torch.equal(ox, ox2) and torch.equal(oe, oe2)                     # bitwise reproducible
    o1x, o1e = layer(x[1:2].contiguous(), e[1:2].contiguous())
    assert torch.equal(o1x[0], ox[1]) and torch.equal(o1e[0], oe[1])          # snapshots independent
    p = cast(layer.export_params(), torch.float64)
    sel = [1, S - 1]
    rx, re = OS.spatial_layer_csr(x[sel].double().cpu(), e[sel].double().cpu(), p, (gph.adj.rowptr, gph.adj.col),
                                  (gph.edge_adj.rowptr, gph.edge_adj.col), (gph.inc_n.rowptr, gph.inc_n.col),
                                  (gph.inc_e.rowptr, gph.inc_e.col))
    close(ox[sel], rx, TOL_BF16X3); close(oe[sel], re, TOL_BF16X3)      # two full snapshots against the fp64 oracle


def test_cpu_tensors_are_refused(dev):
    layer = U.Dense(4, 'relu', in_features=4)
    with pytest.raises(_lib.UdsError):
        layer(torch.zeros(2, 4))


def test_spatial_block_graphed_equals_eager(dev):
    """`SpatialBlock.graphed`: the block forward replayed from one captured HIP graph reads its input buffers in place and gives
    the eager result bit for bit, also after the inputs were overwritten."""
    gph = U.DrainageGraph.from_edges(U.synthetic_drainage_network(600, 720, 0))
    block = U.SpatialBlock(gph, 64, 2, 'relu', sparse_params=True, generator=torch.Generator().manual_seed(1)).to(dev)
    g = torch.Generator().manual_seed(3)
    x, e = torch.rand(4, 600, 64, generator=g).to(dev), torch.rand(4, 720, 64, generator=g).to(dev)
    replay = block.graphed(x, e)
    for _ in range(2):
        with torch.no_grad():
            ex, ee = block(x, e)
        ox, oe = replay()
        assert torch.equal(ox, ex) and torch.equal(oe, ee)
        x.copy_(torch.rand(4, 600, 64, generator=g))          # new snapshots into the same buffers
        e.copy_(torch.rand(4, 720, 64, generator=g))


@pytest.mark.parametrize('d', [64, 128])
def test_small_network_snapshots_share_a_tile(dev, d):
    """A 30-node network fills a quarter of a 128-row tile: with many snapshots the layer lays k = 4 of them side by side as
    disjoint copies of the network (DrainageGraph.replicated; (S, N, F) -> (S / 4, 4 N, F) is a view).  S = 131 = 32 packed
    groups + 3 left-over snapshots: against the fp64 oracle, and against the same layer with packing switched off."""
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'networks.json')) as fh:
        net = json.load(fh)['astlingen']
    g = U.DrainageGraph.from_edges(np.array(net['edges']), net['n_node'])
    S = 131
    layer = U.SpatialLayer(g, d, 'relu', sparse_params=True, generator=torch.Generator().manual_seed(4)).to(dev)
    assert layer.pack_factor() == 4
    gen = torch.Generator().manual_seed(5)
    x, e = torch.rand(S, g.n_node, d, generator=gen), torch.rand(S, g.n_edge, d, generator=gen)
    ox, oe = layer(x.to(dev), e.to(dev))
    assert layer.last_path == 'fused' and layer._rep[0] == 4
    p = cast(layer.export_params(), torch.float64)
    rx, re = OS.spatial_layer_csr(x.double(), e.double(), p, (g.adj.rowptr, g.adj.col), (g.edge_adj.rowptr, g.edge_adj.col),
                                  (g.inc_n.rowptr, g.inc_n.col), (g.inc_e.rowptr, g.inc_e.col))
    close(ox, rx, TOL_BF16X3)
    close(oe, re, TOL_BF16X3)
    layer.PACK_ROWS = 0                      # packing off: one snapshot per tile
    px, pe = layer(x.to(dev), e.to(dev))
    assert layer.pack_factor() == 1
    assert float((px - ox).abs().max()) <= 4e-6 * max(1.0, float(rx.abs().max())) and float((pe - oe).abs().max()) <= 4e-6 * max(1.0, float(re.abs().max()))


@pytest.mark.parametrize('R,M,S,F,h,act', [(130, 77, 3, 64, 32, 'relu'), (443, 444, 9, 128, 64, 'tanh'), (300, 1000, 5, 64, 32, 'linear'),
                                          (1000, 129, 4, 128, 64, 'relu')])
def test_remainder_gemm_with_the_dense_layer_on_the_way(dev, R, M, S, F, h, act):
    """uds_remainder_forward_dense: rest @ act(e W + b) with the Dense output written straight into the GEMM's bf16 operand planes
    (no fp32 x_e in between), against the fp64 product and against the two-step path it replaces (Dense, then uds_remainder_forward:
    the same split values reach the same GEMM -- equal bit for bit); M not a multiple of 64 (zero padding of the planes), ragged R."""
    g = torch.Generator().manual_seed(R + M)
    rest, e = rnd(g, R, M), rnd(g, S, M, F)
    W, b = rnd(g, F, h) * 0.2, rnd(g, h) * 0.1
    ref = torch.matmul(rest, OD.activation(act)(e @ W + b))
    dense = U.Dense(h, activation=act, in_features=F, precision='bf16x3').to(dev)
    dense.kernel.data, dense.bias.data = W.float().to(dev), b.float().to(dev)
    packed = _lib.remainder_pack(rest.float().to(dev))
    from gnn_uds_amd.layers import _packed_kernel
    ed = e.float().to(dev)
    out = _lib.remainder_forward_dense(packed, (R, M), ed, _packed_kernel(dense, dense.kernel), dense.bias, act, h)
    assert tuple(out.shape) == (S, R, h)
    close(out, ref, TOL_BF16X3)
    two_step = _lib.remainder_forward(packed, (R, M), dense(ed))
    close(out, two_step.double().cpu(), 2e-6)      # (the Dense module may take another kernel at small sizes: not always bit-equal)
