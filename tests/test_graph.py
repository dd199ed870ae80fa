"""Integer bookkeeping: the product's CSR builders vs the networkx oracle, bit for bit."""
import numpy as np
import pytest

from gnn_uds_amd import graph as G
from oracle import graphs as OG
from oracle import sparse_csr as OS

# networks on which the reference's own networkx calls are defined (chaohu has a parallel link, for
# which get_edge_adj raises inside networkx; shunqing has a link the directed line graph never sees)
DEFINED = {'RedChicoSur': (False, True), 'astlingen': (False, True), 'hague': (False, True),
           'chaohu': (), 'shunqing': (False,)}


def test_fixture_sizes(networks):
    sizes = {k: (v['n_node'], len(v['edges'])) for k, v in networks.items()}
    assert sizes == {'RedChicoSur': (443, 444), 'astlingen': (30, 29), 'chaohu': (140, 141),
                     'hague': (210, 210), 'shunqing': (113, 131)}


@pytest.mark.parametrize('name', sorted(DEFINED))
@pytest.mark.parametrize('order', [0, 1, 2, 3])
def test_adjacency_matches_networkx(networks, name, order):
    e = np.array(networks[name]['edges'])
    for directed in (False, True):
        ref = OG.adjacency(e, directed, 0, order)
        csr = G.adjacency_csr(e, None, directed, order)
        assert np.array_equal(ref, csr.to_dense())
        rowptr, col, _ = OS.csr_from_dense(OG.gat_filter(ref), add_self_loops=True)
        assert np.array_equal(rowptr, csr.rowptr) and np.array_equal(col, csr.col)
        assert csr.rowptr.dtype == np.int32 and csr.col.dtype == np.int32


@pytest.mark.parametrize('name', sorted(DEFINED))
@pytest.mark.parametrize('order', [0, 1, 2, 3])
def test_edge_adjacency_matches_networkx(networks, name, order):
    e = np.array(networks[name]['edges'])
    for directed in DEFINED[name]:
        ref = OG.edge_adjacency(e, directed, 0, order)
        csr = G.edge_adjacency_csr(e, directed, order)
        assert np.array_equal(ref, csr.to_dense())
        rowptr, col, _ = OS.csr_from_dense(OG.gat_filter(ref), add_self_loops=True)
        assert np.array_equal(rowptr, csr.rowptr) and np.array_equal(col, csr.col)


@pytest.mark.parametrize('name', sorted(DEFINED))
def test_incidence_matches_reference_definition(networks, name):
    net = networks[name]
    e = np.array(net['edges'])
    ref = OG.node_edge_incidence(net['n_node'], e)
    inc_n, inc_e = G.incidence_csr(net['n_node'], e)
    assert np.array_equal(ref, inc_n.to_dense()) and np.array_equal(ref.T, inc_e.to_dense())
    assert np.array_equal(G.get_node_edge(net['n_node'], e), ref)
    # each link has exactly one +1 and one -1
    assert (inc_e.degrees() == 2).all() and (np.abs(inc_n.val) == 1).all()


def test_undefined_cases_keep_every_link(networks):
    """chaohu: the reference raises (parallel links collapse in nx.Graph); here both parallel links
    stay line-graph vertices adjacent to each other and to everything at their endpoints."""
    e = np.array(networks['chaohu']['edges'])
    with pytest.raises(Exception):
        OG.edge_adjacency(e)
    csr = G.edge_adjacency_csr(e)
    dense = csr.to_dense()
    assert np.array_equal(dense, dense.T) and (np.diag(dense) == 1).all()
    for i in range(len(e)):
        for j in range(len(e)):
            assert dense[i, j] == float(bool(set(e[i]) & set(e[j])))


def test_self_loop_link_and_isolated_node():
    e = np.array([[0, 1], [1, 1], [1, 2], [4, 2]])          # link 1 is self-referential, node 3 isolated
    g = G.DrainageGraph.from_edges(e, n_node=5)
    ne = g.inc_n.to_dense()
    assert (ne[:, 1] == 0).all() and g.inc_e.degrees().tolist() == [2, 0, 2, 2]
    assert g.adj.to_dense()[3].tolist() == [0, 0, 0, 1, 0]
    with pytest.raises(Exception):                          # the reference's networkx call raises on node 3
        OG.adjacency(e)
    e2 = np.array([[0, 1], [1, 1], [1, 2], [3, 2]])         # same shape without the isolated node
    assert np.array_equal(OG.adjacency(e2), G.adjacency_csr(e2).to_dense())
    assert np.array_equal(OG.edge_adjacency(e2), G.edge_adjacency_csr(e2).to_dense())
    assert g.edge_adj.to_dense()[1].tolist() == [1, 1, 1, 0]


def test_from_dense_equals_from_edges(networks):
    net = networks['shunqing']
    e = np.array(net['edges'])
    g1 = G.DrainageGraph.from_edges(e, net['n_node'])
    g2 = G.DrainageGraph.from_dense(OG.adjacency(e), OG.edge_adjacency(e), OG.node_edge_incidence(net['n_node'], e))
    for a, b in ((g1.adj, g2.adj), (g1.edge_adj, g2.edge_adj), (g1.inc_n, g2.inc_n), (g1.inc_e, g2.inc_e)):
        assert np.array_equal(a.rowptr, b.rowptr) and np.array_equal(a.col, b.col)
    assert np.array_equal(g1.inc_n.val, g2.inc_n.val)


def test_degree_sorted_schedule():
    e = np.array([[0, 1], [0, 2], [0, 3], [3, 4]])
    adj = G.adjacency_csr(e)
    assert adj.degrees().tolist() == [4, 2, 2, 3, 2]
    assert adj.degree_sorted_rows().tolist() == [0, 3, 1, 2, 4]


def test_synthetic_network_is_seeded_and_drainage_like():
    a = G.synthetic_drainage_network(2000, 2500, seed=0)
    b = G.synthetic_drainage_network(2000, 2500, seed=0)
    c = G.synthetic_drainage_network(2000, 2500, seed=1)
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    assert a.shape == (2500, 2) and a.dtype == np.int32 and a.max() == 1999
    g = G.DrainageGraph.from_edges(a)
    assert g.adj.nnz == 2000 + 2 * 2500                      # N + 2E (no parallel links, no self loops)
    assert g.adj.degrees().max() <= 7                        # self + max_degree 6
    assert len({(min(u, v), max(u, v)) for u, v in a}) == 2500 and (a[:, 0] != a[:, 1]).all()
    import networkx as nx
    assert nx.is_connected(nx.Graph(a.tolist()))
    # headline graph: sizes quoted in DESIGN.md / bench.py
    h = G.DrainageGraph.from_edges(G.synthetic_drainage_network(10000, 12000, seed=0))
    assert (h.adj.nnz, h.edge_adj.nnz, h.inc_n.nnz) == (34000, 63000, 24000)


@pytest.mark.parametrize('name', ['astlingen', 'shunqing', 'hague'])
@pytest.mark.parametrize('cutoff', [50.0, 200.0])
def test_gaussian_length_kernel_matches_networkx(networks, name, cutoff):
    """`length > 0` (base.py:370-380,396-425): Dijkstra balls with the Gaussian kernel exp(-(l/std)^2)."""
    net = networks[name]
    e, ln = np.array(net['edges']), np.array(net['lengths'])
    for directed in (False, True):
        ref = OG.adjacency(e, directed, cutoff, 1, ln)
        csr = G.adjacency_csr(e, None, directed, 1, cutoff, ln)
        assert np.array_equal(ref > 0, csr.to_dense() > 0) and np.allclose(ref, csr.to_dense(), rtol=1e-12, atol=0)
        if directed in DEFINED[name]:
            ref = OG.edge_adjacency(e, directed, cutoff, 1, ln)
            csr = G.edge_adjacency_csr(e, directed, 1, cutoff, ln)
            assert np.array_equal(ref > 0, csr.to_dense() > 0) and np.allclose(ref, csr.to_dense(), rtol=1e-12, atol=0)
    with pytest.raises(ValueError):
        G.adjacency_csr(e, length=cutoff)


@pytest.mark.parametrize('name', ['astlingen', 'shunqing', 'hague', 'RedChicoSur'])
def test_graph_base_adjacencies_match_the_networkx_oracle(networks, name):
    """`get_node_based_adj` / `get_edge_based_adj` (base.py:471-532, graph_base 1 / 2): nodes and links in ONE graph; the CSR
    builders equal the networkx restatement bit for bit for every direction / DFS depth."""
    net = networks[name]
    e = np.array(net['edges'])
    for directed in (False, True):
        for order in (0, 1, 2):
            a = G.node_based_adj_csr(e, None, directed, order)
            assert np.array_equal(a.to_dense(), OG.node_based_adjacency(e, directed, order))
            b = G.edge_based_adj_csr(e, None, directed, order)
            assert np.array_equal(b.to_dense(), OG.edge_based_adjacency(e, directed, order))
            assert a.n_rows == b.n_rows == int(e.max()) + 1 + len(e)
    # length > 0: Gaussian kernel over the Dijkstra ball of the combined graph (base.py:479-487,526-529), CSR == networkx
    rng = np.random.default_rng(3)
    lengths = 20.0 + 200.0 * rng.random(len(e))
    for directed in (False, True):
        for cutoff in (60.0, 250.0):
            a = G.node_based_adj_csr(e, None, directed, 1, cutoff, lengths)
            ra = OG.node_based_adjacency(e, directed, 1, cutoff, lengths)
            assert np.array_equal(a.to_dense() > 0, ra > 0) and np.allclose(a.to_dense(), ra, rtol=1e-12, atol=0)
            b = G.edge_based_adj_csr(e, None, directed, 1, cutoff, lengths)
            rb = OG.edge_based_adjacency(e, directed, 1, cutoff, lengths)
            assert np.array_equal(b.to_dense() > 0, rb > 0) and np.allclose(b.to_dense(), rb, rtol=1e-12, atol=0)
    with pytest.raises(ValueError):
        G.node_based_adj_csr(e, None, False, 1, length=100.0)
