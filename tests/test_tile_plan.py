"""Tile plans of the fused kernel: integer bookkeeping, checked on the host (no GPU) against the
definitions -- every row owned exactly once, halos = exactly the missing neighbours, local indices
resolve to the right global rows, weight positions address the global support arrays."""
import numpy as np
import pytest

import gnn_uds_amd as U
from gnn_uds_amd import _lib


def decode(hdr, pool, t):
    n_own, n_prim, n_sec, n_inc, n_adj, off, side, meta_len = (int(v) for v in hdr[t])
    p = off
    prim = pool[p:p + n_prim]; p += n_prim
    sec = pool[p:p + n_sec]; p += n_sec
    inc_ptr = pool[p:p + n_prim + 1]; p += n_prim + 1
    inc_loc = pool[p:p + n_inc]; p += n_inc
    inc_w = pool[p:p + n_inc]; p += n_inc
    adj_ptr = pool[p:p + n_own + 1]; p += n_own + 1
    adj_loc = pool[p:p + n_adj]; p += n_adj
    assert p - off == meta_len and off % 4 == 0
    return dict(n_own=n_own, prim=prim, sec=sec, inc_ptr=inc_ptr, inc_loc=inc_loc, inc_w=inc_w, adj_ptr=adj_ptr,
                adj_loc=adj_loc, side=side)


def check_plan(g, hdr, pool, caps, t_node, t_link):
    sides = [(g.adj, g.inc_n, t_node), (g.edge_adj, g.inc_e, t_link)]
    owned = [np.zeros(g.n_node, int), np.zeros(g.n_edge, int)]
    for t in range(len(hdr)):
        d = decode(hdr, pool, t)
        adj, inc, t_max = sides[d['side']]
        own = d['prim'][:d['n_own']]
        assert 1 <= d['n_own'] <= t_max and len(set(own.tolist())) == len(own)
        deg = (adj.rowptr[own + 1] - adj.rowptr[own]).astype(int)
        order = sorted(range(len(own)), key=lambda k: (-deg[k], own[k]))          # degree-sorted schedule, ties by id
        assert order == list(range(len(own)))
        owned[d['side']][own] += 1
        nb = np.unique(np.concatenate([adj.col[adj.rowptr[r]:adj.rowptr[r + 1]] for r in own]))
        halo = np.setdiff1d(nb, own)
        assert np.array_equal(d['prim'][d['n_own']:], halo)                 # halo = exactly the outside neighbours
        need = np.unique(np.concatenate([inc.col[inc.rowptr[r]:inc.rowptr[r + 1]] for r in d['prim']] + [np.zeros(0, np.int32)]))
        assert np.array_equal(d['sec'], need)                              # secondary rows = everything incident
        for i, r in enumerate(d['prim']):
            lo, hi = d['inc_ptr'][i], d['inc_ptr'][i + 1]
            assert np.array_equal(d['sec'][d['inc_loc'][lo:hi]], inc.col[inc.rowptr[r]:inc.rowptr[r + 1]])
            assert np.array_equal(d['inc_w'][lo:hi], np.arange(inc.rowptr[r], inc.rowptr[r + 1]))
        for i, r in enumerate(own):
            lo, hi = d['adj_ptr'][i], d['adj_ptr'][i + 1]
            assert np.array_equal(d['prim'][d['adj_loc'][lo:hi]], adj.col[adj.rowptr[r]:adj.rowptr[r + 1]])
        assert len(d['prim']) <= caps['p_cap'] and len(d['sec']) <= caps['q_cap'] and hdr[t, 7] <= caps['meta_cap']
    assert (owned[0] == 1).all() and (owned[1] == 1).all()                 # a partition of nodes and of links


@pytest.mark.parametrize('name', ['astlingen', 'shunqing', 'chaohu', 'hague', 'RedChicoSur'])
def test_plan_on_real_networks(networks, name):
    net = networks[name]
    g = U.DrainageGraph.from_edges(np.array(net['edges']), net['n_node'])
    for t_node, t_link, p_lim, q_lim in ((128, 120, 0, 0), (48, 48, 64, 80), (16, 12, 0, 0), (1, 1, 0, 0), (40, 40, 16, 16)):
        hdr, pool, caps = _lib.tile_plan(g, t_node, t_link, p_lim, q_lim)
        check_plan(g, hdr, pool, caps, t_node, t_link)
        if p_lim:                                                # limits hold unless a single row already exceeds them
            multi = hdr[:, 0] > 1
            assert (hdr[multi, 1] <= p_lim).all() and (hdr[multi, 2] <= q_lim).all()


def test_plan_with_isolated_rows_and_self_loop_link():
    e = np.array([[0, 1], [1, 1], [1, 2], [4, 2]])
    g = U.DrainageGraph.from_edges(e, n_node=6)            # nodes 3 and 5 isolated, link 1 self-referential
    hdr, pool, caps = _lib.tile_plan(g, 4, 4)
    check_plan(g, hdr, pool, caps, 4, 4)


def test_headline_plan_fits_the_cu_lds():
    """The plan uds_network_create builds for 64-float rows (own rows <= 128, footprint limits 128 primary / 208
    secondary rows): the agglomeration fills the limits (few, full tiles), halos stay a modest fraction, LDS <= 160 KiB
    (one 8-wave workgroup per CU)."""
    g = U.DrainageGraph.from_edges(U.synthetic_drainage_network(10000, 12000, 0))
    hdr, pool, caps = _lib.tile_plan(g, 128, 128, 128, 208)
    check_plan(g, hdr, pool, caps, 128, 128)
    assert caps['p_cap'] <= 128 and caps['q_cap'] <= 208
    for side, n in ((0, g.n_node), (1, g.n_edge)):
        h = hdr[hdr[:, 6] == side]
        assert h[:, 0].sum() == n
        assert h[:, 1].sum() <= 1.25 * n                   # prim rows (own + halo) per side
        assert h[:, 0].mean() >= 100                       # tiles are filled: a half-empty tile costs a workgroup the same time
        assert np.ceil(h[:, 1] / 16).mean() >= 7.5         # ~8 sixteen-row blocks for the 8 waves
    assert len(hdr) <= 215
    # LDS of the fused kernel at F = 64: meta + scalars + sec rows + hx rows + DMA stage of raw rows
    lds = 4 * (160 + caps["meta_cap"] + 2 * caps["p_cap"] + caps['q_cap'] * 36 + caps['p_cap'] * 64) + \
        256 * (caps['q_cap'] + caps['p_cap'])
    assert lds <= 160 * 1024
    a, b, c = _lib.tile_plan(g, 128, 128, 128, 208)
    assert np.array_equal(a, hdr) and np.array_equal(b, pool)   # deterministic


def a4(x):
    return (x + 3) & ~3


def check_blocks(hdr, pool, blocks):
    """The tile blocks of k_fused_tile (fixed-width index lists) hold exactly the plan's CSR lists."""
    for t in range(len(hdr)):
        d = decode(hdr, pool, t)
        b = blocks[t]
        n_own, n_prim, n_sec, flags, n_ovf, n_adj, side, width = (int(v) for v in b[:8])
        assert (n_own, n_prim, n_sec, side) == (d['n_own'], len(d['prim']), len(d['sec']), d['side'])
        p = 8
        assert np.array_equal(b[p:p + n_prim], d['prim']); p = a4(p + n_prim)
        assert np.array_equal(b[p:p + n_sec], d['sec']); p = a4(p + n_sec)
        locs = b[p:p + n_prim].view(np.uint32); p = a4(p + n_prim)
        w = b[p:p + 4 * n_prim].reshape(n_prim, 4); p += 4 * n_prim
        adj = b[p:p + 4 * n_own].view(np.uint8).reshape(n_own, 16); p += 4 * n_own
        inc_deg = np.diff(d['inc_ptr'])
        adj_deg = np.diff(d['adj_ptr'])
        assert width == min(4, inc_deg.max(initial=0))
        assert bool(flags & 1) == bool((inc_deg > 4).any()) and bool(flags & 2) == bool((adj_deg > 16).any())
        assert n_ovf == int(np.maximum(inc_deg - 4, 0).sum())
        if flags & 1:
            ovf_ptr = b[p:p + n_prim + 1]; p = a4(p + n_prim + 1)
            ovf_loc = b[p:p + n_ovf]; p = a4(p + n_ovf)
            ovf_w = b[p:p + n_ovf]; p = a4(p + n_ovf)
        for i in range(n_prim):
            lo, hi = d['inc_ptr'][i], d['inc_ptr'][i + 1]
            k = min(4, hi - lo)
            got = [(int(locs[i]) >> (8 * j)) & 0xff for j in range(4)]
            assert got[:k] == d['inc_loc'][lo:lo + k].tolist() and got[k:] == [0] * (4 - k)
            assert w[i, :k].tolist() == d['inc_w'][lo:lo + k].tolist() and (w[i, k:] == -1).all()
            if flags & 1:
                assert np.array_equal(ovf_loc[ovf_ptr[i]:ovf_ptr[i + 1]], d['inc_loc'][lo + k:hi])
                assert np.array_equal(ovf_w[ovf_ptr[i]:ovf_ptr[i + 1]], d['inc_w'][lo + k:hi])
        for i in range(n_own):
            lo, hi = d['adj_ptr'][i], d['adj_ptr'][i + 1]
            k = min(16, hi - lo)
            assert adj[i, :k].tolist() == d['adj_loc'][lo:lo + k].tolist() and (adj[i, k:] == 0xFF).all()
        if flags & 2:
            assert n_adj == len(d['adj_loc'])
            assert np.array_equal(b[p:p + n_own + 1], d['adj_ptr']); p = a4(p + n_own + 1)
            assert np.array_equal(b[p:p + n_adj], d['adj_loc']); p = a4(p + n_adj)
        assert p <= blocks.shape[1] and not b[p:].any()


@pytest.mark.parametrize('name', ['astlingen', 'shunqing', 'chaohu', 'hague', 'RedChicoSur'])
def test_tile_blocks_on_real_networks(networks, name):
    net = networks[name]
    g = U.DrainageGraph.from_edges(np.array(net['edges']), net['n_node'])
    for t, p_lim, q_lim in ((128, 128, 208), (16, 32, 48)):
        hdr, pool, caps = _lib.tile_plan(g, t, t, p_lim, q_lim, blocks=True)
        check_blocks(hdr, pool, caps['blocks'])


def test_tile_blocks_headline_and_hub():
    g = U.DrainageGraph.from_edges(U.synthetic_drainage_network(10000, 12000, 0))
    hdr, pool, caps = _lib.tile_plan(g, 128, 128, 128, 208, blocks=True)
    check_blocks(hdr, pool, caps['blocks'])
    assert caps['blocks'].shape[1] <= caps['meta_cap'] + 8          # no larger than the CSR form of the same lists
    # a hub junction (30 conduits) + chain: overflow of the four-wide incidence lists and rows with more than 16 neighbours
    edges = np.array([[0, i] for i in range(1, 31)] + [[i, i + 1] for i in range(30, 90)])
    g = U.DrainageGraph.from_edges(edges)
    hdr, pool, caps = _lib.tile_plan(g, 128, 128, 128, 208, blocks=True)
    check_blocks(hdr, pool, caps['blocks'])
    assert (caps['blocks'][:, 3] & 1).any() and (caps['blocks'][:, 3] & 2).any()
    # a tile beyond the byte-wide local indices: no blocks (the layer then runs the unfused kernels)
    star = np.array([[0, i] for i in range(1, 400)])
    g = U.DrainageGraph.from_edges(star)
    hdr, pool, caps = _lib.tile_plan(g, 128, 128, 128, 208, blocks=True)
    assert caps['blocks'] is None
