"""Graph-matrix oracle (TEST INFRASTRUCTURE ONLY): the reference's networkx constructions,
restated from `surrogate/envs/scenario/base.py:367-439` on a plain link list.

Integer / index bookkeeping: the product's builders (gnn_uds_amd/graph.py) must
match these bit for bit.  Pinned by the link lists of the five SWMM `.inp` data
files the reference ships (tests/golden/networks.json) -- the matrices themselves
are "parity unpinned" (the reference holds no expected outputs for them).

networkx is what the reference itself calls (`base.py:5`); it is used here on
purpose, so traversal-order quirks (depth-limited DFS pre-order, `combinations`
over `Graph.edges(n)`) are inherited rather than re-guessed.
"""
from itertools import combinations

import networkx as nx
import numpy as np


def adjacency(edges, directed=False, length=0, order=1, lengths=None):
    """`get_adj` (`base.py:367-391`).  edges:(E,2) int [from,to].  Note n_node =
    edges.max()+1 as in the reference (`:373,384`)."""
    edges = np.asarray(edges)
    g = nx.DiGraph() if directed else nx.Graph()
    n_node = int(edges.max()) + 1
    adj = np.zeros((n_node, n_node))
    if length:
        sigma = np.std(lengths)
        for (u, v), ln in zip(edges, lengths):
            g.add_edge(int(u), int(v), length=ln)
        for n in range(n_node):
            reach = nx.single_source_dijkstra_path_length(g, n, weight='length', cutoff=length)
            for a, dist in reach.items():
                adj[n, a] = np.exp(-(dist / (sigma + 1e-5)) ** 2)
    else:
        for u, v in edges:
            g.add_edge(int(u), int(v))
        for n in range(n_node):
            ball = list(nx.dfs_preorder_nodes(g, n, order)) if order > 0 else [n]
            for a in ball:
                adj[n, a] = 1
                if not directed:
                    adj[a, n] = 1
    return adj


def line_graph(edges, directed=False, lengths=None):
    """The link graph EX of `get_edge_adj` (`base.py:394-417`): links are vertices,
    joined when they share a node (undirected) or chain in->out (directed)."""
    edges = np.asarray(edges)
    g = nx.DiGraph() if directed else nx.Graph()
    for i, (u, v) in enumerate(edges):
        g.add_edge(int(u), int(v), edge=i)
        if lengths is not None:
            g[int(u)][int(v)].update(length=lengths[i])
    ex = nx.DiGraph() if directed else nx.Graph()
    for n in g.nodes():
        if directed:
            pairs = [(p, q) for p in g.in_edges(n) for q in g.out_edges(n)]
        else:
            pairs = list(combinations(g.edges(n), 2))
        for (a, b), (c, d) in pairs:
            p, q = g[a][b], g[c][d]
            ex.add_edge(p['edge'], q['edge'])
            if lengths is not None:
                ex[p['edge']][q['edge']].update(length=(p['length'] + q['length']) / 2)
    return ex


def edge_adjacency(edges, directed=False, length=0, order=1, lengths=None):
    """`get_edge_adj` (`base.py:393-429`).  Not symmetrised explicitly (`:428`)."""
    edges = np.asarray(edges)
    ex = line_graph(edges, directed, lengths if length else None)
    n_edge = edges.shape[0]
    adj = np.zeros((n_edge, n_edge))
    sigma = np.std(lengths) if length else None
    for n in range(n_edge):
        if length:
            reach = nx.single_source_dijkstra_path_length(ex, n, weight='length', cutoff=length)
            for a, dist in reach.items():
                adj[n, a] = np.exp(-(dist / (sigma + 1e-5)) ** 2)
        else:
            for a in (list(nx.dfs_preorder_nodes(ex, n, order)) if order > 0 else [n]):
                adj[n, a] = 1
    return adj


def node_based_adjacency(edges, directed=False, order=1, length=0, lengths=None):
    """`get_node_based_adj` (`base.py:471-498`): nodes and links in one graph, vertex n_node + i = link i; with length > 0 the
    three edges of link i weigh lengths[i] / 2 and the entries are the Gaussian kernel over the Dijkstra ball (:479-487)."""
    edges = np.asarray(edges)
    n_node, n_edge = int(edges.max()) + 1, edges.shape[0]
    adj = np.zeros((n_node + n_edge, n_node + n_edge))
    g = nx.DiGraph() if directed else nx.Graph()
    if length:
        sigma = np.std(lengths)
        for i, ((u, v), ln) in enumerate(zip(edges, lengths)):
            g.add_edge(int(u), int(v), length=ln / 2)
            g.add_edge(int(u), n_node + i, length=ln / 2)
            g.add_edge(n_node + i, int(v), length=ln / 2)
        for n in range(n_node + n_edge):
            for a, dist in nx.single_source_dijkstra_path_length(g, n, weight='length', cutoff=length).items():
                adj[n, a] = np.exp(-(dist / (sigma + 1e-5)) ** 2)
        return adj
    for i, (u, v) in enumerate(edges):
        g.add_edge(int(u), int(v))
        g.add_edge(int(u), n_node + i)
        g.add_edge(n_node + i, int(v))
    for n in range(n_node + n_edge):
        for a in (list(nx.dfs_preorder_nodes(g, n, order)) if order > 0 else [n]):
            adj[n, a] = 1
            if not directed:
                adj[a, n] = 1
    return adj


def edge_based_adjacency(edges, directed=False, order=1, length=0, lengths=None):
    """`get_edge_based_adj` (`base.py:500-532`); length > 0: link pairs (l_p + l_q) / 2 apart, link -- end node l / 2, entries
    = Gaussian kernel over the Dijkstra ball (:526-529)."""
    from itertools import product
    edges = np.asarray(edges)
    n_node, n_edge = int(edges.max()) + 1, edges.shape[0]
    g = nx.DiGraph() if directed else nx.Graph()
    for i, (u, v) in enumerate(edges):
        g.add_edge(int(u), int(v), edge=n_node + i, length=lengths[i] if length else 0)
    ex = nx.DiGraph() if directed else nx.Graph()
    for n in g.nodes():
        pairs = product(g.in_edges(n), g.out_edges(n)) if directed else combinations(g.edges(n), 2)
        for (a, b), (c, d) in pairs:
            ex.add_edge(g[a][b]['edge'], g[c][d]['edge'], length=(g[a][b]['length'] + g[c][d]['length']) / 2 if length else 0)
        if directed:
            for a, b in g.in_edges(n):
                ex.add_edge(g[a][b]['edge'], n, length=g[a][b]['length'] / 2 if length else 0)
            for c, d in g.out_edges(n):
                ex.add_edge(n, g[c][d]['edge'], length=g[c][d]['length'] / 2 if length else 0)
        else:
            for a, b in g.edges(n):
                ex.add_edge(g[a][b]['edge'], n, length=g[a][b]['length'] / 2 if length else 0)
    adj = np.zeros((n_node + n_edge, n_node + n_edge))
    if length:
        sigma = np.std(lengths)
        for n in range(n_node + n_edge):
            for a, dist in nx.single_source_dijkstra_path_length(ex, n, weight='length', cutoff=length).items():
                adj[n, a] = np.exp(-(dist / (sigma + 1e-5)) ** 2)
        return adj
    for n in range(n_node + n_edge):
        for a in (list(nx.dfs_preorder_nodes(ex, n, order)) if order > 0 else [n]):
            adj[n, a] = 1
    return adj


def node_edge_incidence(n_node, edges):
    """`get_node_edge` (`base.py:432-439`): +1 at the from-node, -1 at the to-node
    (a self-referential link nets to 0)."""
    edges = np.asarray(edges)
    ne = np.zeros((n_node, len(edges)))
    for i, (u, v) in enumerate(edges):
        ne[u, i] += 1
        ne[v, i] += -1
    return ne


def gat_filter(adj):
    """`Emulator.get_conv` GAT branch (`emulator.py:143-145`): (adj>0).astype(int)."""
    return (np.asarray(adj) > 0).astype(int)
