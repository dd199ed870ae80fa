"""Drainage-network graph bookkeeping (host side, integer-exact).

Sparse counterparts of the reference's dense graph matrices
(`surrogate/envs/scenario/base.py:367-439`, consumed by `Emulator.get_conv`,
`surrogate/emulator.py:129-152`):

  get_adj        (N,N)  -> adjacency_csr        rows = nodes,  cols = nodes in the order-ball
  get_edge_adj   (E,E)  -> edge_adjacency_csr   rows = links,  cols = links sharing a node
  get_node_edge  (N,E)  -> incidence_csr        +1 from-node / -1 to-node

The reference materialises these as dense matrices (what blocks N >= 50k there);
here they are CSR from the start and a dense view exists only for small graphs.
Index arrays are int32, columns ascend inside a row, and the arrays are compared
bit for bit with the networkx oracle in tests/test_graph.py.

Where the reference is undefined (it raises inside networkx for a node / link that
touches nothing, and silently collapses parallel links in `nx.Graph`) this module
keeps every link: a link is adjacent to every link it shares an endpoint with, an
isolated vertex keeps its self loop only.
"""
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

I32 = np.int32


@dataclass
class CSR:
    """Row-major CSR pattern (+ optional float64 values); columns ascend per row."""
    rowptr: np.ndarray
    col: np.ndarray
    n_rows: int
    n_cols: int
    val: Optional[np.ndarray] = None

    @property
    def nnz(self):
        return int(self.col.shape[0])

    def rows(self):
        """Row index of every stored entry (int64)."""
        return np.repeat(np.arange(self.n_rows, dtype=np.int64), np.diff(self.rowptr.astype(np.int64)))

    def degrees(self):
        return np.diff(self.rowptr.astype(np.int64)).astype(I32)

    def to_dense(self, dtype=np.float64):
        out = np.zeros((self.n_rows, self.n_cols), dtype=dtype)
        out[self.rows(), self.col.astype(np.int64)] = 1.0 if self.val is None else self.val
        return out

    ORDER_WINDOW = 1024

    def degree_sorted_rows(self):
        """Row schedule used by the kernels: inside every window of ORDER_WINDOW consecutive rows, rows by descending
        degree, ties in row order (stable), so a wave's lanes see equal trip counts while neighbouring workgroups
        still work on neighbouring rows (a global sort scatters the gathers of a 2M-row batch over the whole array:
        every neighbour row then comes from HBM instead of L2)."""
        deg = self.degrees().astype(np.int64)
        out = np.empty(self.n_rows, dtype=I32)
        for r0 in range(0, self.n_rows, self.ORDER_WINDOW):
            r1 = min(self.n_rows, r0 + self.ORDER_WINDOW)
            out[r0:r1] = r0 + np.argsort(-deg[r0:r1], kind='stable')
        return out


def _csr_from_pairs(rows, cols, n_rows, n_cols, val=None):
    """Unique (row, col) pairs -> CSR, columns ascending."""
    rows = np.asarray(rows, dtype=np.int64)
    cols = np.asarray(cols, dtype=np.int64)
    key = rows * np.int64(n_cols) + cols
    key, first = np.unique(key, return_index=True)
    r = key // np.int64(n_cols)
    c = key - r * np.int64(n_cols)
    rowptr = np.zeros(n_rows + 1, dtype=np.int64)
    np.add.at(rowptr, r + 1, 1)
    rowptr = np.cumsum(rowptr)
    if rowptr[-1] >= 2 ** 31:
        raise ValueError('CSR with %d entries does not fit int32 indices' % rowptr[-1])
    v = None if val is None else np.asarray(val, dtype=np.float64)[first]
    return CSR(rowptr.astype(I32), c.astype(I32), int(n_rows), int(n_cols), v)


def csr_from_dense(a, add_self_loops=False, keep_values=False):
    """Non-zero pattern of a dense matrix; with add_self_loops the diagonal is set to
    one first (Spektral GATConv: tf.linalg.set_diag(a, 1), used via `emulator.py:229`)."""
    a = np.array(a, dtype=np.float64, copy=True)
    if a.ndim != 2:
        raise ValueError('expected a 2-D matrix, got shape %r' % (a.shape,))
    if add_self_loops:
        if a.shape[0] != a.shape[1]:
            raise ValueError('self loops need a square matrix')
        np.fill_diagonal(a, 1.0)
    rows, cols = np.nonzero(a)
    return _csr_from_pairs(rows, cols, a.shape[0], a.shape[1], a[rows, cols] if keep_values else None)


# ----------------------------------------------------------------------------------------------
# Ordered adjacency lists + depth-limited DFS pre-order (what `nx.dfs_preorder_nodes(X, n, order)`
# visits, `base.py:387,427`).  Neighbour order = insertion order, as in networkx's dict-of-dict.
# ----------------------------------------------------------------------------------------------
class _OrderedGraph:
    def __init__(self, directed=False):
        self.directed = directed
        self.succ = {}
        self.pred = {}

    def add_vertex(self, n):
        if n not in self.succ:
            self.succ[n] = {}
            self.pred[n] = {}

    def add_edge(self, u, v, **attr):
        self.add_vertex(u)
        self.add_vertex(v)
        self.succ[u].setdefault(v, {}).update(attr)
        if self.directed:
            self.pred[v].setdefault(u, self.succ[u][v])
        else:
            self.succ[v][u] = self.succ[u][v]

    def ball(self, source, depth_limit):
        """Vertices in depth-limited DFS pre-order from `source`."""
        if source not in self.succ:
            return [source]
        seen = {source}
        out = [source]
        stack = [iter(self.succ[source])]
        while stack:
            for child in stack[-1]:
                if child in seen:
                    continue
                seen.add(child)
                out.append(child)
                if len(stack) < depth_limit:
                    stack.append(iter(self.succ[child]))
                    break
            else:
                stack.pop()
        return out


def _check_edges(edges):
    edges = np.asarray(edges)
    if edges.ndim != 2 or edges.shape[1] != 2:
        raise ValueError('edges must have shape (E,2), got %r' % (edges.shape,))
    if edges.size and edges.min() < 0:
        raise ValueError('negative node index in edges')
    return edges.astype(np.int64)


def adjacency_csr(edges, n_node=None, directed=False, order=1, length=0, lengths=None):
    """Sparse `get_adj` (`base.py:367-391`): row n holds n and every node within the
    depth-`order` DFS ball of n; symmetrised unless `directed`.  Self entries are always
    present (the ball contains n), so Spektral's forced diagonal changes nothing."""
    edges = _check_edges(edges)
    if n_node is None:
        n_node = int(edges.max()) + 1 if edges.size else 0          # base.py:373,384
    if length:
        return _gaussian_ball_csr(n_node, edges, lengths, length, directed, line=False)
    u, v = edges[:, 0], edges[:, 1]
    me = np.arange(n_node, dtype=np.int64)
    if order <= 1:
        if order <= 0:
            rows, cols = me, me
        elif directed:
            rows, cols = np.concatenate([me, u]), np.concatenate([me, v])
        else:
            rows, cols = np.concatenate([me, u, v]), np.concatenate([me, v, u])
        return _csr_from_pairs(rows, cols, n_node, n_node)
    g = _OrderedGraph(directed)
    for a, b in edges:
        g.add_edge(int(a), int(b))
    rows, cols = [], []
    for n in range(n_node):
        for a in g.ball(n, order):
            rows.append(n)
            cols.append(a)
            if not directed:
                rows.append(a)
                cols.append(n)
    return _csr_from_pairs(rows, cols, n_node, n_node)


def _link_pairs_sharing_a_node(edges, directed):
    """(row link, col link) for every ordered pair of links meeting at a node.
    Undirected: any two links with a common endpoint.  Directed: in-link -> out-link."""
    n_edge = edges.shape[0]
    u, v = edges[:, 0], edges[:, 1]
    ids = np.arange(n_edge, dtype=np.int64)
    if directed:
        in_node, in_link = v, ids                    # link arrives at v
        out_node, out_link = u, ids                  # link leaves u
        o = np.argsort(out_node, kind='stable')
        out_node, out_link = out_node[o], out_link[o]
        n_node = int(edges.max()) + 1 if edges.size else 0
        start = np.searchsorted(out_node, np.arange(n_node + 1))
        cnt = (start[1:] - start[:-1])[in_node]
        rows = np.repeat(in_link, cnt)
        off = np.arange(cnt.sum()) - np.repeat(np.cumsum(cnt) - cnt, cnt)
        cols = out_link[np.repeat(start[:-1][in_node], cnt) + off]
        return rows, cols
    loop = u == v
    node = np.concatenate([u, v[~loop]])
    link = np.concatenate([ids, ids[~loop]])
    o = np.argsort(node, kind='stable')
    node, link = node[o], link[o]
    n_node = int(edges.max()) + 1 if edges.size else 0
    start = np.searchsorted(node, np.arange(n_node + 1))
    cnt = (start[1:] - start[:-1])[node]
    rows = np.repeat(link, cnt)
    off = np.arange(cnt.sum()) - np.repeat(np.cumsum(cnt) - cnt, cnt)
    cols = link[np.repeat(start[:-1][node], cnt) + off]
    return rows, cols


def edge_adjacency_csr(edges, directed=False, order=1, length=0, lengths=None):
    """Sparse `get_edge_adj` (`base.py:393-429`): the line graph (links sharing a node)
    and, per link, the depth-`order` DFS ball in it, self included."""
    edges = _check_edges(edges)
    n_edge = edges.shape[0]
    if length:
        return _gaussian_ball_csr(n_edge, edges, lengths, length, directed, line=True)
    me = np.arange(n_edge, dtype=np.int64)
    if order <= 0:
        return _csr_from_pairs(me, me, n_edge, n_edge)
    pr, pc = _link_pairs_sharing_a_node(edges, directed)
    if order == 1:
        return _csr_from_pairs(np.concatenate([me, pr]), np.concatenate([me, pc]), n_edge, n_edge)
    # order >= 2: neighbour order matters for a depth-limited DFS, so rebuild the line graph
    # with the reference's insertion order (nodes by first appearance, pairs by combinations).
    g = _OrderedGraph(directed)
    for i, (a, b) in enumerate(edges):
        g.add_edge(int(a), int(b), edge=i)
    ex = _OrderedGraph(directed)
    for n in list(g.succ):
        if directed:
            ins = [g.succ[a][n]['edge'] for a in g.pred[n]]
            outs = [g.succ[n][d]['edge'] for d in g.succ[n]]
            pairs = [(p, q) for p in ins for q in outs]
        else:
            inc = [g.succ[n][b]['edge'] for b in g.succ[n]]
            pairs = [(inc[i], inc[j]) for i in range(len(inc)) for j in range(i + 1, len(inc))]
        for p, q in pairs:
            ex.add_edge(p, q)
    rows, cols = [], []
    for n in range(n_edge):
        for a in ex.ball(n, order):
            rows.append(n)
            cols.append(a)
    return _csr_from_pairs(rows, cols, n_edge, n_edge)


def _gaussian_rows(g, n, sigma, cutoff):
    """Rows of exp(-(dist / (sigma + 1e-5))^2) over the Dijkstra ball (weight 'length', cutoff inclusive) of every vertex of
    the weighted `_OrderedGraph` g -- the `length > 0` loops of base.py:484-487,526-529.  A vertex without any edge keeps
    itself only (the reference would raise NodeNotFound for it)."""
    nbrs = {v: [(u, float(at['length'])) for u, at in adj.items()] for v, adj in g.succ.items()}
    rows, cols, vals = [], [], []
    for r in range(n):
        for a, dist in _dijkstra_ball(nbrs, r, float(cutoff)).items():
            v = float(np.exp(-(dist / (sigma + 1e-5)) ** 2))
            if v > 0.0:
                rows.append(r)
                cols.append(a)
                vals.append(v)
    return _csr_from_pairs(rows, cols, n, n, vals)


def node_based_adj_csr(edges, n_node=None, directed=False, order=1, length=0, lengths=None):
    """Sparse `get_node_based_adj` (`base.py:471-498`, `graph_base = 1`): ONE graph over the N nodes and the E links
    (vertex N + i = link i) with the edges (u, v), (u, link), (link, v) of every link; row n holds the depth-`order` DFS
    ball of n, symmetrised unless `directed`.  `length > 0` (needs `lengths`, one per link): every one of the three edges
    of link i weighs lengths[i] / 2 and row n holds the Gaussian kernel over the Dijkstra ball of radius `length`."""
    edges = _check_edges(edges)
    if n_node is None:
        n_node = int(edges.max()) + 1 if edges.size else 0
    n = n_node + edges.shape[0]
    if length:
        if lengths is None:
            raise ValueError('length > 0 needs the link lengths')
        lengths = np.asarray(lengths, dtype=np.float64)
        g = _OrderedGraph(directed)
        for i, ((u, v), ln) in enumerate(zip(edges, lengths)):
            g.add_edge(int(u), int(v), length=ln / 2)
            g.add_edge(int(u), n_node + i, length=ln / 2)
            g.add_edge(n_node + i, int(v), length=ln / 2)
        return _gaussian_rows(g, n, float(np.std(lengths)), length)
    me = np.arange(n, dtype=np.int64)
    if order <= 0:
        return _csr_from_pairs(me, me, n, n)
    g = _OrderedGraph(directed)
    for i, (u, v) in enumerate(edges):
        g.add_edge(int(u), int(v))
        g.add_edge(int(u), n_node + i)
        g.add_edge(n_node + i, int(v))
    rows, cols = [], []
    for r in range(n):
        for a in g.ball(r, order):
            rows.append(r)
            cols.append(a)
            if not directed:
                rows.append(a)
                cols.append(r)
    return _csr_from_pairs(rows, cols, n, n)


def edge_based_adj_csr(edges, n_node=None, directed=False, order=1, length=0, lengths=None):
    """Sparse `get_edge_based_adj` (`base.py:500-532`, `graph_base = 2`): the line graph (links meeting at a node) plus
    an edge between every link and its end nodes, over the N + E vertices; row n = depth-`order` DFS ball, NOT symmetrised
    (as the reference).  `length > 0` (needs `lengths`): two links meeting at a node are (l_p + l_q) / 2 apart, a link and
    its end node l / 2; row n holds the Gaussian kernel over the Dijkstra ball of radius `length`."""
    edges = _check_edges(edges)
    if n_node is None:
        n_node = int(edges.max()) + 1 if edges.size else 0
    n = n_node + edges.shape[0]
    me = np.arange(n, dtype=np.int64)
    if length and lengths is None:
        raise ValueError('length > 0 needs the link lengths')
    if order <= 0 and not length:
        return _csr_from_pairs(me, me, n, n)
    ln_of = np.asarray(lengths, dtype=np.float64) if length else np.zeros(edges.shape[0])
    g = _OrderedGraph(directed)
    for i, (u, v) in enumerate(edges):
        g.add_edge(int(u), int(v), edge=n_node + i, length=float(ln_of[i]))
    ex = _OrderedGraph(directed)
    for v in list(g.succ):
        if directed:
            ins = [g.succ[a][v] for a in g.pred[v]]
            outs = [g.succ[v][d] for d in g.succ[v]]
            for p in ins:
                for q in outs:
                    ex.add_edge(p['edge'], q['edge'], length=(p['length'] + q['length']) / 2)
            for p in ins:
                ex.add_edge(p['edge'], v, length=p['length'] / 2)
            for q in outs:
                ex.add_edge(v, q['edge'], length=q['length'] / 2)
        else:
            inc = [g.succ[v][b] for b in g.succ[v]]
            for i in range(len(inc)):
                for j in range(i + 1, len(inc)):
                    ex.add_edge(inc[i]['edge'], inc[j]['edge'], length=(inc[i]['length'] + inc[j]['length']) / 2)
            for p in inc:
                ex.add_edge(p['edge'], v, length=p['length'] / 2)
    if length:
        return _gaussian_rows(ex, n, float(np.std(ln_of)), length)
    rows, cols = [], []
    for r in range(n):
        for a in ex.ball(r, order):
            rows.append(r)
            cols.append(a)
    return _csr_from_pairs(rows, cols, n, n)


def _dijkstra_ball(nbrs, src, cutoff):
    """{vertex: distance} for every vertex within `cutoff` of src (inclusive), as
    nx.single_source_dijkstra_path_length(G, src, weight='length', cutoff=cutoff)."""
    import heapq
    dist = {}
    seen = {src: 0.0}
    heap = [(0.0, 0, src)]
    tick = 1
    while heap:
        d, _, v = heapq.heappop(heap)
        if v in dist:
            continue
        dist[v] = d
        for u, w in nbrs.get(v, ()):
            nd = d + w
            if nd > cutoff:
                continue
            if u in dist:
                continue
            if u not in seen or nd < seen[u]:
                seen[u] = nd
                heapq.heappush(heap, (nd, tick, u))
                tick += 1
    return dist


def _gaussian_ball_csr(n_vertices, edges, lengths, cutoff, directed, line):
    """`length > 0` branches of get_adj / get_edge_adj (`base.py:370-380,396-425`): every vertex within
    weighted distance `length` (Dijkstra, cutoff inclusive) gets exp(-(dist / (std(lengths) + 1e-5))^2).
    Node graph: link lengths (a later link between the same two nodes overrides an earlier one, as in
    nx.Graph.add_edge).  Line graph: two links sharing a node are (len_p + len_q) / 2 apart.
    Entries whose kernel value underflows to 0 are dropped (the GAT filter is `adj > 0`)."""
    if lengths is None:
        raise ValueError('length > 0 needs the link lengths')
    lengths = np.asarray(lengths, dtype=np.float64)
    sigma = float(np.std(lengths))
    nbrs = {}
    if not line:
        w = {}
        for (a, b), ln in zip(edges.tolist(), lengths.tolist()):
            w[(a, b)] = ln
            if not directed:
                w[(b, a)] = ln
        for (a, b), ln in w.items():
            nbrs.setdefault(a, []).append((b, ln))
    else:
        pr, pc = _link_pairs_sharing_a_node(edges, directed)
        keep = pr != pc
        for p, q in zip(pr[keep].tolist(), pc[keep].tolist()):
            nbrs.setdefault(p, []).append((q, (lengths[p] + lengths[q]) / 2))
    rows, cols, vals = [], [], []
    for n in range(n_vertices):
        for a, dist in _dijkstra_ball(nbrs, n, float(cutoff)).items():
            v = float(np.exp(-(dist / (sigma + 1e-5)) ** 2))
            if v > 0.0:
                rows.append(n)
                cols.append(a)
                vals.append(v)
    return _csr_from_pairs(rows, cols, n_vertices, n_vertices, vals)


def incidence_csr(n_node, edges):
    """Sparse `get_node_edge` (`base.py:432-439`).  Returns (node_side, link_side):
    node_side rows = nodes, cols = incident links, val = +1 (from-node) / -1 (to-node);
    link_side is its transpose.  A self-referential link nets to 0 and is absent."""
    edges = _check_edges(edges)
    n_edge = edges.shape[0]
    u, v = edges[:, 0], edges[:, 1]
    keep = u != v
    ids = np.arange(n_edge, dtype=np.int64)
    rows = np.concatenate([u[keep], v[keep]])
    cols = np.concatenate([ids[keep], ids[keep]])
    vals = np.concatenate([np.ones(keep.sum()), -np.ones(keep.sum())])
    node_side = _csr_from_pairs(rows, cols, n_node, n_edge, vals)
    link_side = _csr_from_pairs(cols, rows, n_edge, n_node, vals)
    return node_side, link_side


# Dense views under the reference's names, for small graphs / drop-in args -----------------------
def get_adj(edges, directed=False, length=0, order=1, lengths=None):
    return adjacency_csr(edges, None, directed, order, length, lengths).to_dense()


def get_edge_adj(edges, directed=False, length=0, order=1, lengths=None):
    return edge_adjacency_csr(edges, directed, order, length, lengths).to_dense()


def get_node_edge(n_node, edges):
    return incidence_csr(n_node, edges)[0].to_dense()


@dataclass
class DrainageGraph:
    """Everything the spatial layers need about one network, in CSR.

    adj / edge_adj carry the forced self loops (GAT's set_diag); inc_n is (N x E) and
    inc_e its transpose, values are the signed incidence (the layers use |value|)."""
    n_node: int
    n_edge: int
    edges: np.ndarray
    adj: CSR
    edge_adj: CSR
    inc_n: CSR
    inc_e: CSR
    meta: dict = field(default_factory=dict)

    def replicated(self, k):
        """k disjoint copies of this network as ONE network (block-diagonal patterns: node c*N + n, link c*E + e of copy c).
        (S, N, F) features of S = k * S' snapshots ARE (S', k*N, F) features of the replicated network, in place: this is
        how a small network (30-60 rows: a quarter of a 128-row tile) fills the tiles of the fused kernel."""
        k = int(k)

        def rep(c):
            rp = np.asarray(c.rowptr, dtype=np.int64)
            rowptr = np.concatenate([rp[:-1] + i * c.nnz for i in range(k)] + [np.array([k * c.nnz], dtype=np.int64)])
            col = np.concatenate([np.asarray(c.col, dtype=np.int64) + i * c.n_cols for i in range(k)])
            val = None if c.val is None else np.tile(np.asarray(c.val), k)
            return CSR(rowptr.astype(I32), col.astype(I32), k * c.n_rows, k * c.n_cols, val)
        edges = np.concatenate([np.asarray(self.edges, dtype=np.int64) + i * self.n_node for i in range(k)]).astype(I32)
        return DrainageGraph(k * self.n_node, k * self.n_edge, edges, rep(self.adj), rep(self.edge_adj), rep(self.inc_n), rep(self.inc_e),
                             dict(self.meta, copies=k))

    @classmethod
    def from_edges(cls, edges, n_node=None, directed=False, order=1, length=0, lengths=None):
        """Patterns of the GAT filters `(adj > 0)` with forced self loops (`emulator.py:143-145`) for the reference's
        `get_args(directed, length, order)` options (`base.py:277,321-328`)."""
        edges = _check_edges(edges)
        if n_node is None:
            n_node = int(edges.max()) + 1
        adj = adjacency_csr(edges, n_node, directed, order, length, lengths)
        eadj = edge_adjacency_csr(edges, directed, order, length, lengths)
        me_n, me_e = np.arange(n_node, dtype=np.int64), np.arange(edges.shape[0], dtype=np.int64)
        adj = _csr_from_pairs(np.concatenate([adj.rows(), me_n]), np.concatenate([adj.col.astype(np.int64), me_n]), n_node, n_node)
        eadj = _csr_from_pairs(np.concatenate([eadj.rows(), me_e]), np.concatenate([eadj.col.astype(np.int64), me_e]),
                               edges.shape[0], edges.shape[0])
        inc_n, inc_e = incidence_csr(n_node, edges)
        return cls(n_node, edges.shape[0], edges.astype(I32), adj, eadj, inc_n, inc_e,
                   dict(directed=directed, order=order, length=length))

    @classmethod
    def from_inp(cls, path):
        """The network of a SWMM `.inp` file (node / link order of the reference, `base.py:335-365`)."""
        from .inp import load_network
        return load_network(path).graph

    @classmethod
    def from_dense(cls, adj, edge_adj, node_edge, edges=None):
        """From the reference's dense `args.adj`, `args.edge_adj`, `args.node_edge`
        (`emulator.py:79,83,88`): filters are (m>0) with the diagonal forced (`:143-145`)."""
        node_edge = np.asarray(node_edge, dtype=np.float64)
        n_node, n_edge = node_edge.shape
        a = csr_from_dense(np.asarray(adj) > 0, add_self_loops=True)
        ea = csr_from_dense(np.asarray(edge_adj) > 0, add_self_loops=True)
        inc_n = csr_from_dense(node_edge, keep_values=True)
        inc_e = csr_from_dense(node_edge.T, keep_values=True)
        if edges is None:
            edges = np.zeros((n_edge, 2), dtype=I32)
        return cls(n_node, n_edge, np.asarray(edges, dtype=I32), a, ea, inc_n, inc_e, {})


def synthetic_drainage_network(n_node, n_edge, seed=0, window=64, max_degree=6, loop_hops=8):
    """Seeded drainage-like network: a random recursive tree whose
    node i > 0 drains into a parent drawn from the `window` preceding nodes (node 0 is the
    outfall), plus n_edge-(n_node-1) loop links between nodes at most `loop_hops` tree hops
    apart; no node exceeds `max_degree`.  Links point downstream (child -> parent) and are
    numbered by their upstream end, so node ids and link ids are both locality-preserving.
    Returns edges:(E,2) int32."""
    if n_edge < n_node - 1:
        raise ValueError('need at least n_node-1 links for a connected network')
    rng = np.random.default_rng(seed)
    parent = np.zeros(n_node, dtype=np.int64)
    deg = np.zeros(n_node, dtype=np.int64)
    draws = rng.random(n_node * 4)
    k = 0
    for i in range(1, n_node):
        lo = max(0, i - window)
        while True:
            if k >= draws.shape[0]:
                draws = rng.random(n_node)
                k = 0
            p = lo + int(draws[k] * (i - lo))
            k += 1
            if deg[p] < max_degree - 1 or (i - lo) <= 2:
                break
        parent[i] = p
        deg[p] += 1
        deg[i] += 1
    links = [(i, int(parent[i])) for i in range(1, n_node)]
    have = set((min(a, b), max(a, b)) for a, b in links)
    extra = n_edge - (n_node - 1)
    guard = 0
    while extra > 0:
        guard += 1
        if guard > 200 * n_edge + 1000:
            raise RuntimeError('could not place the requested loop links')
        a = int(rng.integers(1, n_node))
        b = a
        for _ in range(int(rng.integers(2, loop_hops + 1))):
            b = int(parent[b])
        key = (min(a, b), max(a, b))
        if a == b or key in have or deg[a] >= max_degree or deg[b] >= max_degree:
            continue
        have.add(key)
        links.append((a, b))
        deg[a] += 1
        deg[b] += 1
        extra -= 1
    links = np.asarray(links, dtype=np.int64)
    order = np.argsort(links.max(axis=1), kind='stable')
    return links[order].astype(I32)
