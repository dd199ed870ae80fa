"""Graph-sharded spatial block: node-cut partition of one big drainage network over the GPUs of a node, with the
boundary rows exchanged once per message-passing layer (one process per GPU, `torch.distributed` over RCCL / xGMI).

The reference has nothing like this (whole graph = one dense matrix on one device); it exists for the
200k-node case of BASELINE.json (`configs[3]`).  Snapshots of the headline-size network do NOT use it -- they shard
by snapshot with no collective (bench.py).

Dependency radius of one spatial layer (`emulator.py:225-230`): the outputs of a rank's own nodes / links need
  hx of   A1 = adj-neighbours of own nodes        -> x of A1 and e of L1 = links incident to A1
  he of   B1 = edge_adj-neighbours of own links   -> e of B1 and x of M1 = end nodes of B1
so every rank computes the layer on the sub-network induced by (A1 u M1, L1 u B1): rows it owns come out exact, the
halo rows are recomputed redundantly where needed (halo hx / he) and otherwise ignored; after the layer every rank
sends the exact outputs of the own rows its peers hold as halo.  Near-tree networks cut into P connected parts have
O(P) cut links, so a message is a few dozen rows: latency-bound, hence ONE message per peer per layer (node rows and
link rows packed together), posted as grouped isend/irecv (ncclSend/ncclRecv on RCCL, each peer pair on its own
xGMI link).

All index bookkeeping here is host-side numpy and deterministic: every rank derives the same plan from the same
network, nothing is negotiated at run time.  Tested with world_size-2 gloo processes on CPU (tests/test_dist.py).
"""
from dataclasses import dataclass, field
from typing import Dict, List

import numpy as np
import torch
import torch.distributed as dist

from .graph import CSR, DrainageGraph

I32 = np.int32


def _link_matrix(graph):
    """Symmetric 0/1 node x node matrix of the LINKS themselves (not the `order`-hop GAT pattern): what a node cut cuts."""
    import scipy.sparse as sp
    n = graph.n_node
    e = np.asarray(graph.edges, dtype=np.int64)
    e = e[e[:, 0] != e[:, 1]]
    a = sp.coo_matrix((np.ones(2 * len(e), dtype=np.int8), (np.concatenate([e[:, 0], e[:, 1]]), np.concatenate([e[:, 1], e[:, 0]]))),
                      shape=(n, n)).tocsr()
    a.data[:] = 1                                                      # parallel links count once
    return a


def _bfs_level_order(a, graph):
    """Breadth-first order of every connected component, started at its outfall (a node no link leaves; the lowest-numbered
    node where a component has none): a range of it is a band of tree levels, whose boundary is one level wide."""
    from scipy.sparse.csgraph import breadth_first_order, connected_components
    n = a.shape[0]
    _, label = connected_components(a, directed=False)
    has_out = np.zeros(n, dtype=bool)
    has_out[np.asarray(graph.edges, dtype=np.int64)[:, 0]] = True
    first = np.full(label.max() + 1, n, dtype=np.int64)
    np.minimum.at(first, label, np.arange(n))
    sink = np.full(label.max() + 1, n, dtype=np.int64)
    idx = np.nonzero(~has_out)[0]
    np.minimum.at(sink, label[idx], idx)
    roots = np.where(sink < n, sink, first)
    return np.concatenate([breadth_first_order(a, int(r), directed=False, return_predecessors=False) for r in roots])


def _cut_links(graph, part):
    e = np.asarray(graph.edges, dtype=np.int64)
    return int((part[e[:, 0]] != part[e[:, 1]]).sum())


def _refine_boundary(a, part, n_parts, slack, passes=4):
    """Greedy boundary refinement (one Kernighan-Lin style sweep per pass): a boundary node moves to the neighbouring part
    that holds MORE of its links than its own part does, as long as no part leaves [n/P - slack, n/P + slack] nodes.  Nodes are
    visited in ascending id, gains are re-evaluated at visit time: deterministic, every rank derives the same result."""
    indptr, indices = a.indptr, a.indices
    n = a.shape[0]
    size = np.bincount(part, minlength=n_parts).astype(np.int64)
    lo, hi = n // n_parts - slack, -(-n // n_parts) + slack
    for _ in range(passes):
        src = np.repeat(np.arange(n), np.diff(indptr))
        boundary = np.unique(src[part[src] != part[indices]])
        moved = 0
        for v in boundary:
            nb = part[indices[indptr[v]:indptr[v + 1]]]
            own = part[v]
            cnt = np.bincount(nb, minlength=n_parts)
            best = int(np.argmax(cnt))                               # lowest part id among ties
            if best != own and cnt[best] > cnt[own] and size[own] - 1 >= lo and size[best] + 1 <= hi:
                part[v] = best
                size[own] -= 1
                size[best] += 1
                moved += 1
        if not moved:
            break
    return part


def partition_nodes(graph, n_parts, return_info=False):
    """Node -> part (int32), a node cut into `n_parts` parts of n/P nodes (+- 2 %).

    Three candidate orders are cut into equal contiguous ranges -- the node ids as given (SWMM files and the synthetic
    generator number a network roughly upstream -> downstream, so an id range is already a band of the tree), breadth-first
    levels from the outfalls, and reverse Cuthill-McKee -- and the one that cuts the fewest links is kept, then improved by a
    greedy boundary refinement.  (Round 2 used ranges of a DEPTH-first pre-order: on a deep narrow tree such a range cuts
    about one link per level -- 5 382 cut links 8-way on the 200k-node network against 463 for the plain id ranges.)
    All of it is deterministic host-side integer work; every rank derives the same partition from the same network."""
    from scipy.sparse.csgraph import reverse_cuthill_mckee
    n = graph.n_node
    a = _link_matrix(graph)
    ranges = (np.arange(n, dtype=np.int64) * n_parts // n).astype(I32)
    orders = {'id': np.arange(n, dtype=np.int64), 'bfs': _bfs_level_order(a, graph),
              'rcm': np.asarray(reverse_cuthill_mckee(a, symmetric_mode=True), dtype=np.int64)}
    best, info = None, {}
    for name, order in orders.items():
        part = np.empty(n, dtype=I32)
        part[order] = ranges
        info[name] = _cut_links(graph, part)
        if best is None or info[name] < info[best[0]]:
            best = (name, part)
    part = _refine_boundary(a, best[1].copy(), n_parts, slack=max(1, n // n_parts // 50))
    info['refined'] = _cut_links(graph, part)
    info['kept'] = best[0]
    if info['refined'] > info[best[0]]:                              # never worse than the best plain range split
        part, info['refined'] = best[1], info[best[0]]
    return (part, info) if return_info else part


def _rows_union(csr, rows):
    if len(rows) == 0:
        return np.zeros(0, dtype=np.int64)
    rp = csr.rowptr.astype(np.int64)
    cnt = rp[rows + 1] - rp[rows]
    idx = np.repeat(rp[rows], cnt) + (np.arange(cnt.sum()) - np.repeat(np.cumsum(cnt) - cnt, cnt))
    return np.unique(csr.col[idx].astype(np.int64))


def _induced(csr, rows, col_map, n_cols_local):
    """Sub-CSR on `rows` with columns restricted to the local set (col_map: global -> local or -1); also returns the
    global position of every kept entry (to gather per-entry parameters such as the NodeEdge support values)."""
    rp = csr.rowptr.astype(np.int64)
    cnt = rp[rows + 1] - rp[rows]
    pos = np.repeat(rp[rows], cnt) + (np.arange(cnt.sum()) - np.repeat(np.cumsum(cnt) - cnt, cnt))
    lcol = col_map[csr.col[pos].astype(np.int64)]
    keep = lcol >= 0
    row_of = np.repeat(np.arange(len(rows)), cnt)[keep]
    lcol, pos = lcol[keep], pos[keep]
    o = np.lexsort((lcol, row_of))
    row_of, lcol, pos = row_of[o], lcol[o], pos[o]
    rowptr = np.zeros(len(rows) + 1, dtype=np.int64)
    np.add.at(rowptr, row_of + 1, 1)
    val = None if csr.val is None else csr.val[pos]
    return CSR(np.cumsum(rowptr).astype(I32), lcol.astype(I32), len(rows), n_cols_local, val), pos


@dataclass
class LocalProblem:
    """What one rank computes and exchanges.  Local row order: own rows first (ascending id), then halo rows."""
    rank: int
    own_nodes: np.ndarray
    own_links: np.ndarray
    nodes: np.ndarray                 # local node list (global ids)
    links: np.ndarray
    graph: DrainageGraph              # induced sub-network on (nodes, links)
    inc_n_pos: np.ndarray             # global inc_n / inc_e entry of every local entry
    inc_e_pos: np.ndarray
    send_nodes: Dict[int, np.ndarray] = field(default_factory=dict)   # peer -> LOCAL indices of own rows it needs
    send_links: Dict[int, np.ndarray] = field(default_factory=dict)
    recv_nodes: Dict[int, np.ndarray] = field(default_factory=dict)   # peer -> LOCAL indices of halo rows it owns
    recv_links: Dict[int, np.ndarray] = field(default_factory=dict)


def build_partition_plan(graph, n_parts, part=None):
    """Deterministic plan for all ranks: List[LocalProblem]."""
    if part is None:
        part = partition_nodes(graph, n_parts)
    part = np.asarray(part, dtype=np.int64)
    link_part = part[graph.edges[:, 0].astype(np.int64)]            # a link belongs to the part of its from-node
    probs: List[LocalProblem] = []
    for r in range(n_parts):
        own_n = np.nonzero(part == r)[0]
        own_e = np.nonzero(link_part == r)[0]
        a1 = _rows_union(graph.adj, own_n)
        l1 = _rows_union(graph.inc_n, a1)
        b1 = _rows_union(graph.edge_adj, own_e)
        m1 = _rows_union(graph.inc_e, b1)
        nodes = np.concatenate([own_n, np.setdiff1d(np.union1d(a1, m1), own_n)])
        links = np.concatenate([own_e, np.setdiff1d(np.union1d(l1, b1), own_e)])
        nmap = np.full(graph.n_node, -1, dtype=np.int64)
        nmap[nodes] = np.arange(len(nodes))
        lmap = np.full(graph.n_edge, -1, dtype=np.int64)
        lmap[links] = np.arange(len(links))
        adj, _ = _induced(graph.adj, nodes, nmap, len(nodes))
        eadj, _ = _induced(graph.edge_adj, links, lmap, len(links))
        inc_n, pos_n = _induced(graph.inc_n, nodes, lmap, len(links))
        inc_e, pos_e = _induced(graph.inc_e, links, nmap, len(nodes))
        edges = np.stack([nmap[graph.edges[links, 0].astype(np.int64)], nmap[graph.edges[links, 1].astype(np.int64)]], axis=1)
        sub = DrainageGraph(len(nodes), len(links), edges.astype(I32), adj, eadj, inc_n, inc_e, dict(part=r, n_parts=n_parts))
        probs.append(LocalProblem(r, own_n, own_e, nodes, links, sub, pos_n, pos_e))
    for p in probs:                                                   # who needs what from whom
        halo_n, halo_e = p.nodes[len(p.own_nodes):], p.links[len(p.own_links):]
        for q in range(n_parts):
            if q == p.rank:
                continue
            hn = np.nonzero(part[halo_n] == q)[0]
            he = np.nonzero(link_part[halo_e] == q)[0]
            if len(hn) == 0 and len(he) == 0:
                continue
            p.recv_nodes[q] = (len(p.own_nodes) + hn).astype(np.int64)
            p.recv_links[q] = (len(p.own_links) + he).astype(np.int64)
            peer = probs[q]
            p_n = np.searchsorted(peer.own_nodes, halo_n[hn])        # own rows are sorted and come first locally
            p_e = np.searchsorted(peer.own_links, halo_e[he])
            peer.send_nodes[p.rank] = p_n.astype(np.int64)
            peer.send_links[p.rank] = p_e.astype(np.int64)
    return probs


class HaloExchange:
    """One message per peer: [node rows | link rows] of width F, posted as grouped isend / irecv."""

    def __init__(self, prob, device, group=None):
        self.prob, self.group = prob, group
        t = lambda a: torch.as_tensor(a, dtype=torch.int64, device=device)
        self.peers = sorted(set(prob.send_nodes) | set(prob.recv_nodes))
        self.send_n = {q: t(prob.send_nodes.get(q, np.zeros(0, np.int64))) for q in self.peers}
        self.send_e = {q: t(prob.send_links.get(q, np.zeros(0, np.int64))) for q in self.peers}
        self.recv_n = {q: t(prob.recv_nodes.get(q, np.zeros(0, np.int64))) for q in self.peers}
        self.recv_e = {q: t(prob.recv_links.get(q, np.zeros(0, np.int64))) for q in self.peers}
        # on the GPU a message is packed / unpacked by ONE launch (uds_halo_pack / uds_halo_unpack) with int32 row lists
        self.on_gpu = torch.device(device).type == 'cuda'
        if self.on_gpu:
            i32 = lambda d: {q: v.to(torch.int32) for q, v in d.items()}
            self._i32 = (i32(self.send_n), i32(self.send_e), i32(self.recv_n), i32(self.recv_e))

    def bytes_per_layer(self, S, F):
        return sum((len(self.send_n[q]) + len(self.send_e[q])) * S * F * 4 for q in self.peers)

    def __call__(self, x, e):
        """x (S, n_local_nodes, F), e (S, n_local_links, F): overwrite the halo rows with the owners' exact rows.  Everything
        is enqueued on the CURRENT stream (RCCL: the send / receive kernels are ordered after the stream's earlier work and
        `wait()` blocks the stream, not the host), so a caller that runs this under a side stream overlaps it with compute."""
        if not self.peers:
            return x, e
        ops, recvs, keep = [], [], []
        for q in self.peers:
            nn_, ne_ = len(self.recv_n[q]), len(self.recv_e[q])
            if nn_ + ne_:
                buf = torch.empty((x.shape[0], nn_ + ne_, x.shape[-1]), device=x.device, dtype=x.dtype)
                recvs.append((q, buf, nn_))
                ops.append(dist.P2POp(dist.irecv, buf, q, self.group))
            sn, se = self.send_n[q], self.send_e[q]
            if len(sn) + len(se):
                out = self.pack(x, e, q)
                keep.append(out)
                ops.append(dist.P2POp(dist.isend, out, q, self.group))
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        for q, buf, nn_ in recvs:
            self.unpack(buf, x, e, q)
        return x, e

    def pack(self, x, e, q):
        """The message for peer q: [own node rows it holds as halo | own link rows], (S, n, F)."""
        if self.on_gpu and x.is_cuda and x.shape[-1] % 4 == 0 and x.is_contiguous() and e.is_contiguous():
            from . import _lib
            return _lib.halo_pack(x, e, self._i32[0][q], self._i32[1][q])
        return torch.cat([x.index_select(1, self.send_n[q]), e.index_select(1, self.send_e[q])], dim=1).contiguous()

    def unpack(self, buf, x, e, q):
        """Scatter peer q's message into the halo rows it owns (in place)."""
        if self.on_gpu and x.is_cuda and x.shape[-1] % 4 == 0 and x.is_contiguous() and e.is_contiguous():
            from . import _lib
            _lib.halo_unpack(buf, x, e, self._i32[2][q], self._i32[3][q])
            return
        nn_ = len(self.recv_n[q])
        x.index_copy_(1, self.recv_n[q], buf[:, :nn_])
        e.index_copy_(1, self.recv_e[q], buf[:, nn_:])


class ShardedSpatialBlock:
    """The L-layer spatial block of one rank of a graph-sharded run.

    layer_fn(prob, layer_index, x_local, e_local) -> (x', e') computes one spatial layer on the rank's sub-network
    (exact on own rows).  The product passes HIP `SpatialLayer`s built on `prob.graph` (see `hip_layers`); the CPU
    tests pass the oracle."""

    def __init__(self, prob, n_layers, layer_fn, device, group=None):
        self.prob, self.n_layers, self.layer_fn = prob, n_layers, layer_fn
        self.exchange = HaloExchange(prob, device, group)
        self._side = None                     # side stream of the pipelined exchange, created on first use

    def scatter_inputs(self, x_global, e_global):
        """Local buffers from replicated global inputs (own + halo rows are simply read)."""
        dev = x_global.device
        ni = torch.as_tensor(self.prob.nodes, dtype=torch.int64, device=dev)
        li = torch.as_tensor(self.prob.links, dtype=torch.int64, device=dev)
        return x_global.index_select(1, ni).contiguous(), e_global.index_select(1, li).contiguous()

    def forward(self, x_local, e_local, stages=2):
        """x_local (S, n_local_nodes, F), e_local likewise, halo rows valid.  Returns the own rows of the block output.

        The snapshots are independent, so the exchange is hidden by PIPELINING OVER SNAPSHOT GROUPS: the S snapshots are cut
        into `stages` groups; as soon as a group's layer is computed its boundary rows are packed, sent and the received
        halo rows scattered on a side stream, while the main stream computes the same layer of the next group (and then the
        next layer of the first group, whose halo has arrived by then).  Only a group whose exchange is slower than one
        group-layer of compute leaves the main stream waiting.  The fused kernel is launched per group (it takes any S); no
        kernel needs a boundary / interior split of its tiles.  stages=1 (or one snapshot, or no peers): compute, then
        exchange, in order on one stream."""
        S = x_local.shape[0]
        L, ex = self.n_layers, self.exchange
        G = max(1, min(int(stages), S)) if ex.peers else 1
        if G == 1:
            for i in range(L):
                x_local, e_local = self.layer_fn(self.prob, i, x_local, e_local)
                if i + 1 < L:
                    x_local, e_local = ex(x_local, e_local)
            return x_local[:, :len(self.prob.own_nodes)], e_local[:, :len(self.prob.own_links)]
        cuts = [g * S // G for g in range(G + 1)]
        xs = [x_local[cuts[g]:cuts[g + 1]] for g in range(G)]
        es = [e_local[cuts[g]:cuts[g + 1]] for g in range(G)]
        on_gpu = x_local.is_cuda
        if on_gpu:
            if self._side is None:
                self._side = torch.cuda.Stream(device=x_local.device)
            main, side = torch.cuda.current_stream(x_local.device), self._side
        ready = [None] * G                    # event: the halo rows of group g hold the previous layer's exchanged values
        for i in range(L):
            for g in range(G):
                if ready[g] is not None:
                    main.wait_event(ready[g])
                xs[g], es[g] = self.layer_fn(self.prob, i, xs[g], es[g])
                if i + 1 == L:
                    continue
                if not on_gpu:                # CPU tensors (gloo tests): same order of messages, no streams
                    ex(xs[g], es[g])
                    continue
                computed = torch.cuda.Event()
                computed.record(main)
                with torch.cuda.stream(side):
                    side.wait_event(computed)
                    ex(xs[g], es[g])          # pack, send / receive, scatter: all ordered on the side stream
                    ready[g] = torch.cuda.Event()
                    ready[g].record(side)
                xs[g].record_stream(side)
                es[g].record_stream(side)
        no, lo = len(self.prob.own_nodes), len(self.prob.own_links)
        return torch.cat([t[:, :no] for t in xs], dim=0), torch.cat([t[:, :lo] for t in es], dim=0)


def hip_layers(prob, global_params, embed_size, activation='relu', precision='bf16x3', device='cuda'):
    """HIP `SpatialLayer`s for a rank's sub-network from GLOBAL per-layer parameters (dicts with the keys of
    `SpatialLayer.export_params()` in sparse form: 'ne_n_v' / 'ne_e_v' are indexed by global support entry)."""
    from .layers import SpatialLayer
    layers = []
    for p in global_params:
        fx, fe = p['ex_k'].shape[0], p['xe_k'].shape[0]
        ly = SpatialLayer(prob.graph, embed_size, activation, fx=fx, fe=fe, sparse_params=True, precision=precision).to(device)
        f = lambda t: None if t is None else t.to(torch.float32).to(device).contiguous()
        ly.dense_xe.kernel.data, ly.dense_xe.bias.data = f(p['xe_k']), f(p['xe_b'])
        ly.dense_ex.kernel.data, ly.dense_ex.bias.data = f(p['ex_k']), f(p['ex_b'])
        ly.node_edge_n.weight.data = f(p['ne_n_v'][torch.as_tensor(prob.inc_n_pos)])
        ly.node_edge_e.weight.data = f(p['ne_e_v'][torch.as_tensor(prob.inc_e_pos)])
        ly.node_edge_n.bias.data = torch.zeros_like(ly.node_edge_n.weight.data)
        ly.node_edge_e.bias.data = torch.zeros_like(ly.node_edge_e.weight.data)
        for m, k in ((ly.gat_x, 'gx'), (ly.gat_e, 'ge')):
            m.kernel.data, m.bias.data = f(p[k + '_k']), f(p[k + '_b'])
            m.attn_kernel_self.data, m.attn_kernel_neighs.data = f(p[k + '_as']), f(p[k + '_an'])
        layers.append(ly)
    return layers


def all_ranks_finite(value, group=None):
    """True only when `value` (a tensor) is finite on EVERY rank: a MIN all-reduce of the local 0/1 flag, so that all ranks
    take the same branch -- a rank that raised on its own non-finite loss would leave its peers waiting in the gradient
    all-reduce.  The local test alone when torch.distributed is not initialised or the world has one rank."""
    ok = torch.isfinite(value).all()
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        flag = ok.to(torch.float32).reshape(1)
        if dist.get_backend(group) == 'gloo':
            flag = flag.cpu()
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        return bool(flag.item() > 0.5)
    return bool(ok)


def allreduce_gradients(params, group=None, bucket_bytes=64 << 20):
    """Data-parallel training (snapshot / scenario sharding adds ONE gradient all-reduce per step): average
    the `.grad` of `params` over the ranks.  Gradients are packed into flat fp32 buckets (one bucket for a whole emulator:
    a few MB) so the ring all-reduce over xGMI runs once on a large message instead of once per tensor; parameters a rank
    did not touch count as zero.  No-op when torch.distributed is not initialised or the world has one rank."""
    if not (dist.is_available() and dist.is_initialized()):
        return 0
    world = dist.get_world_size(group)
    if world == 1:
        return 0
    params = [p for p in params if p.requires_grad]
    n_calls, i = 0, 0
    while i < len(params):
        j, size = i, 0
        while j < len(params) and (j == i or (size + params[j].numel()) * 4 <= bucket_bytes):
            size += params[j].numel()
            j += 1
        flat = torch.zeros(size, dtype=torch.float32, device=params[i].device)
        off = 0
        for p in params[i:j]:
            if p.grad is not None:
                flat[off:off + p.numel()] = p.grad.reshape(-1)
            off += p.numel()
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat /= world
        off = 0
        for p in params[i:j]:
            p.grad = flat[off:off + p.numel()].reshape(p.shape).clone()
            off += p.numel()
        n_calls += 1
        i = j
    return n_calls
