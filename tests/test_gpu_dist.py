"""GPU tests of the graph-sharded path (BASELINE.json config 4) on ONE card: every part of the node-cut plan runs its HIP
layers (`dist.hip_layers`) in turn on cuda:0 and the halo exchange is done by direct copies between the parts' buffers
(the pattern of tests/test_dist.py, with the HIP layers instead of the oracle as the local compute), plus the 200k-node
network itself through the unsharded HIP layer.  The N > 1 launch over RCCL is covered by the world_size-2 gloo test on
CPU (tests/test_dist.py) and by `UDS_DIST_BACKEND=gloo` rehearsals of bench.py; real multi-GPU runs are the driver's."""
import numpy as np
import pytest
import torch

import gnn_uds_amd as U
from gnn_uds_amd import dist as D
from oracle import sparse_csr as OS
from tests.util import cast, close

pytestmark = pytest.mark.gpu
TOL_BF16X3 = 1e-5      # as tests/test_gpu_parity.py: fused split-bf16 layer vs the fp64 oracle (measured <= 1.6e-6)


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need the MI355X'
    return torch.device('cuda', 0)


@pytest.fixture(scope='module')
def c2_problem(dev):
    """C2-size network, 3-layer block with sparse NodeEdge parameters, unsharded HIP result and the fp64 oracle."""
    g = U.DrainageGraph.from_edges(U.synthetic_drainage_network(2000, 2500, 0))
    d, L, S = 64, 3, 3
    block = U.SpatialBlock(g, d, L, 'relu', sparse_params=True, generator=torch.Generator().manual_seed(1)).to(dev)
    params = [ly.export_params() for ly in block.layers]
    gen = torch.Generator().manual_seed(2)
    x, e = torch.rand(S, 2000, d, generator=gen), torch.rand(S, 2500, d, generator=gen)
    with torch.no_grad():
        ox, oe = block(x.to(dev), e.to(dev))
    rx, re = x.double(), e.double()
    for p in params:
        rx, re = OS.spatial_layer_csr(rx, re, cast(p, torch.float64), (g.adj.rowptr, g.adj.col), (g.edge_adj.rowptr, g.edge_adj.col),
                                      (g.inc_n.rowptr, g.inc_n.col), (g.inc_e.rowptr, g.inc_e.col))
    return g, params, x, e, ox, oe, rx, re, d, L


@pytest.mark.parametrize('n_parts', [2, 4, 8])
def test_sharded_hip_block_matches_unsharded_and_oracle(dev, c2_problem, n_parts):
    """`build_partition_plan` + `hip_layers` per part + per-layer exchange: the own rows of every part equal the unsharded
    HIP block (same kernels; the order in which a row's incident values are added differs with the local numbering, so
    to rounding, not bit for bit) and the fp64 oracle of the whole network."""
    g, params, x, e, ox, oe, rx, re, d, L = c2_problem
    probs = D.build_partition_plan(g, n_parts)
    layers = [D.hip_layers(p, params, d, 'relu', 'bf16x3', dev) for p in probs]
    loc = [(x[:, p.nodes].to(dev).contiguous(), e[:, p.links].to(dev).contiguous()) for p in probs]
    with torch.no_grad():
        for i in range(L):
            loc = [layers[k][i](lx, le) for k, (lx, le) in enumerate(loc)]
            for k in range(n_parts):
                assert layers[k][i].network().plan_info()['fused'] & 1          # the fused kernel ran on every part
            if i + 1 < L:
                new = [(lx.clone(), le.clone()) for lx, le in loc]
                for p, (lx, le) in zip(probs, new):
                    for q in p.recv_nodes:
                        ti = lambda a: torch.as_tensor(a, dtype=torch.int64, device=dev)
                        lx[:, ti(p.recv_nodes[q])] = loc[q][0][:, ti(probs[q].send_nodes[p.rank])]
                        le[:, ti(p.recv_links[q])] = loc[q][1][:, ti(probs[q].send_links[p.rank])]
                loc = new
    for p, (lx, le) in zip(probs, loc):
        no, lo = len(p.own_nodes), len(p.own_links)
        close(lx[:, :no], rx[:, p.own_nodes], TOL_BF16X3)
        close(le[:, :lo], re[:, p.own_links], TOL_BF16X3)
        ux, ue = ox[:, torch.as_tensor(p.own_nodes, device=dev)], oe[:, torch.as_tensor(p.own_links, device=dev)]
        assert float((lx[:, :no] - ux).abs().max()) <= 4e-6 * max(1.0, float(ux.abs().max()))
        assert float((le[:, :lo] - ue).abs().max()) <= 4e-6 * max(1.0, float(ue.abs().max()))


def test_sharded_block_driver_single_rank_equals_plain_block(dev, c2_problem):
    """`ShardedSpatialBlock` with one part (no peers, no exchange) on the HIP layers is the plain block."""
    g, params, x, e, ox, oe, rx, re, d, L = c2_problem
    prob = D.build_partition_plan(g, 1)[0]
    layers = D.hip_layers(prob, params, d, 'relu', 'bf16x3', dev)
    block = D.ShardedSpatialBlock(prob, L, lambda p, i, xx, ee: layers[i](xx, ee), dev)
    lx, le = block.scatter_inputs(x.to(dev), e.to(dev))
    with torch.no_grad():
        sx, se = block.forward(lx, le)
    assert torch.equal(sx, ox) and torch.equal(se, oe)


def test_c4_size_network_properties(dev):
    """The 200k-node / 240k-link network of BASELINE.json config 4 through the HIP layer on one GPU (unsharded; the 8-way
    cut of the same network is checked on the host in tests/test_dist.py and at C2 size above): two runs bitwise identical,
    snapshots independent, and one FULL snapshot of both sides against the fp64 oracle."""
    g = U.DrainageGraph.from_edges(U.synthetic_drainage_network(200000, 240000, 0))
    d, S = 64, 3
    layer = U.SpatialLayer(g, d, 'relu', sparse_params=True, generator=torch.Generator().manual_seed(1)).to(dev)
    gen = torch.Generator().manual_seed(2)
    x, e = torch.rand(S, 200000, d, generator=gen).to(dev), torch.rand(S, 240000, d, generator=gen).to(dev)
    ox, oe = layer(x, e)
    assert layer.network().plan_info()['fused'] & 1
    ox2, oe2 = layer(x, e)
    assert torch.equal(ox, ox2) and torch.equal(oe, oe2)
    o1x, o1e = layer(x[2:3].contiguous(), e[2:3].contiguous())
    assert torch.equal(o1x[0], ox[2]) and torch.equal(o1e[0], oe[2])
    p = cast(layer.export_params(), torch.float64)
    rx, re = OS.spatial_layer_csr(x[2:3].double().cpu(), e[2:3].double().cpu(), p, (g.adj.rowptr, g.adj.col), (g.edge_adj.rowptr, g.edge_adj.col),
                                  (g.inc_n.rowptr, g.inc_n.col), (g.inc_e.rowptr, g.inc_e.col))
    close(ox[2:3], rx, TOL_BF16X3)
    close(oe[2:3], re, TOL_BF16X3)
    # the 8-way node cut of this network: a small cut and level parts (what the >= 6x scaling target rests on)
    part = D.partition_nodes(g, 8)
    cut = int((part[g.edges[:, 0]] != part[g.edges[:, 1]]).sum())
    assert cut <= 600, cut                  # id ranges: 463, round 2's depth-first ranges: 5 382 (tests/test_dist.py has the halo bounds)
    sizes = np.bincount(part, minlength=8)
    assert sizes.min() >= 24500 and sizes.max() <= 25500


def test_c3_size_batch_of_32_scenarios(dev):
    """BASELINE.json config 3 at its stated size: the 50 000-node / 65 000-link catchment with a batch of S = 32 rainfall
    scenarios through the fused layer: two runs bitwise identical, every snapshot independent of its batch (the batched
    result of snapshots 0, 13 and 31 equals the same snapshot run alone), and two FULL snapshots against the fp64 oracle."""
    g = U.DrainageGraph.from_edges(U.synthetic_drainage_network(50000, 65000, 0))
    d, S = 64, 32
    layer = U.SpatialLayer(g, d, 'relu', sparse_params=True, generator=torch.Generator().manual_seed(1)).to(dev)
    gen = torch.Generator().manual_seed(2)
    x, e = torch.rand(S, 50000, d, generator=gen).to(dev), torch.rand(S, 65000, d, generator=gen).to(dev)
    ox, oe = layer(x, e)
    assert layer.last_path == 'fused'
    ox2, oe2 = layer(x, e)
    assert torch.equal(ox, ox2) and torch.equal(oe, oe2)
    for s in (0, 13, 31):
        o1x, o1e = layer(x[s:s + 1].contiguous(), e[s:s + 1].contiguous())
        assert torch.equal(o1x[0], ox[s]) and torch.equal(o1e[0], oe[s])
    p = cast(layer.export_params(), torch.float64)
    sel = [5, 30]
    rx, re = OS.spatial_layer_csr(x[sel].double().cpu(), e[sel].double().cpu(), p, (g.adj.rowptr, g.adj.col), (g.edge_adj.rowptr, g.edge_adj.col),
                                  (g.inc_n.rowptr, g.inc_n.col), (g.inc_e.rowptr, g.inc_e.col))
    close(ox[sel], rx, TOL_BF16X3)
    close(oe[sel], re, TOL_BF16X3)


class _MailboxExchange(D.HaloExchange):
    """The exchange of `dist.HaloExchange` between rank THREADS of one process: the packed rows travel through queues with
    the event that marks them written, the receiver's current stream waits on it.  Same packing, same message order and the
    same stream discipline as the RCCL path (everything enqueued on the caller's current stream)."""

    def __init__(self, prob, device, mail):
        super().__init__(prob, device)
        self.mail = mail

    def __call__(self, x, e):
        st = torch.cuda.current_stream()
        for q in self.peers:
            sn, se = self.send_n[q], self.send_e[q]
            if len(sn) + len(se):
                out = self.pack(x, e, q)                      # uds_halo_pack
                assert torch.equal(out, torch.cat([x.index_select(1, sn), e.index_select(1, se)], dim=1))
                ev = torch.cuda.Event()
                ev.record(st)
                self.mail[(self.prob.rank, q)].put((out, ev))
        for q in self.peers:
            nn_ = len(self.recv_n[q])
            if nn_ + len(self.recv_e[q]):
                buf, ev = self.mail[(q, self.prob.rank)].get(timeout=120)
                st.wait_event(ev)
                self.unpack(buf, x, e, q)                     # uds_halo_unpack
                assert torch.equal(x[:, self.recv_n[q]], buf[:, :nn_]) and torch.equal(e[:, self.recv_e[q]], buf[:, nn_:])
                buf.record_stream(st)
        return x, e


@pytest.mark.parametrize('n_parts,stages', [(2, 2), (4, 2), (4, 3), (8, 1)])
def test_pipelined_exchange_on_side_streams(dev, c2_problem, n_parts, stages):
    """`ShardedSpatialBlock.forward` with the exchange pipelined over snapshot groups (side stream + events) on the HIP
    layers: one thread per rank on cuda:0, rows handed over in-process; own rows equal the fp64 oracle of the whole network
    and the in-order (stages=1) schedule bit for bit."""
    import queue
    import threading
    g, params, x, e, ox, oe, rx, re, d, L = c2_problem
    probs = D.build_partition_plan(g, n_parts)
    layers = [D.hip_layers(p, params, d, 'relu', 'bf16x3', dev) for p in probs]
    results = {}

    def run(stages_):
        mail = {(p, q): queue.Queue() for p in range(n_parts) for q in range(n_parts)}
        out, errs = [None] * n_parts, []

        def rank_main(k):
            try:
                torch.cuda.set_device(dev)
                blk = D.ShardedSpatialBlock(probs[k], L, lambda p, i, xx, ee: layers[p.rank][i](xx, ee), dev)
                blk.exchange = _MailboxExchange(probs[k], dev, mail)
                lx, le = x[:, probs[k].nodes].to(dev).contiguous(), e[:, probs[k].links].to(dev).contiguous()
                with torch.no_grad():
                    sx, se = blk.forward(lx, le, stages=stages_)
                torch.cuda.synchronize()
                out[k] = (sx.clone(), se.clone())
            except Exception as exc:       # surfaced in the main thread
                errs.append(exc)
        ts = [threading.Thread(target=rank_main, args=(k,)) for k in range(n_parts)]
        for t in ts:
            t.start()
        for t in ts:
            t.join(timeout=300)
        assert not errs, errs
        return out
    piped = run(stages)
    inorder = run(1)
    for p, (sx, se), (ix, ie) in zip(probs, piped, inorder):
        close(sx, rx[:, p.own_nodes], TOL_BF16X3)
        close(se, re[:, p.own_links], TOL_BF16X3)
        assert torch.equal(sx, ix) and torch.equal(se, ie)
