"""Graph-sharded spatial block (gnn_uds_amd/dist.py): partition plans and the per-layer halo exchange.

CPU only.  The local layer compute is the fp64 sparse oracle (the product plugs HIP layers into the same driver), so
what is tested is exactly the distributed bookkeeping: own / halo sets, induced sub-networks, parameter gathering by
global support position, and the one-message-per-peer exchange (world_size 2 over gloo; 4 and 8 parts simulated in
one process)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import gnn_uds_amd as U
from gnn_uds_amd import dist as D
from oracle import sparse_csr as OS
from tests.util import spatial_params

N, E, DM, L, S = 400, 480, 8, 3, 2


def _problem():
    g = U.DrainageGraph.from_edges(U.synthetic_drainage_network(N, E, seed=3))
    params = [spatial_params(N, E, DM, DM, DM, seed=10 + i, dense_ne=False, nnz_n=g.inc_n.nnz, nnz_e=g.inc_e.nnz) for i in range(L)]
    gen = torch.Generator().manual_seed(7)
    x = torch.rand(S, N, DM, generator=gen, dtype=torch.float64)
    e = torch.rand(S, E, DM, generator=gen, dtype=torch.float64)
    return g, params, x, e


def _csr(c):
    return (c.rowptr, c.col)


def _full_oracle(g, params, x, e):
    for p in params:
        x, e = OS.spatial_layer_csr(x, e, p, _csr(g.adj), _csr(g.edge_adj), _csr(g.inc_n), _csr(g.inc_e))
    return x, e


def _oracle_layer_fn(params):
    def fn(prob, i, x, e):
        p = dict(params[i])
        p['ne_n_v'] = params[i]['ne_n_v'][torch.as_tensor(prob.inc_n_pos)]       # gathered by global support position
        p['ne_e_v'] = params[i]['ne_e_v'][torch.as_tensor(prob.inc_e_pos)]
        sg = prob.graph
        return OS.spatial_layer_csr(x, e, p, _csr(sg.adj), _csr(sg.edge_adj), _csr(sg.inc_n), _csr(sg.inc_e))
    return fn


@pytest.mark.parametrize('n_parts', [2, 4, 8])
def test_partition_plan_bookkeeping(n_parts):
    g, _, _, _ = _problem()
    part = D.partition_nodes(g, n_parts)
    slack = max(1, N // n_parts // 50)
    sizes = np.bincount(part, minlength=n_parts)
    assert part.dtype == np.int32 and sizes.min() >= N // n_parts - slack and sizes.max() <= -(-N // n_parts) + slack     # balanced to 2 %
    probs = D.build_partition_plan(g, n_parts)
    own_n = np.concatenate([p.own_nodes for p in probs])
    own_e = np.concatenate([p.own_links for p in probs])
    assert sorted(own_n.tolist()) == list(range(N)) and sorted(own_e.tolist()) == list(range(E))       # a partition
    cut = int((part[g.edges[:, 0]] != part[g.edges[:, 1]]).sum())
    id_range = (np.arange(N, dtype=np.int64) * n_parts // N).astype(np.int32)
    assert cut <= int((id_range[g.edges[:, 0]] != id_range[g.edges[:, 1]]).sum())      # never worse than the plain id ranges
    for p in probs:
        assert (np.diff(p.own_nodes) > 0).all() and np.array_equal(p.nodes[:len(p.own_nodes)], p.own_nodes)
        # every adj-neighbour of an own node and every line-graph neighbour of an own link is local, rows complete
        for i, n in enumerate(p.own_nodes):
            want = g.adj.col[g.adj.rowptr[n]:g.adj.rowptr[n + 1]]
            got = p.nodes[p.graph.adj.col[p.graph.adj.rowptr[i]:p.graph.adj.rowptr[i + 1]]]
            assert np.array_equal(np.sort(got), np.sort(want))
        for i, l in enumerate(p.own_links):
            want = g.edge_adj.col[g.edge_adj.rowptr[l]:g.edge_adj.rowptr[l + 1]]
            got = p.links[p.graph.edge_adj.col[p.graph.edge_adj.rowptr[i]:p.graph.edge_adj.rowptr[i + 1]]]
            assert np.array_equal(np.sort(got), np.sort(want))
        # send / recv lists mirror each other and address the same global rows
        for q, idx in p.send_nodes.items():
            assert np.array_equal(p.nodes[idx], probs[q].nodes[probs[q].recv_nodes[p.rank]])
        for q, idx in p.send_links.items():
            assert np.array_equal(p.links[idx], probs[q].links[probs[q].recv_links[p.rank]])
        assert (p.graph.inc_n.col[:0] == 0).all() and np.array_equal(g.inc_n.col[p.inc_n_pos], p.links[p.graph.inc_n.col])


def _halo_stats(g, part, n_parts):
    probs = D.build_partition_plan(g, n_parts, part)
    return dict(cut=int((part[g.edges[:, 0]] != part[g.edges[:, 1]]).sum()),
                halo=max(len(p.nodes) - len(p.own_nodes) + len(p.links) - len(p.own_links) for p in probs),
                sent=max(sum(len(v) for v in p.send_nodes.values()) + sum(len(v) for v in p.send_links.values()) for p in probs),
                peers=max(len(set(p.send_nodes) | set(p.recv_nodes)) for p in probs))


def test_c4_partition_cut_and_halo():
    """The 8-way node cut of the 200k-node / 240k-link network (BASELINE.json config 4), as numbered by the generator AND on a
    label-shuffled copy (no help from the numbering): <= 600 cut links, <= 500 halo rows and <= 500 rows sent per rank and
    layer, <= 3 peers -- the plain id-range split of the unshuffled network gives 463 / 415 / 2, round 2's depth-first ranges
    gave 5 382 cut links and 5 peers.  Parts stay within 2 % of n / 8."""
    import gnn_uds_amd as U
    edges = U.synthetic_drainage_network(200000, 240000, 0)
    perm = np.random.default_rng(1).permutation(200000).astype(edges.dtype)
    for name, ed in (('generator numbering', edges), ('shuffled labels', perm[edges])):
        g = U.DrainageGraph.from_edges(ed)
        part, info = D.partition_nodes(g, 8, return_info=True)
        sizes = np.bincount(part, minlength=8)
        assert sizes.min() >= 25000 - 500 and sizes.max() <= 25000 + 500, (name, sizes)
        st = _halo_stats(g, part, 8)
        assert st['cut'] == info['refined'] and st['cut'] <= min(info['id'], info['bfs'], info['rcm']), (name, st, info)
        assert st['cut'] <= 600 and st['halo'] <= 500 and st['sent'] <= 500 and st['peers'] <= 3, (name, st, info)
        part2, info2 = D.partition_nodes(g, 2, return_info=True)
        assert info2['refined'] <= 80, (name, info2)


@pytest.mark.parametrize('n_parts', [4, 8])
def test_sharded_block_simulated_ranks(n_parts):
    """All ranks in one process, the exchange done by direct copies: the plan alone reproduces the full-graph result."""
    g, params, x, e = _problem()
    rx, re = _full_oracle(g, params, x, e)
    probs = D.build_partition_plan(g, n_parts)
    fn = _oracle_layer_fn(params)
    loc = [(x[:, p.nodes].clone(), e[:, p.links].clone()) for p in probs]
    for i in range(L):
        loc = [fn(p, i, lx, le) for p, (lx, le) in zip(probs, loc)]
        new = [(lx.clone(), le.clone()) for lx, le in loc]
        for p, (lx, le) in zip(probs, new):
            for q in p.recv_nodes:
                lx[:, p.recv_nodes[q]] = loc[q][0][:, probs[q].send_nodes[p.rank]]
                le[:, p.recv_links[q]] = loc[q][1][:, probs[q].send_links[p.rank]]
        loc = new
    for p, (lx, le) in zip(probs, loc):
        assert float((lx[:, :len(p.own_nodes)] - rx[:, p.own_nodes]).abs().max()) < 1e-11
        assert float((le[:, :len(p.own_links)] - re[:, p.own_links]).abs().max()) < 1e-11


def _worker(rank, world, port, result):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        g, params, x, e = _problem()
        prob = D.build_partition_plan(g, world)[rank]
        block = D.ShardedSpatialBlock(prob, L, _oracle_layer_fn(params), torch.device('cpu'))
        lx, le = block.scatter_inputs(x, e)
        ox, oe = block.forward(lx, le)                    # pipelined over two snapshot groups (the default)
        o1x, o1e = block.forward(lx, le, stages=1)        # compute, then exchange, layer by layer
        assert torch.equal(ox, o1x) and torch.equal(oe, o1e)
        o3x, _ = block.forward(lx, le, stages=S + 5)      # more groups than snapshots: one snapshot per group
        assert torch.equal(ox, o3x)
        rx, re = _full_oracle(g, params, x, e)
        err = max(float((ox - rx[:, prob.own_nodes]).abs().max()), float((oe - re[:, prob.own_links]).abs().max()))
        t = torch.tensor([err], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if rank == 0:
            result.put((float(t.item()), block.exchange.bytes_per_layer(S, DM), len(block.exchange.peers)))
    finally:
        dist.destroy_process_group()


def test_sharded_block_two_processes_gloo():
    ctx = mp.get_context('spawn')
    result = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, result)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    err, nbytes, peers = result.get(timeout=10)
    assert err < 1e-11 and peers == 1 and 0 < nbytes < S * (N + E) * DM * 4 * 0.2     # a small halo, one peer


def _grad_worker(rank, world, port, result):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(7)
        params = [torch.nn.Parameter(torch.zeros(s)) for s in ((5, 3), (7,), (2, 2, 2))]
        grads = [[torch.rand(p.shape, generator=g) for p in params] for _ in range(world)]      # every rank draws all of them
        for p, gr in zip(params, grads[rank]):
            p.grad = gr.clone()
        params[1].grad = None if rank == 1 else params[1].grad                                   # an untouched parameter counts as 0
        calls_one = D.allreduce_gradients(params)
        want = [sum(grads[r][k] if not (k == 1 and r == 1) else torch.zeros_like(grads[r][k]) for r in range(world)) / world
                for k in range(len(params))]
        err = max(float((p.grad - w).abs().max()) for p, w in zip(params, want))
        for p, gr in zip(params, grads[rank]):
            p.grad = gr.clone()
        calls_small = D.allreduce_gradients(params, bucket_bytes=40)                              # forces several buckets
        err = max(err, max(float((p.grad - sum(grads[r][k] for r in range(world)) / world).abs().max())
                           for k, p in enumerate(params)))
        # a non-finite loss on ONE rank is seen by all of them (so that all raise together, none hangs in the all-reduce)
        flags = (D.all_ranks_finite(torch.tensor(1.0)), D.all_ranks_finite(torch.tensor(float('nan') if rank == 1 else 1.0)))
        t = torch.tensor([float(flags[0]), float(flags[1])])
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        assert flags == (True, False) and t.tolist() == [float(world), 0.0]
        if rank == 0:
            result.put((err, calls_one, calls_small))
    finally:
        dist.destroy_process_group()


def test_gradient_allreduce_two_processes_gloo():
    ctx = mp.get_context('spawn')
    result = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, result)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    err, calls_one, calls_small = result.get(timeout=10)
    assert err < 1e-7 and calls_one == 1 and calls_small == 3


def test_bench_gpus_flag_starts_the_ranks_itself():
    """`python bench.py --gpus N` (how the driver calls it, WORLD_SIZE unset) must start N ranks as a child launcher BEFORE
    anything touches the GPU, and a rank whose WORLD_SIZE disagrees with --gpus must refuse to run.  On this CPU-only host
    every rank stops at 'needs an MI355X' -- after the launcher has started both of them."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert r.stderr.count('bench.py needs an MI355X') >= 2, r.stderr[-2000:]          # both ranks got as far as the device check
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2'], env=dict(env, WORLD_SIZE='3', RANK='0'),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and '--gpus 2 but WORLD_SIZE=3' in r.stderr
