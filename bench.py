"""Headline benchmark: graph-steps/s of the spatial block on the synthetic drainage network
|V|=10k, |E|=12k, d=64 (BASELINE.json metric), one process per GPU.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" = one pass of the hot path over one batch: the L-layer spatial block
(`emulator.py:219-235`) applied to S = B*T snapshots resident in HBM = L*S graph-steps
(graph-step = one spatial layer on one snapshot).  With N > 1 every rank
runs its own S snapshots of the same network (snapshots are independent inside a forward,
`emulator.py:217-218`): weak scaling, no data-path collective.  Rank 0 prints ONE JSON line.

`--workload c4` switches to BASELINE.json's 200k-node mega-catchment: ONE network node-cut partitioned
over the N ranks, boundary rows exchanged once per layer over RCCL (gnn_uds_amd/dist.py): strong scaling.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable


def algorithmic_bytes_per_graph_step(g, f_in, d):
    """compulsory read of x,e + write of x',e' + one pass over the index and
    NodeEdge-weight arrays."""
    N, E = g.n_node, g.n_edge
    return 4 * (N + E) * (f_in + d) + 4 * ((N + 1) + g.adj.nnz + (E + 1) + g.edge_adj.nnz + 4 * 2 * E)


def recorded_traffic(args):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/traffic.json),
    only when this run's configuration is the one they were collected on; None otherwise."""
    try:
        with open(os.path.join(ROOT, 'profiles', 'traffic.json')) as fh:
            rec = json.load(fh)
    except (OSError, ValueError):
        return None
    cfg = rec.get('config', {})
    same = (cfg.get('nodes') == args.nodes and cfg.get('links') == args.links and cfg.get('embed') == args.embed and
            cfg.get('snapshots') == args.snapshots and cfg.get('precision') == args.precision)
    return rec.get('traffic_bytes_per_launch') if same else None


def rollout_forward(U, g, args, dev, reps=30):
    """End-to-end forward of the whole surrogate (`build_network`, emulator.py:166-341: embeddings, 2 x L spatial layers,
    2 x 2 temporal Conv1D stacks, resnet head, flood / flow heads) on the same network with T_in = T_out = S, B = 1:
    simulated time steps per second and graph-steps/s including the non-graph tail."""
    from types import SimpleNamespace
    T = args.snapshots
    a = SimpleNamespace(state_shape=(g.n_node, 4), edge_state_shape=(g.n_edge, 4), seq_in=T, seq_out=T, embed_size=args.embed,
                        hidden_dim=64, kernel_size=3, n_sp_layer=args.layers, n_tp_layer=3, activation='relu', if_flood=3,
                        edge_fusion=True, edges=g.edges, act=False, graph=g, model_dir=None)
    X = torch.rand(1, T, g.n_node, 5, device=dev)
    B = torch.rand(1, T, g.n_node, 1, device=dev)
    E = torch.rand(1, T, g.n_edge, 4, device=dev)

    def timed(recurrent):
        emul = U.Emulator('GAT', True, recurrent, a, precision=args.precision, generator=torch.Generator().manual_seed(1)).to(dev)
        for _ in range(5):
            emul(X, B, E)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            emul(X, B, E)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps
    dt = timed('Conv1D')               # 75 of the reference's 86 shipped model configurations; 6 use LSTM, 5 the GRU default
    other = {'ms_per_forward_' + r: timed(r) * 1e3 for r in ('GRU', 'LSTM')}
    return {'ms_per_forward': dt * 1e3, **other, 'time_steps_per_s': T / dt, 'graph_steps_per_s_end_to_end': 2 * args.layers * T / dt,
            'config': 'Emulator(GAT, resnet, Conv1D) B=1 T_in=T_out=%d, %d+%d spatial layers, 3+3 temporal layers x2 sides, '
                      'flood head' % (T, args.layers, args.layers)}


def rollout_autoregressive(U, dev, steps=100):
    """BASELINE.json config 2: a real-scale network (2k nodes / 2.5k conduits), d=64, 3+3 spatial layers, 100-step rainfall
    rollout fed back step by step (seq_in 6, seq_out 1, B=1; `Emulator._model` with roll=100, emulator.py:400-438): simulated
    steps per second of the eager loop and with every step replayed from one captured HIP graph."""
    from types import SimpleNamespace
    N, E = 2000, 2500
    edges = U.synthetic_drainage_network(N, E, 0)
    g = U.DrainageGraph.from_edges(edges)
    a = SimpleNamespace(state_shape=(N, 4), edge_state_shape=(E, 4), seq_in=6, seq_out=1, embed_size=64, hidden_dim=64, kernel_size=3,
                        n_sp_layer=3, n_tp_layer=2, activation='relu', if_flood=3, edge_fusion=True, edges=edges, act=False, graph=g,
                        roll=steps, model_dir=None)
    emul = U.Emulator('GAT', True, 'Conv1D', a, generator=torch.Generator().manual_seed(1)).to(dev)
    rng = np.random.default_rng(0)
    emul.set_norm(*[np.stack([0.5 + rng.random((n, c)), np.zeros((n, c))]) for n, c in ((N, 5), (N, 1), (N, 5), (N, 1), (E, 4))])
    x, b, ex = torch.rand(1, 6, N, 5, device=dev), torch.rand(1, steps, N, 1, device=dev) * 0.1, torch.rand(1, 6, E, 4, device=dev)
    out = {}
    for name, fn in (('eager', lambda: emul._model(x, None, b, ex)), ('hip_graph', lambda: emul.rollout_graphed(x, None, b, ex))):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        out[name + '_steps_per_s'] = steps * 3 / (time.perf_counter() - t0)
    out['config'] = 'C2: N=2000 E=2500 d=64, 3+3 spatial layers, seq_in=6 seq_out=1 B=1, %d autoregressive steps' % steps
    return out


def trained_bias_leg(U, g, args, dev, x, e, reps=15):
    """The same block with the reference's trained form of NodeEdge: dense (R, M) weight and bias, the bias non-zero off the
    incidence support (reference emulator.py:36-45 -- what every checkpoint the reference trains looks like).  Per layer:
    secondary MLPs on the row-GEMM kernel, `rest @ x_e` on the split-bf16 MFMA GEMM (k_remainder_gemm2), the rest in the fused
    kernel, which adds the remainder to its NodeEdge aggregate (uds_spatial_layer_forward_rem: k_fused_ws at d = 64, the
    column-split kernel at d = 128)."""
    d, L, S = args.embed, args.layers, args.snapshots
    blk = U.SpatialBlock(g, d, L, 'relu', sparse_params=False, generator=torch.Generator().manual_seed(1), precision=args.precision).to(dev)
    with torch.no_grad():
        for ly in blk.layers:
            ly.node_edge_n.bias.normal_(0.0, 0.01)
            ly.node_edge_e.bias.normal_(0.0, 0.01)
        for _ in range(3):
            blk(x, e)                                # packs weights and remainders, builds the tile plans; warm clocks
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):      # back to back, one synchronize at the end: a rollout does not idle between its steps
            blk(x, e)
        torch.cuda.synchronize()
        ts = [(time.perf_counter() - t0) / reps]
        ly = blk.layers[0]
        x_e = ly.dense_xe(e)
        ly.node_edge_n.remainder(x_e)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            ly.node_edge_n.remainder(x_e)
        torch.cuda.synchronize()
        gemm_ms = (time.perf_counter() - t0) / reps * 1e3
    ms = float(np.median(ts)) * 1e3
    flops = 2.0 * g.n_node * g.n_edge * S * (d // 2)
    # yardstick, not part of the product: the vendor library's plain bf16 GEMM with the SAME number of MFMA flops as the
    # three split products (K tripled) -- what this part sustains in practice at this shape
    la = torch.randn(g.n_node, 3 * g.n_edge, device=dev, dtype=torch.bfloat16)
    lb = torch.randn(3 * g.n_edge, S * (d // 2), device=dev, dtype=torch.bfloat16)
    torch.matmul(la, lb)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        torch.matmul(la, lb)
    torch.cuda.synchronize()
    lib_ms = (time.perf_counter() - t0) / reps * 1e3
    del la, lb
    return {'value': L * S / (ms * 1e-3), 'unit': 'graph-steps/s', 'ms_per_step': ms, 'path': [ly.last_path for ly in blk.layers],
            'remainder_gemm': {'kernel': 'k_remainder_gemm2 (256x256 tiles, LDS-DMA staged, split-bf16, 3 MFMA products; incl. activation split + K-cut reduce)', 'ms': gemm_ms,
                               'shape': '(%d x %d) @ (%d x %d)' % (g.n_node, g.n_edge, g.n_edge, S * (d // 2)),
                               'tflops_fp32_equivalent': flops / gemm_ms / 1e9, 'tflops_bf16_issued': 3 * flops / gemm_ms / 1e9,
                               'bound': 'mfma', 'peak_tflops_bf16': 2500.0, 'frac': 3 * flops / gemm_ms / 1e9 / 2500.0,
                               'launches_per_layer': 2, 'library_bf16_gemm_same_mfma_flops_ms': lib_ms,
                               'frac_of_library_rate': lib_ms / gemm_ms},
            'note': 'NodeEdge with dense trained bias: 2*N*E*S*d/2 flops per side and layer on top of the support-only layer'}


def _cpu_model():
    try:
        with open('/proc/cpuinfo') as fh:
            for line in fh:
                if line.startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def _median_time(fn, warmup=3, reps=10, budget_s=10.0):
    """Median wall time of fn(): `warmup` untimed calls, then at least `reps` timed ones (more while the budget lasts)."""
    for _ in range(warmup):
        fn()
    ts, t_all = [], time.perf_counter()
    while len(ts) < reps or (time.perf_counter() - t_all < budget_s and len(ts) < 200):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), len(ts)


def cpu_baseline(U, g, block_params, d, gpu_out=None, gpu_in=None):
    """The oracle (the build's PyTorch-CPU restatement of the reference forward, kind 'port': the reference's TF path
    cannot run here) timed on this host's cores on a bounded sample, SURVEY.md section 8d: (i) the sparse-CSR form on the
    benchmark network, (ii) the dense-masked form -- what the reference computes -- on a C2-size network (N = 2 000: the
    dense logits of the headline network would be 400 MB per snapshot); fp32, median of >= 10 repetitions after 3
    warm-ups.  Also the observed error of the GPU result on one snapshot of the timed configuration against the same
    oracle in fp64 (`parity_err_vs_fp64`): what backs the `dtype` claim of the bench line."""
    from oracle import sparse_csr as OS
    from oracle import spektral_dense as OD
    cores = min(os.cpu_count() or 1, 16)      # a 1-GPU box's CPU share is 16 cores; more threads only thrash
    torch.set_num_threads(cores)
    L, S = len(block_params), 2
    gen = torch.Generator().manual_seed(2)
    x = torch.rand(S, g.n_node, d, generator=gen)
    e = torch.rand(S, g.n_edge, d, generator=gen)
    adj, eadj = (g.adj.rowptr, g.adj.col), (g.edge_adj.rowptr, g.edge_adj.col)
    inc_n, inc_e = (g.inc_n.rowptr, g.inc_n.col), (g.inc_e.rowptr, g.inc_e.col)

    def sparse_pass(xx=x, ee=e, params=block_params):
        for p in params:
            xx, ee = OS.spatial_layer_csr(xx, ee, p, adj, eadj, inc_n, inc_e)
        return xx, ee

    t_sparse, n_sparse = _median_time(sparse_pass)
    out = {'value': L * S / t_sparse, 'unit': 'graph-steps/s', 'cores': cores, 'kind': 'port', 'cpu_model': _cpu_model(),
           'sample': 'median of %d passes (after 3 warm-ups) of the %d-layer block on S=%d snapshots of the same network, sparse-CSR '
                     'torch-CPU fp32 oracle (%.3f s per pass)' % (n_sparse, L, S, t_sparse)}
    # (ii) the dense-masked formulation (Spektral's, i.e. the reference's) on a C2-size network
    g2 = U.DrainageGraph.from_edges(U.synthetic_drainage_network(2000, 2500, seed=0))
    A, EA, NE = (torch.from_numpy(m.to_dense()).float() for m in (g2.adj, g2.edge_adj, g2.inc_n))
    blk = U.SpatialBlock(g2, d, L, 'relu', sparse_params=False, generator=torch.Generator().manual_seed(1))     # CPU parameters only
    p2 = [ly.export_params() for ly in blk.layers]
    x2, e2 = torch.rand(S, 2000, d, generator=gen), torch.rand(S, 2500, d, generator=gen)

    def dense_pass():
        xx, ee = x2, e2
        for p in p2:
            xx, ee = OD.spatial_layer_dense(xx, ee, p, A, EA, NE)
        return xx

    t_dense, n_dense = _median_time(dense_pass, budget_s=6.0)
    p2s = [U.SpatialLayer(g2, d, 'relu', sparse_params=True, generator=torch.Generator().manual_seed(1)).export_params() for _ in range(L)]
    adj2, eadj2 = (g2.adj.rowptr, g2.adj.col), (g2.edge_adj.rowptr, g2.edge_adj.col)
    incn2, ince2 = (g2.inc_n.rowptr, g2.inc_n.col), (g2.inc_e.rowptr, g2.inc_e.col)

    def sparse2_pass():
        xx, ee = x2, e2
        for p in p2s:
            xx, ee = OS.spatial_layer_csr(xx, ee, p, adj2, eadj2, incn2, ince2)
        return xx

    t_sp2, _ = _median_time(sparse2_pass, budget_s=3.0)
    out['dense_masked_c2'] = {'value': L * S / t_dense, 'unit': 'graph-steps/s', 'sample': 'median of %d passes, N=2000 E=2500 d=%d, %d layers, S=%d, '
                              'dense-masked (Spektral) torch-CPU fp32 oracle' % (n_dense, d, L, S), 'sparse_csr_same_network': L * S / t_sp2}
    if gpu_out is not None:      # observed error of the timed GPU block on its first snapshot against the fp64 oracle
        x64, e64 = gpu_in[0][:1].double().cpu(), gpu_in[1][:1].double().cpu()
        p64 = [{k: (v.double() if isinstance(v, torch.Tensor) else v) for k, v in p.items()} for p in block_params]
        rx, re = sparse_pass(x64, e64, p64)
        ex = float((gpu_out[0][:1].double().cpu() - rx).abs().max()) / max(1.0, float(rx.abs().max()))
        ee = float((gpu_out[1][:1].double().cpu() - re).abs().max()) / max(1.0, float(re.abs().max()))
        out['parity_err_vs_fp64'] = max(ex, ee)
    return out


def bench_c4(args, U, dist, world, rank, dev):
    """200k-node network, node-cut over `world` ranks, halo exchange after every layer but the last (strong scaling).
    Returns the record on rank 0 (None elsewhere); the caller prints it (`--workload c4`) or embeds it in the headline line."""
    from gnn_uds_amd import dist as D
    nodes = 200000 if args.nodes == 10000 else args.nodes
    links = 240000 if args.links == 12000 else args.links
    S = 8 if args.snapshots == 60 else args.snapshots
    d, L = args.embed, args.layers
    g = U.DrainageGraph.from_edges(U.synthetic_drainage_network(nodes, links, seed=0))
    part, part_info = D.partition_nodes(g, world, return_info=True)
    probs = D.build_partition_plan(g, world, part)
    prob = probs[rank]
    ref = U.SpatialBlock(g, d, L, 'relu', sparse_params=True, generator=torch.Generator().manual_seed(1))   # CPU parameters only
    params = [ly.export_params() for ly in ref.layers]
    layers = D.hip_layers(prob, params, d, 'relu', args.precision, dev)
    block = D.ShardedSpatialBlock(prob, L, lambda p, i, x, e: layers[i](x, e), dev)
    gen = torch.Generator().manual_seed(2)
    x, e = block.scatter_inputs(torch.rand(S, nodes, d, generator=gen), torch.rand(S, links, d, generator=gen))
    x, e = x.to(dev), e.to(dev)
    for _ in range(args.warmup):
        block.forward(x, e)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        block.forward(x, e)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    wall = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([wall], device=dev if dist.get_backend() == 'nccl' else 'cpu', dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    # what the exchange costs and how much of it the snapshot-group pipeline leaves exposed (max over ranks, per layer that
    # exchanges): compute only (no exchange, results wrong but same kernels) | in-order compute-then-exchange | pipelined
    def timed(fn, reps=3):
        fn()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        t = torch.tensor([(time.perf_counter() - t0) / reps * 1e3], dtype=torch.float64)
        if dist is not None:
            t = t.to(dev if dist.get_backend() == 'nccl' else 'cpu')
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def compute_only():
        xx, ee = x, e
        for ly in layers:
            xx, ee = ly(xx, ee)
    t_comp, t_inorder, t_pipe = timed(compute_only), timed(lambda: block.forward(x, e, stages=1)), timed(lambda: block.forward(x, e))
    rec = None
    if rank == 0:
        bytes_gs = algorithmic_bytes_per_graph_step(g, d, d)
        achieved = args.steps * L * S * bytes_gs / world / wall / 1e9        # per GPU: each rank streams 1/world of the network
        rec = ({
            'metric': 'graph-steps/sec (forward rollout), 200k-node network partitioned over the GPUs', 'value': args.steps * L * S / wall,
            'unit': 'graph-steps/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': wall / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f32' if args.precision == 'fp32' else
            'f32 storage/accumulate, GEMM operands as bf16 hi+lo split (3 MFMA products)', 'data': 'synthetic',
            'config': {'workload': 'C4 synthetic mega-catchment N=%d E=%d d=%d, %d-layer GAT spatial block, S=%d snapshots, '
                                   '%d-way node cut with per-layer halo exchange' % (nodes, links, d, L, S, world),
                       'own_nodes': int(len(prob.own_nodes)), 'halo_nodes': int(len(prob.nodes) - len(prob.own_nodes)),
                       'halo_links': int(len(prob.links) - len(prob.own_links)), 'peers': len(block.exchange.peers),
                       'halo_bytes_per_layer': block.exchange.bytes_per_layer(S, d), 'precision': args.precision,
                       'partition': {'cut_links': part_info['refined'], 'candidates': {k: part_info[k] for k in ('id', 'bfs', 'rcm')},
                                     'kept': part_info['kept'] + ' + boundary refinement',
                                     'per_rank': [{'own_nodes': int(len(q.own_nodes)), 'own_links': int(len(q.own_links)),
                                                   'halo_nodes': int(len(q.nodes) - len(q.own_nodes)),
                                                   'halo_links': int(len(q.links) - len(q.own_links)),
                                                   'rows_sent_per_layer': int(sum(len(v) for v in q.send_nodes.values()) +
                                                                              sum(len(v) for v in q.send_links.values())),
                                                   'peers': len(set(q.send_nodes) | set(q.recv_nodes))} for q in probs]}},
            'halo': {'compute_only_ms': t_comp, 'in_order_ms': t_inorder, 'pipelined_ms': t_pipe,
                     'halo_ms_per_layer': (t_inorder - t_comp) / max(1, L - 1), 'exposed_ms_per_layer': (t_pipe - t_comp) / max(1, L - 1),
                     'note': 'exchange hidden by pipelining over 2 snapshot groups on a side stream (dist.ShardedSpatialBlock.forward)'},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBPS,
                         'traffic': None, 'kernel': 'k_fused_ws (per rank, its part of the network)'}})
    return rec


def bench_small(args, U, dev):
    """The regime the reference actually ships (all five networks have 30-443 nodes): one spatial layer
    on S = 4 096 snapshots (a batch of rainfall scenarios x time steps) of each shipped network, d = 64 and d = 128
    (`embed_size` default 128, utils/config.yaml:39).  A small network is one or two tiles, so the launch is
    (tiles x snapshot chunks) workgroups of the same fused kernels: graph-steps/s, fraction of the HBM roofline, and how full
    the tiles and the 256 CUs are.  Link lists: tests/golden/networks.json (data extracted from the reference's .inp files)."""
    with open(os.path.join(ROOT, 'tests', 'golden', 'networks.json')) as fh:
        nets = json.load(fh)
    S = 4096 if args.snapshots == 60 else args.snapshots
    legs = []
    for name in ('astlingen', 'shunqing', 'chaohu', 'hague', 'RedChicoSur'):
        net = nets[name]
        g = U.DrainageGraph.from_edges(np.array(net['edges']), net['n_node'])
        for d in (64, 128):
            layer = U.SpatialLayer(g, d, 'relu', sparse_params=True, generator=torch.Generator().manual_seed(1), precision=args.precision).to(dev)
            gen = torch.Generator().manual_seed(2)
            x, e = torch.rand(S, g.n_node, d, generator=gen).to(dev), torch.rand(S, g.n_edge, d, generator=gen).to(dev)
            for _ in range(3):
                layer(x, e)
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
            for a_, b_ in ev:
                a_.record(); layer(x, e); b_.record()
            torch.cuda.synchronize()
            ms_eager = float(np.median([a_.elapsed_time(b_) for a_, b_ in ev]))
            # the same call as a captured HIP graph (what a rollout on a small network replays, Emulator.rollout_graphed /
            # SpatialBlock.graphed): the eager figure of a 30-node network is the host's launch path, not the kernel
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side), torch.no_grad():
                layer(x, e)
            torch.cuda.current_stream(dev).wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph), torch.no_grad():
                layer(x, e)
            for _ in range(20):
                graph.replay()
            torch.cuda.synchronize()
            a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a_.record()
            for _ in range(100):
                graph.replay()
            b_.record()
            torch.cuda.synchronize()
            ms = a_.elapsed_time(b_) / 100
            k = layer.pack_factor() if S >= 16 * layer.pack_factor() else 1
            info = (layer._replica(k) if k > 1 else layer.network()).plan_info()
            bytes_gs = algorithmic_bytes_per_graph_step(g, d, d)
            achieved = S * bytes_gs / (ms * 1e-3) / 1e9
            tiles = info['node_tiles'] + info['link_tiles']
            legs.append({'network': name, 'nodes': g.n_node, 'links': g.n_edge, 'd': d, 'snapshots': S, 'ms_per_launch': ms,
                         'ms_per_launch_eager': ms_eager, 'launch': 'replay of a captured HIP graph (eager: one C-ABI call from Python)',
                         'graph_steps_per_s': S / (ms * 1e-3), 'achieved_GBps': achieved, 'frac': achieved / HBM_PEAK_GBPS,
                         'frac_eager': S * bytes_gs / (ms_eager * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                         'path': layer.last_path, 'snapshots_side_by_side': k, 'tiles': tiles, 'p_cap': info['p_cap'], 'q_cap': info['q_cap'],
                         'rows_per_tile': k * (g.n_node + g.n_edge) / max(1, tiles)})
    print(json.dumps({'metric': 'graph-steps/sec, one spatial layer on the reference\'s shipped networks (30-443 nodes), S=%d snapshots' % S,
                      'unit': 'graph-steps/s', 'n_gpus': 1, 'dtype': 'f32 storage/accumulate, GEMM operands as bf16 hi+lo split',
                      'data': 'link lists of the five shipped networks, synthetic features', 'legs': legs}))


def bench_c5(args, U, dist, world, rank, dev):
    """BASELINE.json config 5: training forward + backward on a block-diagonal batch of mini-graphs (default 1 000 graphs
    x 2 000 nodes / 2 500 links, split over the ranks), L-layer GAT spatial block, MSE loss, reverse mode through the HIP
    backward kernels, one bucketed gradient all-reduce, Adam step.  Unit: one mini-graph through one layer, forward and
    backward ("training graph-step")."""
    from gnn_uds_amd import dist as D
    from gnn_uds_amd.emulator import KerasAdam
    G_total = 1000 if args.snapshots == 60 else args.snapshots           # --snapshots doubles as the number of mini-graphs
    n1 = 2000 if args.nodes == 10000 else args.nodes
    e1 = 2500 if args.links == 12000 else args.links
    G = max(1, G_total // world)
    d, L = args.embed, args.layers
    edges = np.concatenate([U.synthetic_drainage_network(n1, e1, seed=rank * G + k) + n1 * k for k in range(G)])
    g = U.DrainageGraph.from_edges(edges, n1 * G)
    block = U.SpatialBlock(g, d, L, 'relu', sparse_params=True, generator=torch.Generator().manual_seed(1), precision=args.precision).to(dev)
    block.requires_grad_(True)
    params = list(block.parameters())
    opt = KerasAdam(params, 1e-3, clipnorm=1.0)
    gen = torch.Generator().manual_seed(2 + rank)
    x, e = torch.rand(1, g.n_node, d, generator=gen).to(dev), torch.rand(1, g.n_edge, d, generator=gen).to(dev)
    tx, te = torch.rand(1, g.n_node, d, generator=gen).to(dev), torch.rand(1, g.n_edge, d, generator=gen).to(dev)

    def step():
        for p in params:
            p.grad = None
        ox, oe = block(x, e)
        loss = torch.nn.functional.mse_loss(ox, tx) + torch.nn.functional.mse_loss(oe, te)      # mean squared error, fused fwd / bwd kernels
        loss.backward()
        D.allreduce_gradients(params)
        opt.step()
        return loss

    for _ in range(args.warmup):
        step()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    wall = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([wall], device=dev if dist.get_backend() == 'nccl' else 'cpu', dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    if rank == 0:
        units = args.steps * L * G * world
        print(json.dumps({
            'metric': 'training graph-steps/sec (one mini-graph through one spatial layer, forward + backward)', 'value': units / wall,
            'unit': 'graph-steps/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': wall / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32' if args.precision == 'fp32' else 'f32 storage/accumulate, GEMM operands as bf16 hi+lo split (3 MFMA products)',
            'data': 'synthetic',
            'config': {'workload': 'C5 training: %d mini-graphs x (N=%d, E=%d) per GPU as one block-diagonal batch, %d-layer GAT spatial '
                                   'block d=%d, MSE loss, backward, gradient all-reduce, Adam(clipnorm=1)' % (G, n1, e1, L, d),
                       'precision': args.precision, 'final_loss': float(loss)},
            # a training graph-step must at least read the layer's inputs and write its outputs (forward), read them again with the
            # output gradients and write the input gradients (backward): 3 x the forward's algorithmic bytes per mini-graph and layer.
            # The training path is the unfused chain (every operator with its own backward kernel), so it sits far from that bound.
            'roofline': {'bound': 'hbm', 'achieved': 3 * algorithmic_bytes_per_graph_step(g, d, d) / G * (units / world / wall) / 1e9,
                         'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                         'frac': 3 * algorithmic_bytes_per_graph_step(g, d, d) / G * (units / world / wall) / 1e9 / HBM_PEAK_GBPS, 'traffic': None,
                         'algorithmic_bytes_per_graph_step': 3 * algorithmic_bytes_per_graph_step(g, d, d) / G,
                         'kernel': 'unfused training chain (row GEMMs, k_gat_aggregate / k_gat_bwd_rows / k_gat_bwd_cols, k_csr_spmm, '
                                   'k_csr_sddmm, k_wgrad_mfma: profiles/r03h_c5_kernel_stats.csv -- no kernel above 10 % of the step)'}}))
    if dist is not None:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    # 200 timed steps (600 launches, 0.13 s) behind 20 warm-up steps: the chip takes ~50 launches to reach its steady clock after
    # the idle gap of the barrier + synchronize in front of the timed region (260 -> 227 us per launch over the first
    # 60 launches); a 20-step region measures that ramp, not the rollout it stands for
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--nodes', type=int, default=10000)
    ap.add_argument('--links', type=int, default=12000)
    ap.add_argument('--embed', type=int, default=64)
    ap.add_argument('--layers', type=int, default=3)
    ap.add_argument('--snapshots', type=int, default=60, help='S = B*T per rank (default B=1, T=60)')
    ap.add_argument('--precision', default='bf16x3', choices=['bf16x3', 'fp32'],
                    help="bf16x3: fused kernel, GEMM operands split into bf16 hi+lo (3 MFMA products, fp32 accumulate); "
                         "fp32: exact-fp32 unfused kernels")
    ap.add_argument('--workload', default='headline', choices=['headline', 'c4', 'c5', 'small'],
                    help='c4: 200k-node / 240k-link network partitioned over the ranks with per-layer halo exchange')
    ap.add_argument('--graph', action='store_true', help='replay the L layer launches of a step from one captured HIP graph')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-trained-bias', action='store_true', help='skip the dense-trained-bias NodeEdge leg')
    ap.add_argument('--autoregressive', action='store_true',
                    help='also time the C2 autoregressive rollout (100 fed-back steps, eager and HIP graph); off by default so that '
                         'the rocprofv3 averages of the default command are those of the timed headline launches')
    ap.add_argument('--no-rollout', action='store_true', help='skip the whole-Emulator forward legs (profiling runs: only the timed headline launches)')
    ap.add_argument('--no-c4', action='store_true', help='skip the 200k-node partitioned leg of the headline line')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N` (how the driver calls it): start the N ranks here, as a CHILD process and before
        # anything in this process touches the GPU (never exec / fork from a process that has initialised HIP), relay
        # what they print -- rank 0's one JSON line -- and leave with the launcher's exit code.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
               '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
        for line in res.stdout.splitlines():      # rank 0's JSON line to stdout, anything else the ranks printed (gloo chatter) to stderr
            print(line, file=sys.stdout if line.startswith('{') else sys.stderr)
        raise SystemExit(res.returncode)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d (launch one rank per GPU, or let `python bench.py --gpus N` start them)'
                         % (args.gpus, world))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the engine has no CPU path')
    # UDS_DIST_BACKEND=gloo rehearses the N > 1 path on a box with fewer GPUs than ranks (ranks share the cards, the timing
    # reduction goes through the host); the driver's runs use the default: one rank per GPU over RCCL
    backend = os.environ.get('UDS_DIST_BACKEND', 'nccl')
    if backend != 'nccl':
        local %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)

    import gnn_uds_amd as U

    if args.workload == 'small':
        return bench_small(args, U, dev)
    if args.workload == 'c5':
        return bench_c5(args, U, dist if world > 1 else None, world, rank, dev)
    if args.workload == 'c4':
        rec = bench_c4(args, U, dist, world, rank, dev)
        if rank == 0:
            print(json.dumps(rec))
        if dist is not None:
            dist.destroy_process_group()
        return
    g = U.DrainageGraph.from_edges(U.synthetic_drainage_network(args.nodes, args.links, seed=0))
    d, L, S = args.embed, args.layers, args.snapshots
    # random-init weights of the reference architecture (Keras initialisers: glorot_uniform kernels, zero
    # biases, N(0, 0.05^2) NodeEdge weights), seed 1
    block = U.SpatialBlock(g, d, L, 'relu', sparse_params=True, generator=torch.Generator().manual_seed(1),
                           precision=args.precision).to(dev)
    params = [layer.export_params() for layer in block.layers]
    gen = torch.Generator().manual_seed(2 + rank)
    x = torch.rand(S, g.n_node, d, generator=gen).to(dev)
    e = torch.rand(S, g.n_edge, d, generator=gen).to(dev)

    # one step = the L layer launches from Python; `--graph` replays them from ONE captured HIP graph instead (SpatialBlock.graphed).
    # At the headline size the two are equal within noise (231-232 k vs 233-235 k graph-steps/s: the host runs ahead of 250-us kernels)
    replay = block.graphed(x, e) if args.graph else None

    def step():
        return replay() if replay is not None else block(x, e)

    for _ in range(args.warmup):
        step()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()          # events on torch's current stream = the stream the kernels are launched on
        step()
        b.record()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    wall = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([wall], device=dev if dist.get_backend() == 'nccl' else 'cpu', dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    dev_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))      # device time of one step

    # BASELINE.json's partitioned case beside the headline (every N, so that the N = 1 line is the base of the strong-scaling
    # ratio): the 200k-node network node-cut over the ranks, boundary rows exchanged after every layer
    c4 = None
    if not args.no_c4 and args.embed == 64 and args.precision == 'bf16x3' and (args.nodes, args.links) == (10000, 12000):
        c4 = bench_c4(argparse.Namespace(**dict(vars(args), steps=min(args.steps, 50), warmup=min(args.warmup, 10), snapshots=60)),
                      U, dist, world, rank, dev)

    if rank == 0:
        gsteps = args.steps * L * S * world
        bytes_gs = algorithmic_bytes_per_graph_step(g, d, d)
        achieved = (L * S * bytes_gs) / (dev_ms * 1e-3) / 1e9
        out = {
            'metric': 'graph-steps/sec (forward rollout), |V|=%gk |E|=%gk d=%d' % (args.nodes / 1000, args.links / 1000, d), 'value': gsteps / wall,
            'unit': 'graph-steps/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': wall / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32' if args.precision == 'fp32' else 'f32 storage/accumulate, GEMM operands as bf16 hi+lo split (3 MFMA products)',
            'data': 'synthetic',
            'config': {'workload': 'headline synthetic drainage network N=%d E=%d d=%d, %d-layer GAT spatial block, '
                                   'S=%d snapshots per GPU (B=1,T=%d)' % (g.n_node, g.n_edge, d, L, S, S),
                       'graph_steps_per_step': L * S, 'nnz_node': g.adj.nnz, 'nnz_line': g.edge_adj.nnz,
                       'parallelism': 'snapshot-sharded x%d' % world, 'precision': args.precision,
                       'launch': 'eager (one C-ABI call per layer from Python)' if not args.graph else
                                 'one captured HIP graph per step (SpatialBlock.graphed): %d kernel launches replayed back to back' % L,
                       'node_edge': 'support-only parameters (one weight and bias per incidence entry): equals the reference model at '
                                    'its initialisation (bias zeros) or with a bias trained on the support; a bias trained off the '
                                    'support is the `trained_bias` leg',
                       'plan': block.layers[0].network().plan_info()},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBPS, 'traffic': recorded_traffic(args),
                         'algorithmic_bytes_per_launch': S * bytes_gs, 'launch_ms': dev_ms / L,
                         'kernel': ('k_fused_cs<128,128,128,8,relu>' if d == 128 else 'k_fused_ws<relu>') +
                                   ' (one launch per layer over S snapshots)' if args.precision == 'bf16x3' else
                                   'uds_spatial_layer_forward, unfused (8 launches per layer over S snapshots)',
                         'algorithmic_bytes_per_graph_step': bytes_gs, 'device_ms_per_step': dev_ms,
                         'timed_region_ms': wall * 1e3, 'run_to_run_spread': 'up to 10 % between the boxes of the pool (268-299 k graph-steps/s for the same code), 1-2 % run to run on one box; the first ~50 launches behind an idle moment run 15 % slower (clock ramp): the default region is 200 steps, --steps 20 --warmup 3 reproduces the region of rounds 1-2 (DESIGN.md 7.00)'},
        }
        if c4 is not None:
            out['c4_partitioned'] = {k: c4[k] for k in ('metric', 'value', 'unit', 'n_gpus', 'ms_per_step', 'scaling', 'config', 'halo', 'roofline')}
        if world == 1 and args.embed == 64 and not args.no_rollout:
            out['rollout'] = rollout_forward(U, g, args, dev)
            if args.autoregressive:
                out['rollout']['autoregressive'] = rollout_autoregressive(U, dev)
        if world == 1 and args.embed in (64, 128) and args.precision == 'bf16x3' and not args.no_trained_bias \
                and g.n_node * g.n_edge <= (1 << 28):
            out['trained_bias'] = trained_bias_leg(U, g, args, dev, x, e)
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(U, g, params, d, gpu_out=step(), gpu_in=(x, e))
            # max |GPU - fp64 oracle| / max(1, max|oracle|) over all rows of the first snapshot after the whole L-layer block
            out['roofline']['parity_err_vs_fp64'] = cb.pop('parity_err_vs_fp64', None)
            out['cpu_baseline'] = cb
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
