"""Experiment: the 3-layer block on S=60 snapshots as ONE chain vs K independent snapshot groups on K HIP streams
(snapshots are independent, so the tail of one group's launch overlaps the next launch of another group)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_uds_amd as U

dev = torch.device('cuda', 0)
g = U.DrainageGraph.from_edges(U.synthetic_drainage_network(10000, 12000, seed=0))
block = U.SpatialBlock(g, 64, 3, 'relu', sparse_params=True, generator=torch.Generator().manual_seed(1)).to(dev)
S = 60
x, e = torch.rand(S, g.n_node, 64, device=dev), torch.rand(S, g.n_edge, 64, device=dev)


def run(K):
    streams = [torch.cuda.Stream(device=dev) for _ in range(K)]
    parts = [(x[i * S // K:(i + 1) * S // K], e[i * S // K:(i + 1) * S // K]) for i in range(K)]

    def step():
        if K == 1:
            return [block(x, e)]
        cur = torch.cuda.current_stream(dev)
        outs = []
        for st, (xp, ep) in zip(streams, parts):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                outs.append(block(xp, ep))
        for st in streams:
            cur.wait_stream(st)
        return outs
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print('K=%d streams: %.3f ms per step = %.1fk graph-steps/s' % (K, dt * 1e3, 3 * S / dt / 1e3))


for K in (1, 2, 3, 4):
    run(K)
