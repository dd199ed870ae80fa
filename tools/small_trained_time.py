"""One spatial layer with a TRAINED dense NodeEdge bias on the reference's shipped networks (S = 4096 snapshots): ms per layer call
beside the support-only layer (captured HIP graphs, 50 replays)."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_uds_amd as U
dev = torch.device('cuda', 0)
nets = json.load(open(os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'networks.json')))
S = 4096

def timed(layer, x, e):
    with torch.no_grad():
        for _ in range(3):
            layer(x, e)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            layer(x, e)
        torch.cuda.current_stream().wait_stream(side)
        with torch.cuda.graph(g):
            layer(x, e)
        for _ in range(10):
            g.replay()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50):
            g.replay()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / 50

for name in ('astlingen', 'shunqing', 'RedChicoSur'):
    net = nets[name]
    gph = U.DrainageGraph.from_edges(np.array(net['edges']), net['n_node'])
    for d in (64, 128):
        x, e = torch.rand(S, gph.n_node, d, device=dev), torch.rand(S, gph.n_edge, d, device=dev)
        sup = U.SpatialLayer(gph, d, 'relu', sparse_params=True, generator=torch.Generator().manual_seed(1)).to(dev)
        trn = U.SpatialLayer(gph, d, 'relu', sparse_params=False, generator=torch.Generator().manual_seed(1)).to(dev)
        with torch.no_grad():
            trn.node_edge_n.bias.normal_(0.0, 0.01)
            trn.node_edge_e.bias.normal_(0.0, 0.01)
        t0, t1 = timed(sup, x, e), timed(trn, x, e)
        print(json.dumps({'network': name, 'd': d, 'support_only_ms': round(t0, 4), 'trained_ms': round(t1, 4), 'path': trn.last_path, 'ratio': round(t1 / t0, 2)}), flush=True)
