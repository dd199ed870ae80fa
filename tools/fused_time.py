"""Time the fused spatial-layer launch of the library in UDS_LIB_PATH on the headline network and dump a slice of its
output (child process of tools/variant_bench.py; one library per process).
    UDS_LIB_PATH=build_variants/x.so python tools/fused_time.py OUT.pt [S] [reps] [fx] [fe]"""
import os, sys, json
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_uds_amd as U

out_path = sys.argv[1]
S = int(sys.argv[2]) if len(sys.argv) > 2 else 60
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
FX = int(sys.argv[4]) if len(sys.argv) > 4 else 64
FE = int(sys.argv[5]) if len(sys.argv) > 5 else 64
N, E = int(os.environ.get('NODES', 10000)), int(os.environ.get('LINKS', 12000))
dev = torch.device('cuda', 0)
g = U.DrainageGraph.from_edges(U.synthetic_drainage_network(N, E, 0))
layer = U.SpatialLayer(g, 64, 'relu', fx=FX, fe=FE, sparse_params=True, generator=torch.Generator().manual_seed(1)).to(dev)
gen = torch.Generator().manual_seed(2)
x, e = torch.rand(S, N, FX, generator=gen).to(dev), torch.rand(S, E, FE, generator=gen).to(dev)
for _ in range(3):
    ox, oe = layer(x, e)
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
for a, b in ev:
    a.record(); layer(x, e); b.record()
torch.cuda.synchronize()
t = np.array([a.elapsed_time(b) for a, b in ev]) * 1e3
sl = slice(0, S, max(1, S // 2))
torch.save({'x': ox[sl].cpu(), 'e': oe[sl].cpu()}, out_path)
print(json.dumps({'lib': os.environ.get('UDS_LIB_PATH', 'default'), 'us_median': float(np.median(t)), 'us_min': float(t.min()),
                  'us_mean': float(t.mean()), 'plan': layer.network().plan_info()}))
