"""Where do the strided float copies of the C5 training step come from?  torch profiler with Python stacks, grouped by caller."""
import os, sys, collections
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_uds_amd as U
from gnn_uds_amd.emulator import KerasAdam

dev = torch.device('cuda', 0)
G, n1, e1, d, L = 200, 2000, 2500, 64, 3
edges = np.concatenate([U.synthetic_drainage_network(n1, e1, seed=k) + n1 * k for k in range(G)])
g = U.DrainageGraph.from_edges(edges, n1 * G)
block = U.SpatialBlock(g, d, L, 'relu', sparse_params=True, generator=torch.Generator().manual_seed(1)).to(dev)
block.requires_grad_(True)
params = list(block.parameters())
opt = KerasAdam(params, 1e-3, clipnorm=1.0)
x, e = torch.rand(1, g.n_node, d, device=dev), torch.rand(1, g.n_edge, d, device=dev)
tx, te = torch.rand(1, g.n_node, d, device=dev), torch.rand(1, g.n_edge, d, device=dev)


def step():
    for p in params:
        p.grad = None
    ox, oe = block(x, e)
    loss = ((ox - tx) ** 2).mean() + ((oe - te) ** 2).mean()
    loss.backward()
    opt.step()


step(); step()
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA], with_stack=True,
                            record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
agg = collections.Counter()
tim = collections.Counter()
for ev in prof.events():
    if ev.name in ('aten::copy_', 'aten::contiguous', 'aten::clone', 'aten::add', 'aten::add_', 'aten::mul', 'aten::cat') and ev.device_time_total > 20:
        st = [s for s in ev.stack if 'gnn_uds_amd' in s or 'c5_copy' in s][:2]
        key = (ev.name, str(ev.input_shapes)[:60], ' <- '.join(s.split('/')[-1] for s in st))
        agg[key] += 1
        tim[key] += ev.device_time_total
for k, v in sorted(tim.items(), key=lambda kv: -kv[1])[:30]:
    print('%8.0f us x%d  %s' % (v, agg[k], k))
