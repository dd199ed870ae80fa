"""Golden forward vectors for BASELINE config C1 (50 junctions / 60 conduits, 1 spatial layer, d=8) and a d=64 case.

Run in the build container:  python tests/golden/make_forward_fixtures.py
The reference itself cannot run here (TensorFlow / Spektral absent), so these vectors are produced by THIS repo's
fp64 dense-masked oracle (oracle/spektral_dense.py) from seeded inputs and seeded weights (Keras initialisers): they
pin the oracle against regressions and give the GPU tests a committed target; they do not pin it against the reference
("parity unpinned", oracle/__init__.py).  Output: tests/golden/forward_c1.npz (float64 arrays, ~400 kB: the dense (N,E) NodeEdge parameters dominate).
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gnn_uds_amd.graph import DrainageGraph, synthetic_drainage_network   # noqa: E402
from oracle import spektral_dense as OD                                    # noqa: E402
from tests.util import spatial_params                                      # noqa: E402


def case(n, m, d, S, seed):
    g = DrainageGraph.from_edges(synthetic_drainage_network(n, m, seed=0))
    p = spatial_params(n, m, d, d, d, seed=seed)
    gen = torch.Generator().manual_seed(seed + 100)
    x = torch.rand(S, n, d, generator=gen, dtype=torch.float64)
    e = torch.rand(S, m, d, generator=gen, dtype=torch.float64)
    ox, oe = OD.spatial_layer_dense(x, e, p, torch.from_numpy(g.adj.to_dense()), torch.from_numpy(g.edge_adj.to_dense()),
                                    torch.from_numpy(g.inc_n.to_dense()))
    return g, p, x, e, ox, oe


if __name__ == '__main__':
    out = {}
    for tag, (n, m, d, S, seed) in {'c1': (50, 60, 8, 1, 1), 'd64': (50, 60, 64, 2, 2)}.items():
        g, p, x, e, ox, oe = case(n, m, d, S, seed)
        out[tag + '_edges'] = g.edges
        out[tag + '_x'], out[tag + '_e'], out[tag + '_out_x'], out[tag + '_out_e'] = x.numpy(), e.numpy(), ox.numpy(), oe.numpy()
        for k, v in p.items():
            out['%s_p_%s' % (tag, k)] = v.numpy()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'forward_c1.npz')
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path))
