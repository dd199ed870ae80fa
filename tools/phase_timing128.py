"""Diagnostic: per-phase cycle shares of k_fused128 (library built with UDS_PHASE_TIMING=1)."""
import os, sys, ctypes
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_uds_amd as U
from gnn_uds_amd import _lib
dev = torch.device('cuda', 0)
g = U.DrainageGraph.from_edges(U.synthetic_drainage_network(10000, 12000, 0))
S, d = 60, 128
layer = U.SpatialLayer(g, d, 'relu', sparse_params=True, generator=torch.Generator().manual_seed(1)).to(dev)
x, e = torch.rand(S, 10000, d, device=dev), torch.rand(S, 12000, d, device=dev)
lib = _lib.load()
net = layer.network(); net.prepare(128, 128)
vn, _ = layer.node_edge_n.support_values(); ve, _ = layer.node_edge_e.support_values()
p = {k: (v.to(dev) if v is not None else None) for k, v in layer.export_params().items() if not k.startswith('ne_')}
p.update(ne_n_val=vn, ne_e_val=ve)
sp = _lib._spatial_params(p)
ws = torch.zeros(lib.uds_spatial_workspace_floats(net.ptr, S, 64, d), device=dev)
ox, oe = torch.empty(S, 10000, d, device=dev), torch.empty(S, 12000, d, device=dev)
for _ in range(2):
    ws.zero_()
    rc = lib.uds_spatial_layer_forward(net.ptr, ctypes.byref(sp), x.data_ptr(), d, e.data_ptr(), d, S, 64, d, 1, 0, ws.data_ptr(), ox.data_ptr(), oe.data_ptr(), None)
    assert rc == 0, lib.uds_last_error()
torch.cuda.synchronize()
raw = ws[65536:].view(torch.int64).cpu().numpy()
rows = raw[:(len(raw) // 16) * 16].reshape(-1, 16)
rows = rows[(rows[:, 13] & 1) == 1]
names = ['setup', 'wait', 'P0', 'bar0', 'P1', 'bar1', 'P1.5', 'bar1.5', 'P2', 'bar2', 'P3']
print('waves', len(rows), 'mean total cycles', int(rows[:, :11].sum(1).mean()))
tot = rows[:, :11].sum()
print('shares', {n: round(float(rows[:, k].sum() / tot), 3) for k, n in enumerate(names)})
for wv in range(8):
    r = rows[rows[:, 12] == wv]
    print('wave', wv, {n: int(r[:, k].mean()) for k, n in enumerate(names)})
