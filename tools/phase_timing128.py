"""Diagnostic: per-phase cycle shares of the d = 128 column-split kernel (k_fused_cs<128,128,128,8>; needs a library built with
-DUDS_PHASE_TIMING: UDS_DEFINES=-DUDS_PHASE_TIMING python tools/build_variant.py pt, then UDS_LIB_PATH=build_variants/pt.so)."""
import os, sys, ctypes
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_uds_amd as U
from gnn_uds_amd import _lib

dev = torch.device('cuda', 0)
g = U.DrainageGraph.from_edges(U.synthetic_drainage_network(10000, 12000, 0))
S = int(sys.argv[1]) if len(sys.argv) > 1 else 60
layer = U.SpatialLayer(g, 128, 'relu', fx=128, fe=128, sparse_params=True, generator=torch.Generator().manual_seed(1)).to(dev)
x, e = torch.rand(S, 10000, 128, device=dev), torch.rand(S, 12000, 128, device=dev)
lib = _lib.load()
net = layer.network()
net.prepare(128, 128)
info = net.plan_info()
vn, _ = layer.node_edge_n.support_values(); ve, _ = layer.node_edge_e.support_values()
p = {k: (v.to(dev) if v is not None else None) for k, v in layer.export_params().items() if not k.startswith('ne_')}
p.update(ne_n_val=vn, ne_e_val=ve)
sp = _lib._spatial_params(p)
ws = torch.zeros(lib.uds_spatial_workspace_floats(net.ptr, S, 64, 128), device=dev)
ox, oe = torch.empty(S, 10000, 128, device=dev), torch.empty(S, 12000, 128, device=dev)
for _ in range(3):
    ws.zero_()
    rc = lib.uds_spatial_layer_forward(net.ptr, ctypes.byref(sp), x.data_ptr(), 128, e.data_ptr(), 128, S, 64, 128, 1, 0, ws.data_ptr(), ox.data_ptr(), oe.data_ptr(), None)
    assert rc == 0, lib.uds_last_error()
torch.cuda.synchronize()
raw = ws[65536:].view(torch.int64).cpu().numpy()
rows = raw[:(len(raw) // 16) * 16].reshape(-1, 16)
rows = rows[(rows[:, 13] & 1) == 1]
print('waves with stamps:', len(rows), 'plan', info)
names = ['setup', 'wait_dma', 'P0 split', 'bar0', 'P1 mlp', 'bar1', 'P1.5 agg', 'bar2', 'P2 hx', 'bar3', 'P3 gather']
tot = rows[:, :11].sum()
for i, n in enumerate(names):
    print('%-10s %5.1f %%   (per wave and chunk: %.0f cycles)' % (n, 100.0 * rows[:, i].sum() / tot, rows[:, i].mean()))
print('cycles per wave and chunk:', rows[:, :11].sum(1).mean())
