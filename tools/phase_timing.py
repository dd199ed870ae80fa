"""Diagnostic: per-phase cycle shares of the fused kernel (needs a library built with UDS_PHASE_TIMING=1).
Run on the GPU box:  UDS_PHASE_TIMING=1 python -m gnn_uds_amd.build --force && python tools/phase_timing.py"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_uds_amd as U
from gnn_uds_amd import _lib

dev = torch.device('cuda', 0)
g = U.DrainageGraph.from_edges(U.synthetic_drainage_network(10000, 12000, 0))
S = int(sys.argv[1]) if len(sys.argv) > 1 else 60
FX, FE = int(os.environ.get('FX', 64)), int(os.environ.get('FE', 64))
layer = U.SpatialLayer(g, 64, 'relu', fx=FX, fe=FE, sparse_params=True, generator=torch.Generator().manual_seed(1)).to(dev)
x, e = torch.rand(S, 10000, FX, device=dev), torch.rand(S, 12000, FE, device=dev)
lib = _lib.load()
net = layer.network()
net.prepare(FX, FE)
info = net.plan_info()
vn, _ = layer.node_edge_n.support_values(); ve, _ = layer.node_edge_e.support_values()
p = {k: (v.to(dev) if v is not None else None) for k, v in layer.export_params().items() if not k.startswith('ne_')}
p.update(ne_n_val=vn, ne_e_val=ve)
import ctypes
sp = _lib._spatial_params(p)
ws = torch.zeros(lib.uds_spatial_workspace_floats(net.ptr, S, 32, 64), device=dev)
ox, oe = torch.empty(S, 10000, 64, device=dev), torch.empty(S, 12000, 64, device=dev)
for _ in range(3):
    ws.zero_()
    rc = lib.uds_spatial_layer_forward(net.ptr, ctypes.byref(sp), x.data_ptr(), FX, e.data_ptr(), FE, S, 32, 64, 1, 0, ws.data_ptr(), ox.data_ptr(), oe.data_ptr(), None)
    assert rc == 0
torch.cuda.synchronize()
n_tiles = info['node_tiles'] + info['link_tiles']
raw = ws[65536:].view(torch.int64).cpu().numpy()
n_wg = 0
rows = raw[:(len(raw) // 16) * 16].reshape(-1, 16)
valid = (rows[:, 7] & 1) == 1
rows = rows[valid]
print('waves with stamps:', len(rows), 'plan', info)
names = os.environ.get('PHASE_NAMES', 'setup,wait_stage,P1,bar1,P2,bar2,P3').split(',')
tot = rows[:, :7].sum()
for side in (0, 1):
    r = rows[((rows[:, 7] >> 8) & 0xff) == side]
    print('side', side, 'waves', len(r), ' mean cycles per wave:', {n: int(r[:, k].mean()) for k, n in enumerate(names)}, 'total', int(r[:, :7].sum(1).mean()))
for wv in range(8):
    r = rows[rows[:, 12] == wv]
    print('wave', wv, {n: int(r[:, k].mean()) for k, n in enumerate(names)}, 'P2a..d', [int(r[:, k].mean()) for k in (8, 9, 10, 11)])
rt, mt = (rows[:, 15] >> 32).astype(np.float64), (rows[:, 15] & 0xffffffff).astype(np.float64)
ok = rt > 0
print('in-kernel clock: median %.3f GHz (shader cycles / 100 MHz real-time ticks per workgroup), workgroup life %.1f us median'
      % (float(np.median(mt[ok] / rt[ok])) * 0.1, float(np.median(rt[ok])) * 0.01))
print('shares:', {n: round(float(rows[:, k].sum() / tot), 3) for k, n in enumerate(names)})
# workgroup life by side and size (wave 0 of every workgroup): how uneven the (tile, chunk) items are
w0 = rows[rows[:, 12] == 0]
life = (w0[:, 15] >> 32).astype(np.float64) * 0.01
sd, n_own = (w0[:, 7] >> 8) & 0xff, (w0[:, 7] >> 16) & 0xffff
for side in (0, 1):
    l = life[(sd == side) & (life > 0)]
    if len(l):
        print('side %d: %d workgroups, life us min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f' % (side, len(l), l.min(), np.percentile(l, 10), np.median(l), np.percentile(l, 90), l.max()))
print('sum of workgroup lives / 256 CUs = %.1f us (the launch if perfectly packed)' % (life.sum() / 256))
