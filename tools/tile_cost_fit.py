"""Calibration of tile_plan.hpp: tile_cost().  Needs a library built with UDS_PHASE_TIMING=1; run with UDS_CHUNK=60 so that workgroup = one tile x all 60 snapshots.  Writes gpurun_out/tile_costs.csv (one row per tile:
side, n_own, n_prim, n_sec, n_inc, n_adj, cycles per snapshot, setup cycles) and prints a least-squares fit."""
import os, sys, ctypes
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_uds_amd as U
from gnn_uds_amd import _lib

assert os.environ.get('UDS_CHUNK') == '60'      # (a library built with UDS_DEFINES='-DUDS_KNOBS -DUDS_PHASE_TIMING')
dev = torch.device('cuda', 0)
g = U.DrainageGraph.from_edges(U.synthetic_drainage_network(10000, 12000, 0))
S = 60
layer = U.SpatialLayer(g, 64, 'relu', sparse_params=True, generator=torch.Generator().manual_seed(1)).to(dev)
x, e = torch.rand(S, 10000, 64, device=dev), torch.rand(S, 12000, 64, device=dev)
lib = _lib.load()
net = layer.network()
net.prepare(64, 64)
vn, _ = layer.node_edge_n.support_values(); ve, _ = layer.node_edge_e.support_values()
p = {k: (v.to(dev) if v is not None else None) for k, v in layer.export_params().items() if not k.startswith('ne_')}
p.update(ne_n_val=vn, ne_e_val=ve)
sp = _lib._spatial_params(p)
ws = torch.zeros(lib.uds_spatial_workspace_floats(net.ptr, S, 32, 64), device=dev)
ox, oe = torch.empty(S, 10000, 64, device=dev), torch.empty(S, 12000, 64, device=dev)
for _ in range(3):
    ws.zero_()
    rc = lib.uds_spatial_layer_forward(net.ptr, ctypes.byref(sp), x.data_ptr(), 64, e.data_ptr(), 64, S, 32, 64, 1, 0, ws.data_ptr(), ox.data_ptr(), oe.data_ptr(), None)
    assert rc == 0
torch.cuda.synchronize()
hdr, pool, caps = _lib.tile_plan(g, 128, 128, 128, 208)
T = hdr.shape[0]
raw = ws[65536:].view(torch.int64).cpu().numpy()
rows = raw[:T * 8 * 16].reshape(T, 8, 16)
q8, r8 = T // 8, T % 8
out = []
for b in range(T):
    xcd = b % 8
    w = (xcd * (q8 + 1) if xcd < r8 else r8 * (q8 + 1) + (xcd - r8) * q8) + b // 8
    r = rows[b]
    st = int(r[0, 7])
    assert st & 1, b
    side, n_own, n_prim, n_sec = (st >> 8) & 0xff, (st >> 16) & 0xffff, (st >> 32) & 0xffff, (st >> 48) & 0xffff
    assert (n_own, n_prim, n_sec, side) == (hdr[w, 0], hdr[w, 1], hdr[w, 2], hdr[w, 6]), (b, w, n_own, n_prim, n_sec, hdr[w])
    per_wave = r[:, 1:7].sum(1)
    out.append((side, n_own, n_prim, n_sec, hdr[w, 3], hdr[w, 4], per_wave.max() / S, r[:, 0].max()))
out = np.array(out, dtype=np.float64)
os.makedirs('gpurun_out', exist_ok=True)
np.savetxt('gpurun_out/tile_costs.csv', out, fmt='%.1f', delimiter=',', header='side,n_own,n_prim,n_sec,n_inc,n_adj,cycles_per_snapshot,setup_cycles')
y = out[:, 6]
A = np.stack([np.ones(T), np.ceil(out[:, 3] / 16), np.ceil(out[:, 2] / 16), out[:, 1], out[:, 5], out[:, 4]], 1)
coef, res, *_ = np.linalg.lstsq(A, y, rcond=None)
print('setup split (mean cycles): block fetch + barrier %.0f | rest of setup %.0f (of which issue of gathers + weight loads %.0f)' % (rows[:, :, 13].max(1).mean(), rows[:, :, 0].max(1).mean(), rows[:, :, 14].max(1).mean()))
print('first stage wait (stamp 1 / S) %.0f' % (rows[:, :, 1].max(1).mean() / S))
print('tiles', T, 'cycles/snapshot min %.0f mean %.0f max %.0f; setup mean %.0f' % (y.min(), y.mean(), y.max(), out[:, 7].mean()))
print('fit: const %.1f + %.2f*sec_blocks + %.2f*prim_blocks + %.3f*n_own + %.4f*n_adj + %.4f*n_inc; rms resid %.1f' % (*coef, np.sqrt(((A @ coef - y) ** 2).mean())))
A2 = A[:, :4]
c2, *_ = np.linalg.lstsq(A2, y, rcond=None)
print('fit4: const %.1f + %.2f*sec_blocks + %.2f*prim_blocks + %.3f*n_own; rms resid %.1f' % (*c2, np.sqrt(((A2 @ c2 - y) ** 2).mean())))
for side in (0, 1):
    m = out[:, 0] == side
    print('side', side, 'mean cycles', y[m].mean(), 'model(old)', (14 + 0.85 * A[m, 1] + 4 * A[m, 2] + 29 * A[m, 3] / 128).mean())
