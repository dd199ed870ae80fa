"""Whole-emulator forward at headline scale (N=10k, E=12k, d=64, 3+3 spatial layers, 3+3 temporal layers, T=60):
wall time per forward; run under `rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import os, sys, time
from types import SimpleNamespace
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_uds_amd as U

N, E, T = int(os.environ.get('UDS_N', 10000)), int(os.environ.get('UDS_E', 12000)), int(os.environ.get('UDS_T', 60))
edges = U.synthetic_drainage_network(N, E, 0)
g = U.DrainageGraph.from_edges(edges)
args = SimpleNamespace(state_shape=(N, 4), edge_state_shape=(E, 4), seq_in=T, seq_out=T, embed_size=64, hidden_dim=64, kernel_size=3,
                       n_sp_layer=3, n_tp_layer=3, activation='relu', if_flood=3, edge_fusion=True, edges=edges, act=False,
                       graph=g, model_dir='/tmp/x')
dev = torch.device('cuda', 0)
emul = U.Emulator('GAT', True, 'Conv1D', args).to(dev)
X, B, Ex = torch.rand(1, T, N, 5, device=dev), torch.rand(1, T, N, 1, device=dev), torch.rand(1, T, E, 4, device=dev)
for _ in range(2):
    y, ey = emul(X, B, Ex)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 5
for _ in range(n):
    y, ey = emul(X, B, Ex)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print('forward %.2f ms; %d graph-steps -> %.0f graph-steps/s end-to-end (incl. temporal/heads); out %s %s' %
      (dt * 1e3, 6 * T, 6 * T / dt, tuple(y.shape), tuple(ey.shape)))
