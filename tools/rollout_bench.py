"""Autoregressive rollout at BASELINE config 2 scale (N=2000, E=2500, d=64, 3+3 spatial layers, seq_in=6, seq_out=1,
100 steps): time per simulated step through Emulator._model (roll = 100)."""
import os, sys, time
from types import SimpleNamespace
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_uds_amd as U

N, E, ROLL = int(os.environ.get('UDS_N', 2000)), int(os.environ.get('UDS_E', 2500)), int(os.environ.get('UDS_ROLL', 100))
B = int(os.environ.get('UDS_B', 1))
edges = U.synthetic_drainage_network(N, E, 0)
g = U.DrainageGraph.from_edges(edges)
args = SimpleNamespace(state_shape=(N, 4), edge_state_shape=(E, 4), seq_in=6, seq_out=1, embed_size=64, hidden_dim=64, kernel_size=3,
                       n_sp_layer=3, n_tp_layer=2, activation='relu', if_flood=3, edge_fusion=True, edges=edges, act=False,
                       graph=g, roll=ROLL, model_dir=None)
dev = torch.device('cuda', 0)
emul = U.Emulator('GAT', True, 'Conv1D', args).to(dev)
rng = np.random.default_rng(0)
emul.set_norm(*[np.stack([0.5 + rng.random((n, c)), np.zeros((n, c))]) for n, c in ((N, 5), (N, 1), (N, 5), (N, 1), (E, 4))])
x = torch.rand(B, 6, N, 5, device=dev)
b = torch.rand(B, ROLL, N, 1, device=dev) * 0.1
ex = torch.rand(B, 6, E, 4, device=dev)
graphed = bool(int(os.environ.get('UDS_GRAPH', 0)))
run = (lambda: emul._model(x, None, b, ex)) if not graphed else (lambda: emul.rollout_graphed(x, None, b, ex))
for _ in range(2):
    y, ey = run()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 3
for _ in range(n):
    y, ey = run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print('rollout of %d steps (B=%d, N=%d): %.1f ms = %.3f ms per step = %.0f simulated steps/s; graphed=%s; out %s %s' %
      (ROLL, B, N, dt * 1e3, dt / ROLL * 1e3, ROLL * B / dt, graphed, tuple(y.shape), tuple(ey.shape)))
