"""Whole-emulator forward at the reference's DEFAULT model size (utils/config.yaml: embed_size 128, hidden_dim 64,
n_sp_layer 2, n_tp_layer 2) on the headline network, T_in = T_out = 60."""
import os, sys, time
from types import SimpleNamespace
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_uds_amd as U

N, E, T = 10000, 12000, 60
edges = U.synthetic_drainage_network(N, E, 0)
g = U.DrainageGraph.from_edges(edges)
act = bool(int(os.environ.get('UDS_ACT', 0)))
args = SimpleNamespace(state_shape=(N, 4), edge_state_shape=(E, 4), seq_in=T, seq_out=T, embed_size=128, hidden_dim=64, kernel_size=3,
                       n_sp_layer=2, n_tp_layer=2, activation='relu', if_flood=3, edge_fusion=True, edges=edges, act=act,
                       act_edges=edges[[1, 4]], graph=g, model_dir=None)
dev = torch.device('cuda', 0)
emul = U.Emulator('GAT', True, 'Conv1D', args).to(dev)
X, B, Ex = torch.rand(1, T, N, 5, device=dev), torch.rand(1, T, N, 1, device=dev), torch.rand(1, T, E, 4, device=dev)
AE = torch.rand(1, T, E, 1, device=dev) if act else None
for _ in range(2):
    y, ey = emul(X, B, Ex, AE)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 5
for _ in range(n):
    y, ey = emul(X, B, Ex, AE)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print('d=128 forward %.2f ms (act=%s); out %s %s' % (dt * 1e3, act, tuple(y.shape), tuple(ey.shape)))
