"""Whole-Emulator forward at the reference's DEFAULT sizes (utils/config.yaml: embed_size 128, hidden_dim 64, 2 + 2 spatial
layers, 2 + 2 temporal layers, conv GAT) on the headline network, for each temporal net: Conv1D (75 of the 86 shipped model
configurations), GRU (the argparse default), LSTM."""
import json
import os
import sys
import time
from types import SimpleNamespace

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_uds_amd as U


def main():
    dev = torch.device('cuda:0')
    N, E, T = int(os.environ.get('NODES', 10000)), int(os.environ.get('LINKS', 12000)), int(os.environ.get('T', 60))
    g = U.DrainageGraph.from_edges(U.synthetic_drainage_network(N, E, 0))
    a = SimpleNamespace(state_shape=(N, 4), edge_state_shape=(E, 4), seq_in=T, seq_out=T, embed_size=128, hidden_dim=64, kernel_size=3,
                        n_sp_layer=2, n_tp_layer=2, activation='relu', if_flood=3, edge_fusion=True, edges=g.edges, act=False, graph=g,
                        model_dir=None)
    X, B, Ex = torch.rand(1, T, N, 5, device=dev), torch.rand(1, T, N, 1, device=dev), torch.rand(1, T, E, 4, device=dev)
    out = {}
    trained = bool(int(os.environ.get('TRAINED', '0')))        # TRAINED=1: dense NodeEdge parameters with a bias trained off the support
    if trained:
        a.sparse_params = False
    for rec in ('Conv1D', 'GRU', 'LSTM')[:1 if trained else 3]:
        emul = U.Emulator('GAT', True, rec, a, generator=torch.Generator().manual_seed(1)).to(dev)
        if trained:
            with torch.no_grad():
                for blk in (emul.block1, emul.block2):
                    for ly in blk.layers:
                        ly.node_edge_n.bias.normal_(0.0, 0.01)
                        ly.node_edge_e.bias.normal_(0.0, 0.01)
        for _ in range(2):
            emul(X, B, Ex)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            emul(X, B, Ex)
        torch.cuda.synchronize()
        out['ms_per_forward_' + rec + ('_trained_bias' if trained else '')] = (time.perf_counter() - t0) / 5 * 1e3
        if trained:
            out['paths'] = [ly.last_path for ly in emul.block1.layers] + [ly.last_path for ly in emul.block2.layers]
    print(json.dumps(out))


if __name__ == '__main__':
    main()
