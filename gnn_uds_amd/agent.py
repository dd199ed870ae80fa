"""The RL agents' graph encoder on the HIP engine: `ConvNet` of the reference
(`surrogate/agent.py:20-99`) -- Dense embeddings, the SAME spatial block as the emulator (fusion MLPs, NodeEdge, GAT on the
node graph and the line graph; or one conv over the combined graph for `graph_base`), then Spektral's
`GlobalAttnSumPool` over the stacked node + link rows.  One snapshot per sample, the batch is large: the snapshots of
the fused kernel are the batch elements.

    ConvNet(args, conv)(X, E[, B]) -> (batch, conv_dim)

Weight names follow the Keras layers (`embed_x`, `embed_e`, `block.layers.i.*`, `pool.attn_kernel` (F, 1)).
"""
import numpy as np
import torch
from torch import nn

from . import _lib
from . import autograd as _ag
from .graph import DrainageGraph, csr_from_dense
from .layers import Dense, GraphBaseBlock, SpatialBlock, _glorot_uniform, _param


class GlobalAttnSumPool(nn.Module):
    """spektral.layers.GlobalAttnSumPool in batch mode: alpha = softmax_n(x @ attn_kernel), out = sum_n alpha_n x_n."""

    def __init__(self, in_features, generator=None):
        super().__init__()
        self.attn_kernel = _param(_glorot_uniform((int(in_features), 1), 'cpu', generator))

    def forward(self, x):
        if not x.is_cuda:
            raise _lib.UdsError('GlobalAttnSumPool input is on %s: gnn_uds_amd runs on the MI355X only' % x.device)
        F = x.shape[-1]
        if x.dim() == 3 and F >= 4 and F <= 256 and (F & (F - 1)) == 0 and not _ag.grad_on(x, self.attn_kernel):
            return _lib.attn_sum_pool(x.contiguous(), self.attn_kernel)                   # one HIP launch, one pass over the rows
        alpha = torch.softmax(torch.matmul(x, self.attn_kernel).squeeze(-1), dim=-1)      # (B, N): differentiable form (RL training)
        return torch.matmul(alpha.unsqueeze(-2), x).squeeze(-2)                          # (B, F)


class ConvNet(nn.Module):
    def __init__(self, args, conv='GAT', precision='bf16x3', generator=None):
        super().__init__()
        g = lambda k, d=None: getattr(args, k, d)
        self.conv_dim, self.n_sp_layer = int(g('conv_dim', 128)), int(g('n_sp_layer', 3))
        self.n_node, self.n_in = g('state_shape', (40, 4))
        if g('if_flood', False):
            self.n_in += 1
        self.use_pred = bool(g('use_pred', False))
        self.b_in = (2 if g('tide', False) else 1) if self.use_pred else 0
        self.graph_base = int(g('graph_base', 0))
        self.n_edge, self.e_in = g('edge_state_shape', (40, 3))
        self.activation = g('activation', None) or 'linear'
        kind = 'GAT' if 'GAT' in conv else ('GCN' if 'GCN' in conv else None)
        if kind is None:
            raise NotImplementedError('conv=%r is not built (GAT and GCN are)' % (conv,))
        d, a, gen = self.conv_dim, self.activation, generator
        self.embed_x = Dense(d, a, in_features=self.n_in + self.b_in, generator=gen)       # agent.py:78
        self.embed_e = Dense(d, a, in_features=self.e_in, generator=gen)                   # agent.py:79
        graph = g('graph')
        if self.graph_base:
            adj = np.asarray(g('adj'))
            filt = csr_from_dense((adj > 0).astype(int), add_self_loops=True) if kind == 'GAT' else None
            if kind == 'GCN':
                from .layers import GCNConv
                filt = GCNConv.preprocess(adj)
            self.block = GraphBaseBlock(self.n_node, self.n_edge, filt, d, self.n_sp_layer, a, generator=gen, conv=kind, precision=precision)
        else:
            if isinstance(graph, DrainageGraph):
                filters = (None, None)
            else:
                adj, edge_adj = np.asarray(g('adj', np.eye(self.n_node))), np.asarray(g('edge_adj', np.eye(self.n_edge)))
                graph = DrainageGraph.from_dense(adj, edge_adj, np.asarray(g('node_edge'), dtype=np.float64), g('edges'))
                if kind == 'GCN':
                    from .layers import GCNConv
                    filters = (GCNConv.preprocess(adj), GCNConv.preprocess(edge_adj))
                else:
                    filters = (None, None)
            self.block = SpatialBlock(graph, d, self.n_sp_layer, a, generator=gen, precision=precision, conv=kind, filters=filters)
        self.pool = GlobalAttnSumPool(d, generator=gen)

    def forward(self, X, E, B=None):
        """X (batch, N, n_in), E (batch, E, e_in)[, B (batch, N, b_in) when use_pred] -> (batch, conv_dim)."""
        if self.use_pred:
            if B is None:
                raise ValueError('use_pred needs the boundary input B')
            X = torch.cat([X, B], dim=-1)
        x, e = self.embed_x(X.contiguous()), self.embed_e(E.contiguous())
        x, e = self.block(x, e)
        return self.pool(torch.cat([x, e], dim=-2))
