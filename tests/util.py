"""Shared helpers for the tests: seeded parameters in the oracle's (Keras-shaped) format."""
import numpy as np
import torch


def glorot(gen, shape, dtype):
    if len(shape) == 2:
        fan_in, fan_out = shape
    else:
        rf = int(np.prod(shape[:-2]))
        fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
    lim = (6.0 / (fan_in + fan_out)) ** 0.5
    return ((torch.rand(shape, generator=gen, dtype=torch.float64) * 2 - 1) * lim).to(dtype)


def spatial_params(n_node, n_edge, fx, fe, d, seed=1, dtype=torch.float64, bias_scale=0.1, dense_ne=True,
                   nnz_n=None, nnz_e=None):
    """Random parameters of one spatial layer (`emulator.py:225-230`), Keras shapes.  Biases are
    non-zero (trained-like) so bias handling is exercised."""
    g = torch.Generator().manual_seed(seed)
    h = d // 2
    rn = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64).to(dtype)
    p = {
        'xe_k': glorot(g, (fe, h), dtype), 'xe_b': rn(h) * bias_scale,
        'ex_k': glorot(g, (fx, h), dtype), 'ex_b': rn(h) * bias_scale,
        'gx_k': glorot(g, (fx + h, 1, d), dtype), 'gx_as': glorot(g, (d, 1, 1), dtype),
        'gx_an': glorot(g, (d, 1, 1), dtype), 'gx_b': rn(d) * bias_scale,
        'ge_k': glorot(g, (fe + h, 1, d), dtype), 'ge_as': glorot(g, (d, 1, 1), dtype),
        'ge_an': glorot(g, (d, 1, 1), dtype), 'ge_b': rn(d) * bias_scale,
    }
    if dense_ne:
        p.update({'ne_n_w': rn(n_node, n_edge) * 0.05, 'ne_n_b': torch.zeros(n_node, n_edge, dtype=dtype),
                  'ne_e_w': rn(n_edge, n_node) * 0.05, 'ne_e_b': torch.zeros(n_edge, n_node, dtype=dtype)})
    else:
        p.update({'ne_n_v': rn(nnz_n) * 0.05 + 0.3, 'ne_e_v': rn(nnz_e) * 0.05 + 0.3})
    return p


def cast(p, dtype):
    return {k: (v.to(dtype) if isinstance(v, torch.Tensor) else v) for k, v in p.items()}


def load_spatial_layer(layer, p, device):
    """Copy oracle-format parameters into a gnn_uds_amd.layers.SpatialLayer."""
    f32 = lambda t: t.to(torch.float32).to(device).contiguous()
    layer.to(device)
    layer.dense_xe.kernel.data = f32(p['xe_k']); layer.dense_xe.bias.data = f32(p['xe_b'])
    layer.dense_ex.kernel.data = f32(p['ex_k']); layer.dense_ex.bias.data = f32(p['ex_b'])
    if 'ne_n_w' in p:
        layer.node_edge_n.weight.data = f32(p['ne_n_w']); layer.node_edge_n.bias.data = f32(p['ne_n_b'])
        layer.node_edge_e.weight.data = f32(p['ne_e_w']); layer.node_edge_e.bias.data = f32(p['ne_e_b'])
    else:
        layer.node_edge_n.weight.data = f32(p['ne_n_v']); layer.node_edge_n.bias.data = torch.zeros_like(layer.node_edge_n.weight.data)
        layer.node_edge_e.weight.data = f32(p['ne_e_v']); layer.node_edge_e.bias.data = torch.zeros_like(layer.node_edge_e.weight.data)
    layer.gat_x.kernel.data = f32(p['gx_k']); layer.gat_x.attn_kernel_self.data = f32(p['gx_as'])
    layer.gat_x.attn_kernel_neighs.data = f32(p['gx_an']); layer.gat_x.bias.data = f32(p['gx_b'])
    layer.gat_e.kernel.data = f32(p['ge_k']); layer.gat_e.attn_kernel_self.data = f32(p['ge_as'])
    layer.gat_e.attn_kernel_neighs.data = f32(p['ge_an']); layer.gat_e.bias.data = f32(p['ge_b'])
    return layer
