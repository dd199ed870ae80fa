"""Committed golden forward vectors (tests/golden/forward_c1.npz, made by make_forward_fixtures.py from the fp64
dense-masked oracle): the sparse oracle must reproduce them on CPU, the HIP path on the GPU."""
import os

import numpy as np
import pytest
import torch

import gnn_uds_amd as U
from oracle import sparse_csr as OS
from oracle import spektral_dense as OD

HERE = os.path.dirname(os.path.abspath(__file__))


def load(tag):
    z = np.load(os.path.join(HERE, 'golden', 'forward_c1.npz'))
    t = lambda k: torch.from_numpy(z['%s_%s' % (tag, k)])
    p = {k[len(tag) + 3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(tag + '_p_')}
    return z[tag + '_edges'], p, t('x'), t('e'), t('out_x'), t('out_e')


@pytest.mark.parametrize('tag', ['c1', 'd64'])
def test_oracles_reproduce_golden(tag):
    edges, p, x, e, gx, ge = load(tag)
    g = U.DrainageGraph.from_edges(edges, 50)
    ne = torch.from_numpy(g.inc_n.to_dense())
    dx, de = OD.spatial_layer_dense(x, e, p, torch.from_numpy(g.adj.to_dense()), torch.from_numpy(g.edge_adj.to_dense()), ne)
    assert torch.allclose(dx, gx, atol=1e-13) and torch.allclose(de, ge, atol=1e-13)
    sx, se = OS.spatial_layer_csr(x, e, p, (g.adj.rowptr, g.adj.col), (g.edge_adj.rowptr, g.edge_adj.col), node_edge=ne)
    assert torch.allclose(sx, gx, atol=1e-12) and torch.allclose(se, ge, atol=1e-12)
    assert gx.shape == x.shape and float(gx.min()) >= 0       # relu outputs, same layout


@pytest.mark.gpu
@pytest.mark.parametrize('tag,precision,tol', [('c1', 'bf16x3', 1e-5), ('d64', 'fp32', 5e-6), ('d64', 'bf16x3', 1e-5)])
def test_hip_reproduces_golden(tag, precision, tol):
    from tests.util import load_spatial_layer
    dev = torch.device('cuda', 0)
    edges, p, x, e, gx, ge = load(tag)
    g = U.DrainageGraph.from_edges(edges, 50)
    layer = load_spatial_layer(U.SpatialLayer(g, x.shape[-1], 'relu', sparse_params=False, precision=precision), p, dev)
    ox, oe = layer(x.float().to(dev), e.float().to(dev))
    for o, r in ((ox, gx), (oe, ge)):
        assert float((o.double().cpu() - r).abs().max()) <= tol * max(1.0, float(r.abs().max()))
