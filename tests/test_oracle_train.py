"""CPU checks of the training-step oracle (oracle/train_ref.py; "parity unpinned" like the rest of oracle/): the
losses follow the Keras definitions on hand-computable cases, autograd through the restated forward agrees with central
finite differences, and the Adam restatement reproduces a hand-computed first step."""
import json
import os

import numpy as np
import torch

from oracle import emulator_ref as OE
from oracle import train_ref as OT
from tests.util import emulator_args, emulator_norms

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_keras_losses_known_answers():
    y, p = torch.tensor([[[0.0, 1.0], [2.0, 2.0]]]), torch.tensor([[[1.0, 1.0], [0.0, 4.0]]])
    assert float(OT.mse(y, p)) == (0.5 + 4.0) / 2                       # mean over the last axis, then over samples
    assert float(OT.mse(y, p, torch.tensor([2.0, 0.5]))) == (1.0 + 2.0) / 2   # weights scale samples, divisor stays the sample count
    yt, pp = torch.tensor([[1.0], [0.0]]), torch.tensor([[0.8], [0.0]])
    expect = (-np.log(0.8) * 3.0 + -np.log(1 - 1e-7) * 1.0) / 2         # probabilities clipped to [1e-7, 1 - 1e-7]
    assert abs(float(OT.bce(yt, pp, torch.tensor([3.0, 1.0]))) - expect) < 1e-6   # fp32 inputs


def test_adam_first_step_known_answer():
    p = {'w': torch.tensor([1.0, -2.0, 0.5], dtype=torch.float64)}
    g = {'w': torch.tensor([3.0, -4.0, 0.0], dtype=torch.float64)}          # norm 5 -> clipped to (0.6, -0.8, 0)
    OT.Adam(lr=1e-2).step(list(OT.tree_leaves(p)), g)
    gc = np.array([0.6, -0.8, 0.0])
    m, v = 0.1 * gc, 0.001 * gc ** 2
    lr_t = 1e-2 * np.sqrt(1 - 0.999) / (1 - 0.9)
    assert np.allclose(p['w'].numpy(), np.array([1.0, -2.0, 0.5]) - lr_t * m / (np.sqrt(v) + 1e-7), rtol=0, atol=1e-15)


def test_gradients_agree_with_finite_differences():
    with open(os.path.join(ROOT, 'tests', 'golden', 'networks.json')) as fh:
        net = json.load(fh)['astlingen']
    edges, n = np.array(net['edges']), net['n_node']
    args = emulator_args(edges, n, embed_size=8, hidden_dim=8, n_sp_layer=1, n_tp_layer=1, if_flood=1, seq_in=3, seq_out=3)
    norms = emulator_norms(args)
    params = OE.init_params(args, seed=1)
    c = OE.config(args)
    g = torch.Generator().manual_seed(0)
    r = lambda *s: torch.rand(*s, generator=g, dtype=torch.float64)
    x, b, ex = r(1, 3, n, c.n_in), r(1, 3, n, 1) * 0.1, r(1, 3, len(edges), 4)
    a, y, ey = r(1, 3, 2), r(1, 3, n, 5), r(1, 3, len(edges), 3)
    y[..., -2] = (y[..., -2] > 0.7).double()
    ls, gr = OT.grads(args, params, norms, x, a, b, y, ex, ey)
    assert len(ls) == 3 and all(np.isfinite(float(l)) for l in ls)
    leaves = dict(OT.tree_leaves(params))
    total = lambda: float(sum(OT.losses(args, params, norms, x, a, b, y, ex, ey)))
    for name in ('embed_x.kernel', 'block1.0.gat_x.attn_kernel_neighs', 'block1.0.node_edge_n.weight', 'tem2_e.0.kernel', 'e_out.bias'):
        t = leaves[name]
        idx = tuple(int(i) for i in np.unravel_index(int(gr[name].abs().argmax()), t.shape))
        h = 1e-6
        with torch.no_grad():
            old = float(t[idx])
            t[idx] = old + h
            up = total()
            t[idx] = old - h
            dn = total()
            t[idx] = old
        fd = (up - dn) / (2 * h)
        assert abs(fd - float(gr[name][idx])) <= 1e-5 * max(1.0, abs(fd)) + 1e-9, (name, fd, float(gr[name][idx]))
