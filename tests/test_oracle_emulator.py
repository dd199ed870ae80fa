"""CPU checks of the whole-emulator oracle (oracle/emulator_ref.py): independent re-derivations of its pieces."""
import numpy as np
import torch

from oracle import emulator_ref as OE
from tests.util import emulator_args, emulator_norms


def test_conv1d_causal_matches_torch_conv1d():
    """Keras causal Conv1D = left zero-padding by (k-1)*dilation + cross-correlation: same as torch.conv1d on the padded signal."""
    g = torch.Generator().manual_seed(0)
    for dil in (1, 2, 4):
        x = torch.rand(3, 9, 5, generator=g, dtype=torch.float64)
        k = torch.rand(3, 5, 4, generator=g, dtype=torch.float64) - 0.5
        b = torch.rand(4, generator=g, dtype=torch.float64)
        ref = torch.nn.functional.conv1d(torch.nn.functional.pad(x.transpose(1, 2), (2 * dil, 0)), k.permute(2, 1, 0), b, dilation=dil)
        out = OE.conv1d_causal(x, k, b, dil, 'linear')
        assert torch.allclose(out, ref.transpose(1, 2), atol=1e-13)
        # causality: output at t does not see inputs after t
        x2 = x.clone(); x2[:, 6:] += 1.0
        assert torch.equal(OE.conv1d_causal(x2, k, b, dil, 'relu')[:, :6], OE.conv1d_causal(x, k, b, dil, 'relu')[:, :6])


def test_forward_shapes_channels_and_ranges(networks):
    net = networks['astlingen']
    args = emulator_args(net['edges'], net['n_node'])
    p = OE.init_params(args, seed=1)
    c = OE.config(args)
    assert (c.n_in, c.n_out, c.e_out, c.b_in) == (5, 1, 3, 1)          # if_flood adds the flood bit; edge fusion drops q_in, q_out
    g = torch.Generator().manual_seed(2)
    X, B, E = (torch.rand(2, 5, 30, 5, generator=g, dtype=torch.float64), torch.rand(2, 5, 30, 1, generator=g, dtype=torch.float64),
               torch.rand(2, 5, 29, 4, generator=g, dtype=torch.float64))
    a = torch.rand(2, 5, 2, generator=g, dtype=torch.float64)
    AE = OE.get_edge_action(c, a)
    assert AE.shape == (2, 5, 29, 1)
    idx = OE._act_edge_index(c)
    assert torch.equal(AE[..., idx, 0], a) and abs(float(AE.sum()) - (float(a.sum()) + 2 * 5 * (29 - 2))) < 1e-9
    y, ey = OE.forward(args, p, X, B, E, AE)
    assert y.shape == (2, 5, 30, 2) and ey.shape == (2, 5, 29, 3)
    assert float(y.min()) >= 0 and float(y.max()) <= 1 and float(ey.abs().max()) <= 1     # hard_sigmoid / sigmoid / tanh heads
    # snapshots of different batch items are independent
    y1, _ = OE.forward(args, p, X[:1], B[:1], E[:1], AE[:1])
    assert torch.allclose(y1, y[:1], atol=1e-13)


def test_flow_balance_conserves_volume(networks):
    """post_proc edge fusion (emulator.py:717-724): what leaves a link's from-node enters its to-node."""
    net = networks['shunqing']
    args = emulator_args(net['edges'], net['n_node'], act=False)
    norms = emulator_norms(args)
    for k in ('y', 'e'):
        norms[k][0] = 1.0                                                # unit scales: balance in raw units
    g = torch.Generator().manual_seed(3)
    preds = torch.rand(1, 5, 113, 2, generator=g, dtype=torch.float64)
    ep = torch.rand(1, 5, 131, 3, generator=g, dtype=torch.float64) - 0.5
    y, _ = OE.post_proc(args, norms, preds, ep, None, None)
    assert y.shape == (1, 5, 113, 4)
    q_in, q_out = y[..., 1], y[..., 2]
    assert torch.allclose(q_in.sum(-1), ep[..., -1].abs().sum(-1), atol=1e-12)
    assert torch.allclose(q_out.sum(-1), ep[..., -1].abs().sum(-1), atol=1e-12)


def test_constrain_and_normalize_roundtrip(networks):
    net = networks['astlingen']
    args = emulator_args(net['edges'], net['n_node'])
    norms = emulator_norms(args)
    g = torch.Generator().manual_seed(4)
    v = torch.rand(2, 3, 30, 5, generator=g, dtype=torch.float64)
    assert torch.allclose(OE.normalize(norms, OE.normalize(norms, v, 'y'), 'y', True), v, atol=1e-14)
    y = torch.rand(2, 3, 30, 4, generator=g, dtype=torch.float64) * 3
    r = torch.rand(2, 3, 30, 1, generator=g, dtype=torch.float64)
    q_w, yc = OE.constrain(args, y, r)
    f = y[..., -1] > 0.5
    hmax = torch.as_tensor(args.hmax)
    assert torch.equal(yc[..., 0][f], hmax.expand_as(yc[..., 0])[f])        # flooded nodes sit at hmax
    assert float(q_w[~f].abs().max()) == 0 and float(q_w[..., 0].abs().max()) == 0   # gated by the flood bit; outfall (node 0) never floods


def test_gru_lstm_sequences_match_torch_modules():
    """The GRU / LSTM restatements (TF 2.10 conventions: GRU reset_after=True with a (2, 3H) bias and gate order z, r, h; LSTM
    gate order i, f, c, o) against torch.nn.GRU / LSTM, an independent implementation of the same recurrences (torch's GRU
    also applies the reset gate AFTER the recurrent product; its gate order is r, z, n)."""
    g = torch.Generator().manual_seed(11)
    M, T, F, H = 5, 9, 6, 4
    x = torch.rand(M, T, F, generator=g, dtype=torch.float64) - 0.5
    r = lambda *s: torch.rand(*s, generator=g, dtype=torch.float64) - 0.5
    # GRU
    k, u, b = r(F, 3 * H), r(H, 3 * H), r(2, 3 * H)
    ref = torch.nn.GRU(F, H, batch_first=True).double()
    perm = torch.cat([torch.arange(H, 2 * H), torch.arange(0, H), torch.arange(2 * H, 3 * H)])       # keras (z, r, h) -> torch (r, z, n)
    with torch.no_grad():
        ref.weight_ih_l0.copy_(k[:, perm].T); ref.weight_hh_l0.copy_(u[:, perm].T)
        ref.bias_ih_l0.copy_(b[0, perm]); ref.bias_hh_l0.copy_(b[1, perm])
        want = ref(x)[0]
    assert torch.allclose(OE.gru_sequence(x, k, u, b), want, atol=1e-13)
    # LSTM (torch keeps two bias vectors that are simply added)
    k, u, b = r(F, 4 * H), r(H, 4 * H), r(4 * H)
    ref = torch.nn.LSTM(F, H, batch_first=True).double()
    with torch.no_grad():
        ref.weight_ih_l0.copy_(k.T); ref.weight_hh_l0.copy_(u.T)
        ref.bias_ih_l0.copy_(b); ref.bias_hh_l0.zero_()
        want = ref(x)[0]
    assert torch.allclose(OE.lstm_sequence(x, k, u, b), want, atol=1e-13)
