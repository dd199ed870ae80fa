"""GPU tests of training-time dropout (uds_dropout, layers.Dropout, Emulator(dropout=...); reference: keras Dropout layers of
`surrogate/emulator.py:199-213,234-235,287-288,314-318`, active under `self.model(inp, training=fit)`, :411,434).

* the kernel's mask and values are BIT-EXACT against the Philox4x32-10 restatement (oracle/dropout_ref.py, itself pinned by the
  generator's published known-answer vectors in tests/test_dropout.py): aligned and unaligned offsets, ragged sizes, in place;
* statistical parity with keras Dropout: keep fraction within 5 sigma of 1 - rate, expectation preserved, identity at inference;
* the backward pass applies the same mask (recomputed from (seed, offset));
* Spektral's attention dropout (GATConv in training mode: dropout on the softmax coefficients, rate 0.5 -- what `training=True` also
  switches on in the reference): layer values and gradients against the dense restatement fed the same mask;
* the gradients of every Emulator parameter under dropout equal torch autograd over the fp64 oracle fed the same mask stream
  (oracle.emulator_ref.DROPOUT and oracle.spektral_dense.ATTN_DROPOUT hooks) within the whole-model tolerance of tests/test_gpu_train.py."""
import numpy as np
import pytest
import torch

import gnn_uds_amd as U
from gnn_uds_amd import _lib
from gnn_uds_amd.layers import Dropout, DropoutStream
from oracle import dropout_ref as DR
from oracle import emulator_ref as OE
from oracle import train_ref as OT
from tests.test_gpu_train import GRAD_TOL, _problem
from tests.util import close, emulator_param_pairs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    _lib.load()
    return torch.device('cuda', 0)


@pytest.fixture(scope='module')
def networks():
    import json, os
    with open(os.path.join(os.path.dirname(__file__), 'golden', 'networks.json')) as fh:
        return json.load(fh)


@pytest.mark.parametrize('n', [1, 5, 64, 1000, 4097])
@pytest.mark.parametrize('offset', [0, 4, 6, (1 << 34) + 3])
@pytest.mark.parametrize('rate', [0.0, 0.2, 0.5])
def test_kernel_mask_is_bit_exact(dev, n, offset, rate):
    g = torch.Generator().manual_seed(n)
    x = torch.randn(n, generator=g)
    seed = 0x1234567890ABCDEF
    got = _lib.dropout(x.to(dev), rate, seed, offset).cpu()
    scale = np.float32(1.0) / (np.float32(1.0) - np.float32(rate))
    ref = np.where(DR.dropout_mask(n, rate, seed, offset), x.numpy() * scale, np.float32(0.0)).astype(np.float32)
    assert np.array_equal(got.numpy(), ref)


def test_unaligned_tensor_takes_the_scalar_path(dev):
    base = torch.randn(1001, device=dev)
    x = base[1:]                                 # 4-byte aligned only; contiguous
    got = _lib.dropout(x, 0.3, 5, 8).cpu().numpy()
    ref = np.where(DR.dropout_mask(1000, 0.3, 5, 8), x.cpu().numpy() * (np.float32(1) / (np.float32(1) - np.float32(0.3))), np.float32(0)).astype(np.float32)
    assert np.array_equal(got, ref)


def test_statistics_match_keras_dropout(dev):
    n = 1 << 22
    x = torch.full((n,), 2.0, device=dev)
    for rate in (0.1, 0.2, 0.5):
        y = _lib.dropout(x, rate, 77, 0)
        keep = float((y != 0).float().mean())
        assert abs(keep - (1 - rate)) < 5 * (rate * (1 - rate) / n) ** 0.5
        assert abs(float(y.mean()) - 2.0) < 0.01                              # inverted dropout: the expectation is the input
        assert torch.equal(y[y != 0], torch.full_like(y[y != 0], float(np.float32(2.0) * (np.float32(1) / (np.float32(1) - np.float32(rate))))))
    with pytest.raises(_lib.UdsError):
        _lib.dropout(x[:8], 1.0, 1, 0)


def test_module_is_identity_at_inference_and_draws_disjoint_masks_in_training(dev):
    st = DropoutStream(seed=11)
    d = Dropout(0.2, st)
    x = torch.randn(3, 50, 7, device=dev)
    assert d(x) is x and d(x, training=False) is x
    y1, y2 = d(x, True), d(x, True)
    assert st.offset == 2 * ((x.numel() + 3) // 4 * 4)
    assert not torch.equal(y1 == 0, y2 == 0)
    st.reseed(11)
    assert torch.equal(d(x, True), y1)                                        # same (seed, offset): same mask


def test_backward_applies_the_same_mask(dev):
    st = DropoutStream(seed=3)
    d = Dropout(0.5, st)
    x = torch.randn(1000, device=dev, requires_grad=True)
    y = d(x, True)
    gy = torch.randn(1000, device=dev)
    y.backward(gy)
    keep = (y != 0) | (x.detach() == 0)
    assert torch.equal(x.grad, torch.where(keep, gy * 2.0, torch.zeros_like(gy)))


class _RefStream:
    """The mask stream of layers.DropoutStream, applied to the oracle's fp64 tensors in logical (row-major) element order."""

    def __init__(self, seed):
        self.seed, self.offset = seed, 0

    def _mask(self, n, rate):
        m = DR.dropout_mask(n, rate, self.seed, self.offset)
        self.offset += (n + 3) // 4 * 4
        return m.astype(np.float64) * float(np.float32(1.0) / (np.float32(1.0) - np.float32(rate)))

    def __call__(self, t, rate):
        return t * torch.from_numpy(self._mask(t.numel(), rate).reshape(tuple(t.shape)))

    def attn(self, coef, a_hat, rate=0.5):
        """Spektral's attention dropout on the dense coefficients (S, N, 1, N): the engine draws one mask entry per snapshot and
        entry of the CSR pattern (rows in order, columns ascending = the row-major order of the non-zeros of a_hat)."""
        S = int(np.prod(coef.shape[:-3])) if coef.dim() > 3 else 1
        rows, cols = np.nonzero(a_hat.numpy() != 0)
        m = self._mask(S * len(rows), rate).reshape(S, len(rows))
        dense = np.zeros((S,) + tuple(a_hat.shape))
        dense[:, rows, cols] = m
        return coef * torch.from_numpy(dense).reshape(tuple(coef.shape[:-3]) + (a_hat.shape[0], 1, a_hat.shape[1]))


def test_attention_dropout_layer_matches_the_dense_restatement(dev, networks):
    """GATConv in Keras' training mode (Spektral: dropout on the softmax coefficients, rate 0.5) against the dense restatement fed
    the same mask: forward values and the gradients of input, kernel, both attention kernels and the bias."""
    from oracle import spektral_dense as OD
    net = networks['shunqing']
    gph = U.DrainageGraph.from_edges(np.array(net['edges']), net['n_node'])
    g = torch.Generator().manual_seed(4)
    S, F, C = 3, 24, 16
    x = torch.randn(S, gph.n_node, F, generator=g, dtype=torch.float64)
    layer = U.GATConv(C, activation='tanh', in_channels=F, generator=g).to(dev)
    with torch.no_grad():
        layer.bias.normal_(0.0, 0.1, generator=None)
    st = DropoutStream(seed=99)
    xd = x.float().to(dev).requires_grad_(True)
    layer.requires_grad_(True)
    a_dense = gph.adj.to_dense()
    out = layer([xd, a_dense], attn_dropout=st)
    gy = torch.randn(out.shape, generator=g, dtype=torch.float64)
    (out * gy.float().to(dev)).sum().backward()
    ref_p = [p.detach().double().cpu().requires_grad_(True) for p in (layer.kernel, layer.attn_kernel_self, layer.attn_kernel_neighs, layer.bias)]
    xr = x.clone().requires_grad_(True)
    rs = _RefStream(99)
    OD.ATTN_DROPOUT = rs.attn
    try:
        ref = OD.gat_conv_dense(xr, torch.from_numpy(a_dense), *ref_p, 'tanh')
    finally:
        OD.ATTN_DROPOUT = None
    (ref * gy).sum().backward()
    assert st.offset == rs.offset
    close(out.detach(), ref.detach(), 2e-5)
    plain = layer([xd.detach(), a_dense])
    assert float((plain - out.detach()).abs().max()) > 1e-2          # the mask does something
    for got, want in [(xd.grad, xr.grad)] + [(p.grad, r.grad) for p, r in zip((layer.kernel, layer.attn_kernel_self, layer.attn_kernel_neighs, layer.bias), ref_p)]:
        err, scale = float((got.double().cpu() - want).abs().max()), float(want.abs().max())
        assert err <= 1e-4 * scale + 1e-7, (err, scale)


@pytest.mark.parametrize('name,over', [('astlingen', dict(dropout=0.1)),
                                       ('astlingen', dict(dropout=0.25, recurrent='GRU', embed_size=64, hidden_dim=64)),
                                       ('astlingen', dict(dropout=0.1, conv='False', seq_in=5, seq_out=5, n_sp_layer=2)),
                                       ('astlingen', dict(dropout=0.1, graph_base=1, n_sp_layer=1))])
def test_emulator_gradients_under_dropout(dev, networks, name, over):
    args, norms, params, emul, cpu_in, dev_in = _problem(networks, name, dev, **over)
    assert emul.dropout and emul.dropout_stream is not None
    x, a, b, y, ex, ey = cpu_in
    from oracle import spektral_dense as OD
    emul.dropout_stream.reseed(2024)
    OE.DROPOUT = _RefStream(2024)
    OD.ATTN_DROPOUT = OE.DROPOUT.attn          # training=True reaches the GATConv layers too (same stream, the engine's order)
    try:
        ref_losses, ref_grads = OT.grads(args, params, norms, x, a, b, y, ex, ey)
    finally:
        OE.DROPOUT = OD.ATTN_DROPOUT = None
    emul.requires_grad_(True)
    xd, ad, bd, yd, exd, eyd = dev_in
    ae = emul.get_edge_action(ad, True) if emul.act else None
    preds, edge_preds = emul._model(xd, ad, bd, exd, ae, None, True)
    lw = emul._loss_setup(dev)
    ls = [emul.get_node_loss(yd, bd, preds)] + ([emul.get_flood_loss(yd, preds)] if emul.if_flood else []) + [emul._mse(eyd, edge_preds, lw['ewei'])]
    for got, ref in zip(ls, ref_losses):
        close(got, ref, 2e-5)
    sum(ls).backward()
    gmax = max(float(t.abs().max()) for t in ref_grads.values())
    n_checked = 0
    for pname, p, ref in emulator_param_pairs(emul, ref_grads):
        got = p.grad.detach().double().cpu() if p.grad is not None else torch.zeros_like(ref)
        ref = ref.reshape(got.shape)
        err, scale = float((got - ref).abs().max()), float(ref.abs().max())
        assert err <= GRAD_TOL[args.conv] * scale + 1e-7 * gmax, '%s: grad err %.3e vs max|grad| %.3e' % (pname, err, scale)
        n_checked += 1
    assert n_checked == len(list(emul.parameters()))
    # inference ignores the Dropout layers: equal to the same model built without dropout
    with torch.no_grad():
        p0, e0 = emul._model(xd, ad, bd, exd, ae, None, False)
        p1, e1 = emul._model(xd, ad, bd, exd, ae, None, False)
    assert torch.equal(p0, p1) and torch.equal(e0, e1)
    OE.DROPOUT = None
    ref_p, ref_e = OT.model(args, params, norms, x, a, b, ex)
    close(p0, ref_p, 2e-5)
    close(e0, ref_e, 2e-5)


def test_fit_eval_with_dropout_trains(dev, networks):
    args, norms, params, emul, cpu_in, dev_in = _problem(networks, 'astlingen', dev, dropout=0.1, embed_size=64, n_sp_layer=1, learning_rate=1e-3)
    first = [float(v) for v in emul.fit_eval(*dev_in)]
    for _ in range(20):
        last = [float(v) for v in emul.fit_eval(*dev_in)]
    assert all(np.isfinite(first + last)) and sum(last) < sum(first)
    ev1 = [float(v) for v in emul.fit_eval(*dev_in, fit=False)]
    ev2 = [float(v) for v in emul.fit_eval(*dev_in, fit=False)]
    assert ev1 == ev2                                                          # evaluation draws no mask
