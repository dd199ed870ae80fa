"""CPU tests of the dropout restatement (oracle/dropout_ref.py): Philox4x32-10 against the known-answer vectors published with
the generator (Random123 `kat_vectors`, philox4x32 10 rounds), and the statistics the reference's keras Dropout has by
construction (keep fraction 1 - rate, expectation preserved).  The reference's own masks come from TensorFlow's stateful generator
and cannot be reproduced; these properties are what "statistical parity" means in tests/test_gpu_dropout.py."""
import numpy as np

from oracle import dropout_ref as DR

KAT = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
       ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
       ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]


def test_philox_known_answers():
    for counter, key, expect in KAT:
        got = DR.philox4x32_10(np.array(counter, dtype=np.uint32), key)
        assert tuple(int(v) for v in got) == expect


def test_philox_is_vectorised_over_counters():
    ctr = np.array([k[0] for k in KAT[:1] * 3], dtype=np.uint32)
    got = DR.philox4x32_10(ctr, KAT[0][1])
    assert got.shape == (3, 4) and all(tuple(int(v) for v in row) == KAT[0][2] for row in got)


def test_mask_statistics_and_stream_properties():
    n = 1 << 20
    for rate in (0.1, 0.2, 0.5):
        m = DR.dropout_mask(n, rate, seed=12345, offset=0)
        sigma = (rate * (1 - rate) / n) ** 0.5
        assert abs(m.mean() - (1 - rate)) < 5 * sigma
    # the mask of element i depends on offset + i only: a stream read in two pieces equals the stream read at once
    whole = DR.dropout_mask(1000, 0.2, 7, 40)
    assert np.array_equal(whole[:333], DR.dropout_mask(333, 0.2, 7, 40)) and np.array_equal(whole[333:], DR.dropout_mask(667, 0.2, 7, 373))
    assert not np.array_equal(whole, DR.dropout_mask(1000, 0.2, 8, 40))
    assert DR.dropout_mask(64, 0.0, 1, 0).all()
    x = np.random.default_rng(0).standard_normal(n)
    y = DR.dropout(x + 3.0, 0.2, 99, 0)
    assert abs(y.mean() - 3.0) < 0.02           # E[dropout(x)] = x


def test_oracle_hooks_are_identity_at_inference_and_scale_in_training():
    """The restatement's Dropout sites (oracle.emulator_ref.DROPOUT, oracle.spektral_dense.ATTN_DROPOUT): with no hook set a model
    built with dropout > 0 computes exactly what the dropout-free model computes (Keras: Dropout is the identity unless
    training=True); with an all-keep hook that only counts, every site of `build_network` is visited in the reference's order
    (x, b, e, ae embeddings; x, e behind each spatial layer; the two resnet Dense outputs) and every GATConv once."""
    import json, os
    import torch
    from oracle import emulator_ref as OE
    from oracle import spektral_dense as OD
    from tests.util import emulator_args
    with open(os.path.join(os.path.dirname(__file__), 'golden', 'networks.json')) as fh:
        net = json.load(fh)['astlingen']
    edges = np.array(net['edges'])
    a0 = emulator_args(edges, net['n_node'], n_sp_layer=2, seq_in=4, seq_out=2)
    a1 = emulator_args(edges, net['n_node'], n_sp_layer=2, seq_in=4, seq_out=2, dropout=0.3)
    params = OE.init_params(a0, seed=2)
    c = OE.config(a0)
    g = torch.Generator().manual_seed(0)
    rnd = lambda *s: torch.rand(*s, generator=g, dtype=torch.float64)
    X, B, E = rnd(2, c.seq_in, c.n_node, c.n_in), rnd(2, c.seq_out, c.n_node, c.b_in), rnd(2, c.seq_in, c.n_edge, c.e_in)
    AE = rnd(2, c.seq_out, c.n_edge, 1) if c.act else None
    y0, e0 = OE.forward(a0, params, X, B, E, AE)
    y1, e1 = OE.forward(a1, params, X, B, E, AE)
    assert torch.equal(y0, y1) and torch.equal(e0, e1)
    seen, attn = [], []
    OE.DROPOUT = lambda t, rate: (seen.append((tuple(t.shape), rate)), t)[1]
    OD.ATTN_DROPOUT = lambda coef, a_hat: (attn.append(tuple(coef.shape)), coef)[1]
    try:
        y2, e2 = OE.forward(a1, params, X, B, E, AE)
    finally:
        OE.DROPOUT = OD.ATTN_DROPOUT = None
    assert torch.equal(y0, y2) and torch.equal(e0, e2)
    n_emb = 4 if c.act else 3
    assert len(seen) == n_emb + 2 * 2 * c.L + 2, seen                          # embeddings, x and e behind each of the 2 L layers, two resnet outputs
    assert [r for _, r in seen[:n_emb]] == [0.2] * n_emb and all(r == 0.3 for _, r in seen[n_emb:])
    assert len(attn) == 2 * 2 * c.L                                             # gat_x and gat_e of every spatial layer
