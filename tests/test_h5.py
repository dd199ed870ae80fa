"""gnn_uds_amd/h5.py: the HDF5 subset a Keras `model.h5` uses, read back from files built by tests/h5_writer.py (the same
format specification written independently as a writer); refusal of what the reader does not cover; Emulator.load('.h5')."""
import json
import os
import struct

import numpy as np
import pytest
import torch

import gnn_uds_amd as U
from gnn_uds_amd import h5
from tests import h5_writer as W
from tests.util import emulator_args

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_round_trip_of_nested_groups_and_layouts(tmp_path):
    rng = np.random.default_rng(0)
    tree = {'dense': {'dense': {'kernel:0': rng.random((5, 64), dtype=np.float32), 'bias:0': rng.random(64, dtype=np.float32)}},
            'node_edge_3': {'node_edge_3': {'weight:0': rng.random((30, 29)).astype(np.float32), 'bias:0': np.zeros((30, 29), np.float32)}},
            'gru': {'gru': {'gru_cell': {'kernel:0': rng.random((8, 12)).astype(np.float32), 'bias:0': rng.random((2, 12)).astype(np.float32)}}},
            'input_1': {}, 'scalar': np.float64(2.5) * np.ones(()), 'ints': np.arange(7, dtype=np.int32), 'empty': np.zeros((0, 3), np.float32)}
    for i in range(40):                                  # many layers: several symbol-table nodes under the root B-tree
        tree['conv1d_%d' % i] = {'conv1d_%d' % i: {'kernel:0': rng.random((3, 4, 2)).astype(np.float32)}}
    p = str(tmp_path / 'model.h5')
    W.write_tree(p, tree, nodes=5, compact=lambda k: k.endswith('bias:0'), split=lambda k: 'kernel' in k,
                 dataspace_version=lambda k: 2 if 'node_edge' in k else 1)
    got = h5.read_datasets(p)

    def flat(node, prefix=''):
        for k, v in node.items():
            if isinstance(v, dict):
                yield from flat(v, prefix + k + '/')
            else:
                yield prefix + k, np.asarray(v)
    want = dict(flat(tree))
    assert set(got) == set(want)
    for k, v in want.items():
        assert got[k].shape == v.shape and got[k].dtype == v.dtype and np.array_equal(got[k], v), k


def test_unsupported_files_fail_loudly(tmp_path):
    data = bytearray(W.write_tree(str(tmp_path / 'a.h5'), {'x': np.ones(3, np.float32)}))
    with pytest.raises(h5.H5FormatError):
        (tmp_path / 'b.h5').write_bytes(b'not an hdf5 file' * 10)
        h5.read_datasets(str(tmp_path / 'b.h5'))
    bad = bytearray(data)
    bad[8] = 2                                           # superblock version 2 = libver='latest'
    (tmp_path / 'c.h5').write_bytes(bytes(bad))
    with pytest.raises(NotImplementedError, match='superblock version 2'):
        h5.read_datasets(str(tmp_path / 'c.h5'))
    bad = bytearray(data)
    # the layout message (type 0x0008): version 3, class 1 (contiguous) -> class 2 (chunked)
    pos = [k for k in range(len(bad) - 1) if bad[k] == 3 and bad[k + 1] == 1 and bad[k - 8:k - 6] == struct.pack('<H', 0x08)]
    assert pos
    bad[pos[0] + 1] = 2
    (tmp_path / 'd.h5').write_bytes(bytes(bad))
    with pytest.raises(NotImplementedError, match='chunked'):
        h5.read_datasets(str(tmp_path / 'd.h5'))


def test_emulator_loads_a_keras_weight_file(tmp_path):
    """`Emulator.load('<dir>')` / `load('<file>.h5')` with a Keras `model.h5` (reference emulator.py:833-838): every parameter
    arrives, under the creation-order layer names, incl. the nested cell weights of a GRU layer."""
    with open(os.path.join(ROOT, 'tests', 'golden', 'networks.json')) as fh:
        net = json.load(fh)['astlingen']
    args = emulator_args(np.array(net['edges']), net['n_node'], recurrent='GRU', embed_size=32, hidden_dim=16, n_sp_layer=1)
    a = U.Emulator(args.conv, args.resnet, args.recurrent, args, generator=torch.Generator().manual_seed(3))
    with torch.no_grad():
        for q in a.parameters():
            if float(q.abs().sum()) == 0:
                q.add_(torch.rand(q.shape, generator=torch.Generator().manual_seed(q.numel())) * 0.1)      # biases: not all zero
    w = a.export_keras_weights()
    assert 'gru/gru_cell/recurrent_kernel:0' in w
    W.write_tree(str(tmp_path / 'model.h5'), W.keras_tree(w), nodes=3)
    for target in (str(tmp_path), str(tmp_path / 'model.h5')):
        b = U.Emulator(args.conv, args.resnet, args.recurrent, args, generator=torch.Generator().manual_seed(9))
        b.load(target)
        for (n1, p1), (n2, p2) in zip(a.named_parameters(), b.named_parameters()):
            assert n1 == n2 and torch.equal(p1, p2), n1
