"""`.inp` loader (gnn_uds_amd/inp.py): a hand-written SWMM input file with every section the
reference reads, and -- in the build container only, where /root/reference exists -- the five networks the reference
ships, against the committed integer fixtures (tests/golden/networks.json)."""
import json
import os

import numpy as np
import pytest

import gnn_uds_amd as U
from gnn_uds_amd import inp as I

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_NETS = '/root/reference/surrogate/envs/network'

TOY = """[TITLE]
toy network ; comment

[JUNCTIONS]
;;Name  Elevation  MaxDepth  InitDepth  SurDepth  Aponded
J1      10.0       2.0       0          0.5       0
J2      9.5        1.5       0          0         0
J3      9.0        3.0       0          0         0

[OUTFALLS]
;;Name  Elevation  Type  Stage  Gated
O1      8.0        FREE         NO

[STORAGE]
;;Name  Elev  MaxDepth  InitDepth  Shape  Params
S1      9.2   4.0       0          FUNCTIONAL 1000 0 0 0 0

[CONDUITS]
;;Name  From  To   Length  Rough  InOff OutOff
C1      J1    J2   100.5   0.01   0     0
C2      J2    S1   50      0.01   0     0
C3      S1    J3   75      0.01   0     0
C9      J3    XX   10      0.01   0     0      ; end node not listed: dropped

[PUMPS]
P1      J3    O1   CURVE1  ON  0  0

[ORIFICES]
R1      J1    J3   SIDE    0   0.65   NO  0

[XSECTIONS]
;;Link  Shape     Geom1  Geom2
C1      CIRCULAR  0.8    0
C2      CIRCULAR  1.0    0
C3      RECT_OPEN 1.2    2
R1      CIRCULAR  0.3    0
"""


def test_toy_inp(tmp_path):
    path = tmp_path / 'toy.inp'
    path.write_text(TOY)
    net = I.load_network(str(path))
    assert net.nodes == ['J1', 'J2', 'J3', 'O1', 'S1']                 # JUNCTIONS, OUTFALLS, (DIVIDERS,) STORAGE
    assert net.links == ['C1', 'C2', 'C3', 'P1', 'R1']                 # CONDUITS, PUMPS, ORIFICES; C9 dropped
    assert net.edges.tolist() == [[0, 1], [1, 4], [4, 2], [2, 3], [0, 2]]
    assert net.lengths.tolist() == [100.5, 50.0, 75.0, 0.0, 0.0]
    assert net.is_outfall.tolist() == [0, 0, 0, 1, 0] and net.is_storage.tolist() == [0, 0, 0, 0, 1]
    assert net.hmax.tolist() == [2.5, 1.5, 3.0, 0.0, 4.0] and net.hmin.tolist() == [0.0] * 5
    assert net.ehmax.tolist() == [0.8, 1.0, 1.2, 0.0, 0.3]
    head = I.load_network(str(path), head=True)
    assert head.hmin.tolist() == [10.0, 9.5, 9.0, 8.0, 9.2] and head.hmax.tolist() == [12.5, 11.0, 12.0, 8.0, 13.2]
    g = U.DrainageGraph.from_inp(str(path))
    assert (g.n_node, g.n_edge) == (5, 5)
    same = U.DrainageGraph.from_edges(net.edges, 5)
    for a, b in ((g.adj, same.adj), (g.edge_adj, same.edge_adj), (g.inc_n, same.inc_n)):
        assert np.array_equal(a.rowptr, b.rowptr) and np.array_equal(a.col, b.col)
    args = I.emulator_args(str(path), embed_size=8, seq_in=3)
    assert args.state_shape == (5, 4) and args.edge_state_shape == (5, 4) and args.embed_size == 8 and args.graph.n_edge == 5


@pytest.mark.skipif(not os.path.isdir(REF_NETS), reason='the reference data files exist in the build container only')
def test_reference_networks_match_the_committed_fixtures():
    with open(os.path.join(ROOT, 'tests', 'golden', 'networks.json')) as fh:
        golden = json.load(fh)
    seen = 0
    for name, ref in golden.items():
        path = os.path.join(REF_NETS, name, name + '.inp')
        if not os.path.exists(path):
            continue
        net = I.load_network(path)
        assert net.n_node == ref['n_node'] and net.edges.tolist() == ref['edges']
        assert net.is_outfall.astype(int).tolist() == ref['is_outfall']
        assert np.allclose(net.lengths, ref['lengths'])
        assert len(net.ehmax) == net.n_edge and (net.hmax >= 0).all()
        seen += 1
    assert seen == 5
