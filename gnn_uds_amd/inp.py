"""SWMM `.inp` -> drainage network, without swmm_api.

The reference reads its networks through `swmm_api.read_inp_file` (`envs/scenario/base.py:277-365`): node order =
sections JUNCTIONS, OUTFALLS, DIVIDERS, STORAGE in that order, link order = CONDUITS, PUMPS, ORIFICES, WEIRS, OUTLETS
(swmm-api==0.2.0.18.3 `NODE_SECTIONS` / `LINK_SECTIONS`, restated: the package is not part of the reference tree), a
link whose end is not a listed node is dropped (`base.py:359`).  This module is a plain text reader for the same
sections; it produces what `get_args` derives from the file (`base.py:277-330`):

    edges (E,2) [from, to], lengths, is_outfall, is_storage, hmax = MaxDepth + SurDepth (+ Elevation for head-based
    states), hmin = Elevation or 0, ehmax = XSECTIONS.Geom1 per link

and a `DrainageGraph` in CSR (no dense N x N matrices).  Host-side integer / text bookkeeping only.
"""
from types import SimpleNamespace

import numpy as np

from .graph import DrainageGraph

NODE_SECTIONS = ('JUNCTIONS', 'OUTFALLS', 'DIVIDERS', 'STORAGE')
LINK_SECTIONS = ('CONDUITS', 'PUMPS', 'ORIFICES', 'WEIRS', 'OUTLETS')


def read_sections(path):
    """{SECTION: [token rows]} of a SWMM input file; `;` starts a comment, section names are upper-cased."""
    sections, cur = {}, None
    with open(path, errors='replace') as fh:
        for raw in fh:
            line = raw.split(';')[0].strip()
            if not line:
                continue
            if line.startswith('['):
                cur = line.strip('[]').strip().upper()
                sections.setdefault(cur, [])
            elif cur is not None:
                sections[cur].append(line.split())
    return sections


def _num(row, idx):
    try:
        return float(row[idx])
    except (IndexError, ValueError):
        return 0.0


def load_network(path, head=False):
    """Parse `path` into a namespace: nodes, links (names), edges, lengths, is_outfall, is_storage, hmax, hmin, ehmax,
    n_node, n_edge, graph.  head=True mirrors the reference's head-based states (`base.py:288-291`: hmin = invert
    elevation, hmax += hmin)."""
    sec = read_sections(path)
    nodes, is_outfall, is_storage, depth, elev = [], [], [], [], []
    for name in NODE_SECTIONS:
        for row in sec.get(name, []):
            nodes.append(row[0])
            is_outfall.append(1 if name == 'OUTFALLS' else 0)
            is_storage.append(1 if name == 'STORAGE' else 0)
            elev.append(_num(row, 1))
            if name == 'JUNCTIONS':          # Name Elevation MaxDepth InitDepth SurDepth Aponded
                depth.append(_num(row, 2) + _num(row, 4))
            elif name == 'STORAGE':          # Name Elevation MaxDepth InitDepth Shape ...   (no SurDepth attribute)
                depth.append(_num(row, 2))
            else:                            # outfalls / dividers expose neither in swmm_api 0.2.0.18.3: getattr(.., 0)
                depth.append(0.0)
    index = {}
    for i, n in enumerate(nodes):
        index.setdefault(n, i)               # list.index(): the first occurrence wins
    geom1 = {row[0]: _num(row, 2) for row in sec.get('XSECTIONS', [])}      # Link Shape Geom1 ...
    links, edges, lengths, ehmax = [], [], [], []
    for name in LINK_SECTIONS:
        for row in sec.get(name, []):
            if len(row) >= 3 and row[1] in index and row[2] in index:
                links.append(row[0])
                edges.append((index[row[1]], index[row[2]]))
                lengths.append(_num(row, 3) if name == 'CONDUITS' else 0.0)
                ehmax.append(geom1.get(row[0], 0.0))
    hmax = np.asarray(depth, dtype=np.float64)
    hmin = np.asarray(elev, dtype=np.float64) if head else np.zeros(len(nodes))
    if head:
        hmax = hmax + hmin
    edges = np.asarray(edges, dtype=np.int64).reshape(-1, 2)
    return SimpleNamespace(nodes=nodes, links=links, edges=edges, lengths=np.asarray(lengths, dtype=np.float64),
                           is_outfall=np.asarray(is_outfall, dtype=np.float64), is_storage=np.asarray(is_storage, dtype=np.float64),
                           hmax=hmax, hmin=hmin, ehmax=np.asarray(ehmax, dtype=np.float64), n_node=len(nodes), n_edge=len(edges),
                           graph=DrainageGraph.from_edges(edges, len(nodes)))


def emulator_args(path, head=False, **over):
    """An `args` namespace for `gnn_uds_amd.Emulator` built from a `.inp` file: the graph-derived attributes of the
    reference's `get_args` (`base.py:277-330`) plus the model defaults of `Emulator.__init__` (`emulator.py:48-127`);
    keyword arguments override (embed_size, seq_in, act_edges, ...)."""
    net = load_network(path, head)
    a = dict(state_shape=(net.n_node, 4), edge_state_shape=(net.n_edge, 4), edges=net.edges, graph=net.graph,
             is_outfall=net.is_outfall, hmax=net.hmax, hmin=net.hmin, ehmax=net.ehmax, nwei=np.ones(net.n_node),
             ewei=np.ones(net.n_edge), seq_in=6, seq_out=1, embed_size=64, hidden_dim=64, kernel_size=3, n_sp_layer=3,
             n_tp_layer=2, activation='relu', if_flood=0, edge_fusion=False, act=False, tide=head, model_dir=None)
    a.update(over)
    return SimpleNamespace(**a)
