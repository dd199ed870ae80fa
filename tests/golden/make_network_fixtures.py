"""Extract the link lists of the five SWMM networks the reference ships into a small fixture.

Run in the build container only (needs /root/reference):
    python tests/golden/make_network_fixtures.py
Reads DATA files (`surrogate/envs/network/*/*.inp`), no reference code.  Node and link
order follow what the reference gets from swmm_api (`base.py:335-365`):
NODE_SECTIONS = JUNCTIONS, OUTFALLS, DIVIDERS, STORAGE; LINK_SECTIONS = CONDUITS, PUMPS,
ORIFICES, WEIRS, OUTLETS (swmm-api==0.2.0.18.3 `input_file/section_lists.py`, restated from
the published package: it is not installed here).  A link whose end is not a listed node is
dropped (`base.py:359`).  Output: tests/golden/networks.json
  {name: {"n_node": N, "is_outfall": [...], "edges": [[from,to],...], "lengths": [...]}}
"""
import glob
import json
import os

NODE_SECTIONS = ['JUNCTIONS', 'OUTFALLS', 'DIVIDERS', 'STORAGE']
LINK_SECTIONS = ['CONDUITS', 'PUMPS', 'ORIFICES', 'WEIRS', 'OUTLETS']


def read_sections(path):
    sections, cur = {}, None
    with open(path, errors='replace') as fh:
        for raw in fh:
            line = raw.split(';')[0].strip()
            if not line:
                continue
            if line.startswith('['):
                cur = line.strip('[]').upper()
                sections.setdefault(cur, [])
            elif cur is not None:
                sections[cur].append(line.split())
    return sections


def extract(path):
    sec = read_sections(path)
    nodes, is_outfall = [], []
    for name in NODE_SECTIONS:
        for row in sec.get(name, []):
            nodes.append(row[0])
            is_outfall.append(1 if name == 'OUTFALLS' else 0)
    index = {n: i for i, n in enumerate(nodes)}
    edges, lengths = [], []
    for name in LINK_SECTIONS:
        for row in sec.get(name, []):
            if row[1] in index and row[2] in index:
                edges.append([index[row[1]], index[row[2]]])
                lengths.append(float(row[3]) if name == 'CONDUITS' else 0.0)
    return dict(n_node=len(nodes), is_outfall=is_outfall, edges=edges, lengths=lengths)


if __name__ == '__main__':
    root = '/root/reference/surrogate/envs/network'
    out = {}
    for path in sorted(glob.glob(os.path.join(root, '*', '*.inp'))):
        out[os.path.basename(os.path.dirname(path))] = extract(path)
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, 'networks.json'), 'w') as fh:
        json.dump(out, fh, separators=(',', ':'))
    for k, v in out.items():
        print(k, v['n_node'], len(v['edges']))
