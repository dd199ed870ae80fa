"""CPU oracle for the GNN-UDS graph-convolution hot path.  TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (PyTorch-CPU / NumPy, fp32 or fp64) of the
reference algorithm for the one path the engine accelerates:
`surrogate/emulator.py:18-45,129-163,166-341,400-438,604-641,680-770,803-810`
and `surrogate/envs/scenario/base.py:349-439` of Zhiyu014/GNN-UDS, plus the
third-party layer arithmetic the reference imports from `spektral==1.3.1`
(`requirements.txt:6`; GATConv / GCNConv, restated from the package's published
algorithm because its source is not in this container) on `tensorflow==2.10.0`
/ `keras==2.10.0` (`requirements.txt:1-2`).

**PARITY UNPINNED.**  The reference ships no tests, no golden vectors, no
weights and cannot be imported here (tensorflow / spektral / pystorms are absent:
ordinary ModuleNotFoundError, nothing was denied).  The oracle is therefore
pinned only by (1) two independent restatements that must agree (dense-masked,
op-for-op what Spektral does, vs sparse CSR), (2) hand-computed known-answer
cases, (3) integer fixtures extracted from the five SWMM `.inp` data files the
reference ships (`tests/golden/`).

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this package.  The product (`gnn_uds_amd`) never does, and has no CPU
fallback: it raises when the HIP library is missing.
"""
