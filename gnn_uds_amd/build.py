"""Compile libuds_hip.so for gfx950 with hipcc (in-tree, no JIT cache).

    python -m gnn_uds_amd.build [--force]

hipcc cross-compiles without a GPU.  The library's only runtime dependency is libamdhip64.so.7;
inside a PyTorch-ROCm process the copy torch already loaded is the one that gets bound (same
SONAME), so kernels run on torch's streams and device pointers.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libuds_hip.so')
ARCH = 'gfx950'


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(ROOT, 'include', 'uds_hip.h')]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Build the shared library if missing or older than its sources; returns its path."""
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    cmd = [hipcc, '-O3', '-std=c++17', '--offload-arch=' + ARCH, '-fPIC', '-shared', '-fgpu-rdc' if False else '-fno-gpu-rdc',
           '-Wall', '-Wno-unused-function', '-I', os.path.join(ROOT, 'include'), '-o', LIB] + sources()
    for extra in os.environ.get('UDS_DEFINES', '').split():   # experiment switches, e.g. UDS_DEFINES='-DUDS_P3_PIPE'
        cmd.insert(1, extra)
    if os.environ.get('UDS_PHASE_TIMING'):      # diagnostic build: per-phase cycle stamps into the workspace
        cmd.insert(1, '-DUDS_PHASE_TIMING')
    if verbose:
        cmd.insert(1, '-Rpass-analysis=kernel-resource-usage')
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError('hipcc failed:\n%s\n%s' % (' '.join(cmd), res.stderr[-4000:]))
    if verbose:
        sys.stderr.write(res.stderr)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose='-v' in sys.argv))
