"""Cross-compile a copy of the library with extra defines into build_variants/NAME.so (git-ignored, travels with gpurun):
    UDS_DEFINES='-DUDS_X=1' python tools/build_variant.py NAME"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gnn_uds_amd import build as B
name = sys.argv[1]
os.makedirs(os.path.join(ROOT, 'build_variants'), exist_ok=True)
out = os.path.join(ROOT, 'build_variants', name + '.so')
cmd = ['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-shared', '-fno-gpu-rdc', '-Wall', '-Wno-unused-function',
       '-I', os.path.join(ROOT, 'include'), '-o', out] + B.sources() + os.environ.get('UDS_DEFINES', '').split()
if '-v' in sys.argv:
    cmd.insert(1, '-Rpass-analysis=kernel-resource-usage')
r = subprocess.run(cmd, capture_output=True, text=True)
if r.returncode:
    sys.exit(r.stderr[-4000:])
if '-v' in sys.argv:
    sys.stderr.write(r.stderr)
print(out)
