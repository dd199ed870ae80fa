import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_uds_amd import _lib
dev = torch.device('cuda', 0)
x = torch.rand(1, 60, 10000, 64, device=dev)
k = torch.rand(192, 64, device=dev) - 0.5
pk = _lib.rowgemm_pack(k)
for dil in (1, 2, 4):
    for _ in range(3):
        _lib.rowgemm_forward(x, pk, None, 64, 'relu', taps=3, dilation=dil)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        _lib.rowgemm_forward(x, pk, None, 64, 'relu', taps=3, dilation=dil)
    torch.cuda.synchronize()
    print('dil', dil, '%.1f us' % ((time.perf_counter() - t0) / 20 * 1e6), os.environ.get('UDS_NO_CONV_STREAM', 'stream'))
