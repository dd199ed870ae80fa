"""Time uds_remainder_forward at the headline shapes: python tools/gemm_time.py [h ...] (UDS_GEMM_FORM = 0 / 2 / 4 in -DUDS_KNOBS builds)."""
import sys, time, json, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_uds_amd import _lib
dev = torch.device('cuda', 0)
R, M, S = 10000, 12000, 60
for h in [int(a) for a in sys.argv[1:]] or [32, 64]:
    g = torch.Generator().manual_seed(0)
    rest = (torch.randn(R, M, generator=g) * 0.01).to(dev)
    x = torch.randn(S, M, h, generator=g).to(dev)
    packed = _lib.remainder_pack(rest)
    _lib.remainder_forward(packed, (R, M), x)
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        t0 = time.perf_counter()
        _lib.remainder_forward(packed, (R, M), x)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ms = sorted(ts)[len(ts) // 2] * 1e3
    fl = 2.0 * R * M * S * h
    print(json.dumps({'h': h, 'form': os.environ.get('UDS_GEMM_FORM'), 'ms': ms, 'tflops_fp32_eq': fl / ms / 1e9, 'frac_bf16_peak': 3 * fl / ms / 1e9 / 2500}), flush=True)
