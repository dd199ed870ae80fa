"""Time the dense remainder GEMM of a trained NodeEdge (uds_remainder_forward) against rocBLAS fp32 (torch.matmul), and one
spatial layer with a trained dense bias through the fused kernel.  env: NODES, LINKS, S."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_uds_amd as U
from gnn_uds_amd import _lib


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); t.append(time.perf_counter() - t0)
    return sorted(t)[len(t) // 2] * 1e3


def main():
    N, E, S = int(os.environ.get('NODES', 10000)), int(os.environ.get('LINKS', 12000)), int(os.environ.get('S', 60))
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(0)
    rest = (torch.randn(N, E, generator=g) * 0.01).to(dev)
    x = torch.randn(S, E, 32, generator=g).to(dev)
    packed = _lib.remainder_pack(rest)
    out = {}
    out['gemm_ms'] = timed(lambda: _lib.remainder_forward(packed, (N, E), x))
    out['rocblas_fp32_ms'] = timed(lambda: torch.matmul(rest, x))
    flops = 2.0 * N * E * S * 32
    out['gemm_tflops_fp32_equiv'] = flops / out['gemm_ms'] / 1e9
    out['gemm_tflops_bf16_issued'] = 3 * flops / out['gemm_ms'] / 1e9
    ref = torch.matmul(rest.double(), x[:2].double())
    got = _lib.remainder_forward(packed, (N, E), x[:2])
    out['err_vs_fp64'] = float((got.double() - ref).abs().max() / ref.abs().max())
    out['rocblas_err_vs_fp64'] = float((torch.matmul(rest, x[:2]).double() - ref).abs().max() / ref.abs().max())
    del rest, packed, ref, got
    gph = U.DrainageGraph.from_edges(U.synthetic_drainage_network(N, E, 0))
    for trained in (False, True):
        layer = U.SpatialLayer(gph, 64, 'relu', sparse_params=False).to(dev)
        if trained:
            with torch.no_grad():
                layer.node_edge_n.bias.normal_(0, 0.01)
                layer.node_edge_e.bias.normal_(0, 0.01)
        xs, es = torch.randn(S, N, 64, device=dev), torch.randn(S, E, 64, device=dev)
        with torch.no_grad():
            out['layer_%s_ms' % ('trained_bias' if trained else 'support_only')] = timed(lambda: layer(xs, es))
        out['path_%s' % ('trained_bias' if trained else 'support_only')] = layer.last_path
        del layer
    print(json.dumps(out))


if __name__ == '__main__':
    main()
