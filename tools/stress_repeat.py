"""Bitwise repeatability of the fused block under load: 150 repeats of a 3-layer block at the headline, C3 and C2 sizes must equal
the first run exactly (the hand-counted vmcnt waits of k_fused_ws are where a race would show as a run-to-run difference)."""
import torch, sys
sys.path.insert(0, '.')
import gnn_uds_amd as U
dev = torch.device('cuda', 0)
for (N, E, S) in [(10000, 12000, 60), (50000, 65000, 32), (2000, 2500, 100)]:
    g = U.DrainageGraph.from_edges(U.synthetic_drainage_network(N, E, 0))
    blk = U.SpatialBlock(g, 64, 3, 'relu', sparse_params=True, generator=torch.Generator().manual_seed(1)).to(dev)
    x, e = torch.rand(S, N, 64, device=dev), torch.rand(S, E, 64, device=dev)
    with torch.no_grad():
        rx, re = blk(x, e)
        bad = 0
        for i in range(150):
            ox, oe = blk(x, e)
            if not (torch.equal(ox, rx) and torch.equal(oe, re)):
                bad += 1
    torch.cuda.synchronize()
    print(N, E, S, 'mismatching repeats:', bad, flush=True)
