set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
cat > /tmp/ar.py <<'PY'
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import gnn_uds_amd as U
from types import SimpleNamespace
dev = torch.device('cuda:0')
N, E, steps = 2000, 2500, 20
edges = U.synthetic_drainage_network(N, E, 0)
g = U.DrainageGraph.from_edges(edges)
a = SimpleNamespace(state_shape=(N, 4), edge_state_shape=(E, 4), seq_in=6, seq_out=1, embed_size=64, hidden_dim=64, kernel_size=3,
                    n_sp_layer=3, n_tp_layer=2, activation='relu', if_flood=3, edge_fusion=True, edges=edges, act=False, graph=g,
                    roll=steps, model_dir=None)
emul = U.Emulator('GAT', True, 'Conv1D', a, generator=torch.Generator().manual_seed(1)).to(dev)
rng = np.random.default_rng(0)
emul.set_norm(*[np.stack([0.5 + rng.random((n, c)), np.zeros((n, c))]) for n, c in ((N, 5), (N, 1), (N, 5), (N, 1), (E, 4))])
x, b, ex = torch.rand(1, 6, N, 5, device=dev), torch.rand(1, steps, N, 1, device=dev) * 0.1, torch.rand(1, 6, E, 4, device=dev)
for _ in range(2):
    emul._model(x, None, b, ex)
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ar -- python3 /tmp/ar.py > $O/ar.log 2>&1
cp $O/ar/*/*_kernel_stats.csv $O/ar_kernel_stats.csv
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/ar/*/*_kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last 1/40 of kernels = one step approx; print the last 80 kernels with durations and gaps
last = rows[-90:]
prev_end = None
out = []
for r in last:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    out.append('%-70s dur=%6.1fus gap=%6.1fus' % (r['Kernel_Name'][:70], (e - s) / 1e3, 0 if prev_end is None else (s - prev_end) / 1e3))
    prev_end = e
open('gpurun_out/ar_tail.txt', 'w').write('\n'.join(out))
print(len(rows))
PY
rm -rf $O/ar
