"""A/B the fused kernel across differently built copies of libuds_hip.so (kernel experiments; cross-compiled with
`UDS_DEFINES=... python tools/build_variant.py NAME`), interleaved rounds, one library per child process:
    python tools/variant_bench.py [--rounds R] [--S 60] build_variants/base.so build_variants/x.so ...
The first library is the reference for the output comparison (max abs difference of four snapshots)."""
import json, os, subprocess, sys
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
args = sys.argv[1:]
rounds, S, fx, fe = 2, 60, 64, 64
while args and args[0].startswith('--'):
    k = args.pop(0)
    v = args.pop(0)
    if k == '--rounds': rounds = int(v)
    elif k == '--S': S = int(v)
    elif k == '--fx': fx = int(v)
    elif k == '--fe': fe = int(v)
libs = args
os.makedirs('gpurun_out', exist_ok=True)
res = {l: [] for l in libs}
ref = None
for r in range(rounds):
    for l in libs:
        out = 'gpurun_out/var_%s.pt' % os.path.basename(l)
        env = dict(os.environ, UDS_LIB_PATH=os.path.abspath(l))
        p = subprocess.run([sys.executable, os.path.join(HERE, 'fused_time.py'), out, str(S), '30', str(fx), str(fe)], env=env, capture_output=True, text=True)
        if p.returncode != 0:
            print('FAILED', l, p.stderr[-2000:], flush=True)
            res[l].append(None)
            continue
        rec = json.loads(p.stdout.strip().splitlines()[-1])
        o = torch.load(out)
        if ref is None:
            ref = o
        os.remove(out)
        rec['max_abs_diff_vs_first'] = max(float((o['x'] - ref['x']).abs().max()), float((o['e'] - ref['e']).abs().max()))
        res[l].append(rec)
        print('%-40s round %d: median %.1f us  min %.1f us  diff %.3g  tiles %s+%s' % (os.path.basename(l), r, rec['us_median'], rec['us_min'],
              rec['max_abs_diff_vs_first'], rec['plan'].get('node_tiles'), rec['plan'].get('link_tiles')), flush=True)
print('--- summary (min over rounds of the median) ---')
base = None
for l in libs:
    ok = [x['us_median'] for x in res[l] if x]
    if not ok:
        print('%-40s FAILED' % os.path.basename(l)); continue
    m = min(ok)
    base = base or m
    print('%-40s %.1f us  (%+.1f %% vs first)  diff %.3g' % (os.path.basename(l), m, 100 * (m / base - 1), max(x['max_abs_diff_vs_first'] for x in res[l] if x)))
