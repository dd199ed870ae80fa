import torch, time, json
dev = torch.device('cuda:0')
def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
out = {}
for (M, N, K) in ((10000, 1920, 12000), (10000, 1920, 36000), (8192, 8192, 8192)):
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    b = torch.randn(K, N, device=dev, dtype=torch.bfloat16)
    ms = timed(lambda: torch.matmul(a, b))
    out['bf16_%dx%dx%d' % (M, N, K)] = {'ms': ms, 'tflops': 2.0 * M * N * K / ms / 1e9}
print(json.dumps(out))
