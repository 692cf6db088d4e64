"""Reality check (not part of the product): what the vendor GEMM library reaches on the plain-GEMM equivalents of the nine
conv launches (same M, N, K; no im2col, no epilogue), bf16 -> bf16 (or f16 -> f16: argv[1]) with fp32 accumulation, via torch.matmul."""
import sys, time, torch
DT = {"bf16": torch.bfloat16, "f16": torch.float16}[sys.argv[1] if len(sys.argv) > 1 else "bf16"]
dev = torch.device("cuda", 0)
def bench(M, N, K, trans=False, reps=20):
    if trans:   # wgrad-like: (K x M)^T (K x N): reduction over the long dimension
        a = torch.randn(K, M, device=dev, dtype=DT); b = torch.randn(K, N, device=dev, dtype=DT)
        f = lambda: torch.matmul(a.t(), b)
    else:
        a = torch.randn(M, K, device=dev, dtype=DT); b = torch.randn(N, K, device=dev, dtype=DT)
        f = lambda: torch.matmul(a, b.t())
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    print("%-6s M=%7d N=%5d K=%7d  %8.1f us  %7.1f TFLOP/s" % ("TN" if trans else "NT", M, N, K, dt * 1e6, 2.0 * M * N * K / dt / 1e12))
n = 256
for (L, cin, cout) in [(3000, 128, 256), (1500, 256, 384), (750, 384, 512)]:
    bench(n * L, cout, 3 * cin)            # forward
    bench(n * L, cin, 3 * cout)            # dgrad
    bench(3 * cin, cout, n * L, True)      # wgrad
