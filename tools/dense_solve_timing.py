"""dev tool: what the dense alternative to the PCG would cost (BASELINE.json north_star: "MFMA only if the dense 6k x 6k
reduced solve proves bandwidth-bound as a GEMM").  Times, with hipEvents, a dense fp32 Cholesky factorisation + solve
(rocSOLVER / hipBLASLt through torch.linalg) of SPD systems of the sizes of this path, next to the GEMM of the same
size (the MFMA-rate reference) and a copy of the matrix (the bandwidth reference):
  2 048   scalar system of the reference-parity solve at C2 (A (x) I3: one matrix, three right-hand sides)
  6 144   the "6k x 6k" system of the north-star text (2 048 nodes x 3 translations as ONE system)
  12 288  the 6-DoF system at C2 (2 048 nodes x 6)
"""
import sys
import numpy as np
import torch

dev = torch.device("cuda", 0)


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))


for n, rhs in ((2048, 3), (6144, 1), (12288, 1)):
    g = torch.Generator(device=dev).manual_seed(n)
    m = torch.randn((n, n), device=dev, generator=g) / np.sqrt(n)
    A = m @ m.T + torch.eye(n, device=dev)
    b = torch.randn((n, rhs), device=dev, generator=g)
    t_chol = timeit(lambda: torch.linalg.cholesky(A))
    L = torch.linalg.cholesky(A)
    t_solve = timeit(lambda: torch.cholesky_solve(b, L))
    t_gemm = timeit(lambda: A @ A)
    t_copy = timeit(lambda: A.clone())
    x = torch.cholesky_solve(b, L)
    err = float((A @ x - b).abs().max())
    print("n = %5d: cholesky %8.3f ms (%.1f TFLOP/s of n^3/3), solve %.3f ms, | GEMM n^3 %.3f ms (%.1f TFLOP/s), copy %.3f ms (%.0f GB/s) | residual %.1e"
          % (n, t_chol, n ** 3 / 3 / t_chol / 1e9, t_solve, t_gemm, 2 * n ** 3 / t_gemm / 1e9, t_copy, 8 * n * n / t_copy / 1e6, err), flush=True)
