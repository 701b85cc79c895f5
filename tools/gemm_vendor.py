"""The vendor library (hipBLASLt behind torch.matmul, fp32, no TF32 on gfx950) on the CDT step's GEMM shapes: the
reference point for linear_pers_kernel / mlp_dw_big_kernel (profiles/r4_gemm_vendor.txt).  Not a product path."""
import torch

torch.backends.cuda.matmul.allow_tf32 = False
dev = torch.device("cuda:0")
M = 81920


def t(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


print("y = x W^T + b (forward) and dx = dy W (input gradient), M = 81920")
for (K, N) in [(256, 768), (256, 256), (256, 1024), (1024, 256), (768, 256)]:
    x, W, b = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), torch.randn(N, device=dev)
    us = t(lambda: torch.nn.functional.linear(x, W, b))
    gf = 2.0 * M * K * N * 1e-9
    Wt = torch.randn(K, N, device=dev)  # dx = dy[M, K'] @ W[K', N']: the NN form
    us2 = t(lambda: torch.mm(x, Wt))
    print(f"  K={K:5d} N={N:5d}  linear {us:7.1f} us {gf / us / 157.3 * 1e3:.3f} of peak | mm (NN) {us2:7.1f} us {gf / us2 / 157.3 * 1e3:.3f}")
print("dW = dy^T a")
tot = 0.0
for (out, inn) in [(768, 256), (256, 256), (1024, 256), (256, 1024)]:
    dy, a = torch.randn(M, out, device=dev), torch.randn(M, inn, device=dev)
    us = t(lambda: torch.mm(dy.t(), a), reps=5)
    gf = 2.0 * M * out * inn * 1e-9
    tot += us
    print(f"  out={out:5d} in={inn:5d}  {us:7.1f} us {gf / us / 157.3 * 1e3:.3f} of peak")
print(f"  one layer's four dW: {tot:.1f} us (x3 layers = {3 * tot:.0f} us; mlp_dw_big_kernel + reduce: ~3170 us)")
