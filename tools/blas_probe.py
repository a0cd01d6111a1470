"""What does the vendor library reach on the encoder GEMM shapes?  (a yardstick only: the product
never calls it)."""
import torch, time, sys
M = int(sys.argv[1]) if len(sys.argv) > 1 else 256 * 197      # batch 256 (r03: 906 / 746 / 878 / 1059 TFLOP/s, profiles/r03_vendor_gemm.txt)
for (N, K) in [(2304, 768), (768, 768), (3072, 768), (768, 3072)]:
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        c = torch.nn.functional.linear(a, w)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        c = torch.nn.functional.linear(a, w)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"M={M} N={N} K={K}: {ms:.3f} ms  {2*M*N*K/ms/1e9:.0f} TFLOP/s", flush=True)
    del a, w, c
