"""Timing of the streaming GEMM kernels at the detection workload's shapes (B=8, 1333x800)."""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "aloception-oss_amd"))
import alo_hip

def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3

torch.manual_seed(0)
dev = "cuda"
with torch.no_grad():
    for (M, K, N, relu, res) in [(177784, 256, 256, False, False), (177784, 256, 128, False, False), (534400, 64, 256, True, True),
                                 (133600, 128, 512, True, True), (33600, 256, 1024, True, True), (534400, 256, 64, True, False),
                                 (534400, 64, 64, True, False)]:
        x = torch.randn(M, K, device=dev).to(torch.bfloat16)
        w = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
        b = torch.randn(N, device=dev).to(torch.bfloat16)
        r = torch.randn(M, N, device=dev).to(torch.bfloat16) if res else None
        ref = F.linear(x.float(), w.float(), b.float())
        if res: ref = ref + r.float()
        if relu: ref = F.relu(ref)
        got = alo_hip.linear_shortk(x, w, b, relu, residual=r)
        err = (got.float() - ref).abs().max().item()
        t = timeit(lambda: alo_hip.linear_shortk(x, w, b, relu, residual=r))
        nbytes = 2.0 * (M * K + M * N * (2 if res else 1))
        print(f"linear_shortk M={M} K={K} N={N} relu={relu} res={res}: err {err:.4f}  {t:.1f} us  {nbytes / t / 1e6:.2f} TB/s", flush=True)
    for (M, K, N, relu, res) in [(133600, 512, 128, True, False), (33600, 1024, 256, True, False), (8400, 512, 2048, True, True),
                                 (33600, 512, 1024, False, False), (8400, 1024, 2048, False, False), (133600, 512, 256, True, False),
                                 (8400, 2048, 512, True, False), (33600, 1024, 512, True, False), (8400, 2048, 256, False, False)]:
        x = torch.randn(M, K, device=dev).to(torch.bfloat16)
        w = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
        b = torch.randn(N, device=dev).to(torch.bfloat16)
        r = torch.randn(M, N, device=dev).to(torch.bfloat16) if res else None
        t = timeit(lambda: alo_hip.linear_packed(x, w, b, relu, residual=r))
        if res:
            t0 = timeit(lambda: torch.relu_(torch.addmm(b, x, w.t()) + r))
        else:
            t0 = timeit(lambda: torch._addmm_activation(b, x, w.t(), use_gelu=False) if relu else torch.addmm(b, x, w.t()))
        print(f"linear_packed M={M} K={K} N={N} res={res}: {t:.1f} us ({2.0 * M * N * K / t / 1e6:.0f} TF/s)   stock {t0:.1f} us", flush=True)
    x = torch.randn(8, 22223, 256, device=dev).to(torch.bfloat16)
    w = (torch.randn(256, 256, device=dev) / 16).to(torch.bfloat16); b = torch.randn(256, device=dev).to(torch.bfloat16)
    t = timeit(lambda: alo_hip.value_proj_head_major(x, w, b, None, 8))
    print(f"value_proj_head_major: {t:.1f} us", flush=True)
    w1 = (torch.randn(1024, 256, device=dev) / 16).to(torch.bfloat16); b1 = torch.randn(1024, device=dev).to(torch.bfloat16)
    w2 = (torch.randn(256, 1024, device=dev) / 32).to(torch.bfloat16)
    t = timeit(lambda: alo_hip.ffn256(x, w1, b1, w2, b))
    print(f"ffn256: {t:.1f} us", flush=True)
