"""Measurement only (never imported by the product): the time-to-first-token GEMM shapes on the vendor library (torch.matmul -> hipBLASLt / rocBLAS)
beside tools/gemm_probe's numbers for gemm_glds_kernel - what a tuned library reaches on shapes this small is the practical ceiling the
"35 % of the bf16 peak" target has to be read against. Same protocol as gemm_probe: bf16 operands, fp32-accumulated, 24 weight
copies rotated so the weights come from HBM, HIP events around 200 launches, best of 5."""
import torch

SHAPES = [("T5 q|k|v", 2048, 3072, 1024), ("T5 o", 2048, 1024, 1024), ("T5 wi (gated)", 2048, 5632, 1024), ("T5 wo", 2048, 1024, 2816),
          ("prefill qkv", 1056, 3072, 1024), ("prefill o / q", 1056, 1024, 1024), ("prefill fc1", 1056, 4096, 1024), ("prefill fc2", 1056, 1024, 4096),
          ("cross k|v", 2048, 2048, 1024), ("Large fc1", 1056, 6144, 1536), ("Large fc2", 1056, 1536, 6144)]


def main():
    dev = torch.device("cuda:0")
    for name, M, N, K in SHAPES:
        x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
        ws = [torch.randn(N, K, device=dev, dtype=torch.bfloat16) for _ in range(24)]
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        for w in ws[:3]:
            torch.matmul(x, w.t(), out=out)
        torch.cuda.synchronize()
        # one captured graph of 192 launches: torch's per-call dispatch (~18 us on this host) would otherwise be what is measured
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(192):
                torch.matmul(x, ws[i % 24].t(), out=out)
        g.replay()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.replay()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / 192)
        gf = 2.0 * M * N * K / 1e9
        print(f"{name:16s} M={M} N={N} K={K} ({gf:.1f} GFLOP)  vendor {best:7.2f} us  {gf / best:7.1f} TFLOP/s  {gf / best / 2500 * 100:5.1f} % of 2.5 PF", flush=True)


if __name__ == "__main__":
    main()
