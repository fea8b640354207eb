import sys, os
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "ltx-video-gpupoor_amd"))
import torch
from ltxmi import ops
def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for (M, N) in [(14976, 2048), (14976, 8192)]:
    for K in [64, 256, 1024, 2048, 4096, 8192]:
        a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
        b = torch.randn(N, device="cuda").to(torch.bfloat16)
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        ms = timeit(lambda: ops.gemm(a, w, b, out=out))
        print(f"M{M} N{N} K{K}: {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:7.1f} TF", flush=True)
