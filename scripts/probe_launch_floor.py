"""Per-launch floor of dependent tiny kernels on one stream, issued from C (launch plan) so that the host is not the limit:
what a launch costs the device when there is nothing to compute, and what two tiny kernels cost against one fused one."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusionmodel_amd import _lib as L
from diffusionmodel_amd.ops import call, ptr
from diffusionmodel_amd.graph import LaunchPlan

dev = "cuda:0"
K = 2000
for n in (1024, 65536, 1 << 20, 1 << 22):
    x = torch.randn(n, device=dev)
    y = torch.empty_like(x)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            call("dm_act_fwd", ptr(x), ptr(y), n, L.ACT_GELU)
        g = torch.cuda.CUDAGraph(keep_graph=True)
        with torch.cuda.graph(g, stream=s):
            for i in range(K):
                call("dm_act_fwd", ptr(x if i & 1 == 0 else y), ptr(y if i & 1 == 0 else x), n, L.ACT_GELU)
        plan = LaunchPlan(g)
        plan.run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        plan.run()
        e1.record()
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        print(f"n={n:8d}: {plan.n_kernels} launches, device {e0.elapsed_time(e1) / K * 1e3:6.2f} us per launch, host {t_host / K * 1e6:6.2f} us per launch "
              f"({n * 8 / 1e3:.0f} KB moved per launch)")
