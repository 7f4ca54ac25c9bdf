"""CFG sampling micro-run for profiling (GPU): n=64, 64x64, F=128, bf16, eager + hipGraph."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import diffusionmodel_amd as D
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
graph = (sys.argv[2] == "graph") if len(sys.argv) > 2 else True
torch.manual_seed(0)
net = D.ContextUnet(3, 128, 4, bottleneck_k=4, dtype=torch.bfloat16)
ddpm = D.DDPM(net, (1e-4, 0.02), 1000, "cuda:0", drop_prob=0.1)
ddpm.eval()
ddpm.sample(64, (3, 64, 64), "cuda:0", guide_w=2.0, steps=2, seed=1)
torch.cuda.synchronize()
t0 = time.perf_counter()
ddpm.sample(64, (3, 64, 64), "cuda:0", guide_w=2.0, steps=steps, seed=1, use_graph=graph)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{steps} steps graph={graph}: {dt / steps * 1e3:.3f} ms/step, {steps / dt:.1f} steps/s")
