"""Which Python call sites issue device-to-device copies during ONE eager sampling step (they become memcpy nodes of the sampler's
hipGraph: __amd_rocclr_copyBuffer launches)?  Wraps the torch entry points that can copy and records the caller when the result
has new storage."""
import collections
import os
import sys
import traceback
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffusionmodel_amd as D

sites = collections.Counter()
ACTIVE = [False]


def wrap(name):
    orig = getattr(torch.Tensor, name)

    def f(self, *a, **k):
        out = orig(self, *a, **k)
        if ACTIVE[0] and isinstance(out, torch.Tensor) and self.is_cuda and (name in ("copy_", "clone") or out.data_ptr() != self.data_ptr()):
            st = [s for s in traceback.extract_stack()[:-1] if "diffusionmodel_amd" in s.filename]
            if st:
                sites[(name, os.path.basename(st[-1].filename), st[-1].lineno, st[-1].line)] += 1
        return out
    setattr(torch.Tensor, name, f)


for n in ("contiguous", "clone", "copy_", "float", "to", "repeat", "reshape", "long", "bfloat16"):
    wrap(n)
torch.manual_seed(0)
net = D.ContextUnet(3, 128, 4, bottleneck_k=4, dtype=torch.bfloat16)
ddpm = D.DDPM(net, (1e-4, 0.02), 1000, "cuda:0", drop_prob=0.1)
ddpm.eval()
ddpm.sample(64, (3, 64, 64), "cuda:0", guide_w=2.0, steps=2, seed=1)
torch.cuda.synchronize()
ACTIVE[0] = True
ddpm.sample(64, (3, 64, 64), "cuda:0", guide_w=2.0, steps=1, seed=1)
ACTIVE[0] = False
for k, v in sites.most_common(40):
    print(v, k)
