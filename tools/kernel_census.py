"""Development tool (GPU box): which aten ops launch how many kernels in ONE eager training step at a small batch —
torch.profiler, grouped by the top-level op that launched them."""
import collections, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mentflow_amd.harness import build_problem
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda", 0)
fused = "--fused" in sys.argv
prob = build_problem(ndim=6, num=25, bins=64, xmax=4.0, seed=2, transforms=5, prior_scale=1.0, device=dev, meas_samples=50000,
                     penalty_parameter=100.0)
m = prob.model
opt = torch.optim.AdamW(m.parameters(), lr=1e-3, weight_decay=0.0, capturable=True, **({"fused": True} if fused else {}))
def step():
    opt.zero_grad(set_to_none=False)
    L, H, D = m.loss(25000)
    L.backward()
    opt.step()
for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()
ev = prof.events()
kern = [e for e in ev if e.device_type == torch.autograd.DeviceType.CUDA]
print("GPU kernels / memcpy / memset in one step:", len(kern))
c = collections.Counter(e.name[:70] for e in kern)
for k, v in c.most_common(30):
    print("  %4d  %s" % (v, k))
cpu = collections.Counter()
for e in ev:
    if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith(("aten::", "Optimizer", "autograd::", "FlowSampleFn", "ProjKde", "HistNorm", "EntropySums")) and e.cpu_parent is None:
        cpu[e.name] += 1
print("top-level CPU ops:")
for k, v in cpu.most_common(40):
    print("  %4d  %s" % (v, k))
