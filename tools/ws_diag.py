"""Development tool: builds a -DMF_WS_DIAG copy of the library, runs one backward and prints where the matrix / vector
waves of pair 0 spend their cycles (s_memtime stamps)."""
import ctypes, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, "mentflow_amd", "csrc")
lib = os.path.join(ROOT, "gpurun_out", "libmentflow_diag.so")
os.makedirs(os.path.dirname(lib), exist_ok=True)
extra = os.environ.get("WS_DIAG_FLAGS", "").split()
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DMF_WS_DIAG", *extra,
                os.path.join(csrc, "api.hip"), os.path.join(csrc, "kde.hip"), os.path.join(csrc, "flow.hip"), "-o", lib], check=True)
import torch
from mentflow_amd import _lib
_lib.use_library(lib)
import mentflow_amd as mf
dev = torch.device("cuda", 0)
torch.manual_seed(0)
gen = mf.generate.build_generator("nsf", device=dev, input_features=6, output_features=6, hidden_layers=3, hidden_units=64, transforms=2, bins=20)
n = 1 << 20
z = torch.randn(n, 6, device=dev)
for it in range(2):
    gen.zero_grad()
    x, lp = gen.sample_and_log_prob(n, z=z)
    (x.sum() / n + lp.mean()).backward()
torch.cuda.synchronize()
raw = (ctypes.c_ulonglong * (256 * 16))()
h = ctypes.CDLL(lib)
assert h.mf_debug_ws_read(raw) == 0
a = np.array(raw, dtype=np.float64).reshape(256, 16)
rounds = (n // 32) / 256 / 4
print("tile rounds per workgroup:", rounds)
for name, cols in (("matrix wave", {"prologue": 0, "barrier wait": 1, "steps": 2, "epilogue": 3, " slot read": 4,
                                    " W3^T mfma": 5, " phi mfma+write": 6}),
                   ("vector wave", {"prologue": 8, "barrier wait": 9, "steps": 10, " rqs_apply": 11, " tile store": 12,
                                    " slot write": 13})):
    print(name)
    for k, c in cols.items():
        print(f"   {k:14s} {a[:, c].mean() / rounds:10.0f} cycles per tile round (100 MHz ticks x ?)  mean over WGs")
