"""Development tool: builds a -DMF_WS_DIAG copy of the library, runs the fused backward and prints where wave 0 of
every workgroup spends its time (s_memtime stamps: shader cycles; the stamps themselves cost ~3 %)."""
import ctypes, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, "mentflow_amd", "csrc")
lib = os.path.join(ROOT, "gpurun_out", "libmentflow_diag.so")
os.makedirs(os.path.dirname(lib), exist_ok=True)
level = int(os.environ.get("FB_DIAG_LEVEL", "0"))        # activation hand-off level whose kernel is stamped (0, 1, 2)
os.environ["MENTFLOW_ACT_LEVEL"] = str(level)
extra = os.environ.get("WS_DIAG_FLAGS", "").split() + [f"-DMF_WS_DIAG_LEVEL={level}"]
# Only the fused-backward unit of the stamped level is rebuilt (with -DMF_WS_DIAG); every other object is the one the last
# __graft_entry__.build() left in mentflow_amd/csrc (they travel with the gpurun snapshot): ~25 s instead of minutes.
tus = [l.split() for l in open(os.path.join(csrc, "SOURCES.txt")) if l.strip() and not l.startswith("#")]
objs = []
for name, src, *flags in tus:
    if name == f"flow_bwd_fused_s{level}":
        obj = os.path.join(os.path.dirname(lib), f"diag_{name}.o")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DMF_WS_DIAG", *flags, *extra, "-c",
                        os.path.join(csrc, src), "-o", obj], check=True)
    else:
        obj = os.path.join(csrc, name + ".o")
        if not os.path.exists(obj):
            raise SystemExit(f"{obj} missing: run python __graft_entry__.py first")
    objs.append(obj)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", lib], check=True)
import torch
from mentflow_amd import _lib
_lib.use_library(lib)
import mentflow_amd as mf
dev = torch.device("cuda", 0)
torch.manual_seed(0)
gen = mf.generate.build_generator("nsf", device=dev, input_features=6, output_features=6, hidden_layers=3, hidden_units=64, transforms=1, bins=20)
n = 1 << 20
z = torch.randn(n, 6, device=dev)
for it in range(2):
    gen.zero_grad()
    x, lp = gen.sample_and_log_prob(n, z=z)
    (x.sum() / n + lp.mean()).backward()
torch.cuda.synchronize()
raw = (ctypes.c_ulonglong * (256 * 4 * 16))()
h = ctypes.CDLL(lib)
assert h.mf_debug_ws_read(raw) == 0
aw = np.array(raw, dtype=np.float64).reshape(256, 4, 16)      # [workgroup][wave][slot]
a = aw[:, 0, :]
groups = (n // 32) / 4 / 256
names = ["trunk fwd / loads+stage", "phi = W3 h | v wait (x d)", "rqs_apply (x d)", "barrier A (x d)", "stage gv (x d)", "barrier B (x d)",
         "dW last layer (x d)", "final flush (x groups!)", "gh += W3^T gv (x d)", "trunk bwd + dW", "gx", "TOTAL", " trunk: stage+barriers", " trunk: dW (x2)", " trunk: W^T chains (x2)", " level 0: stage + dW"]
print(f"hand-off level {level}; groups per workgroup:", groups, " (ticks of s_memtime = shader cycles)")
for q, nm in enumerate(names):
    print(f"  {nm:24s} {a[:, q].mean() / groups:9.1f} ticks per group   {100 * a[:, q].mean() / a[:, 11].mean():5.1f} %")

print("per-wave view (mean over workgroups, ticks per group):")
for q in (0, 1, 2, 3, 5, 6, 8, 9, 11):
    print(f"  {names[q]:24s} " + "  ".join(f"w{w}: {aw[:, w, q].mean() / groups:9.1f}" for w in range(4)))
