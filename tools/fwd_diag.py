"""Development tool: -DMF_WS_DIAG build; cycle stamps of wave 0 of every workgroup in the generic forward kernel
(output-block MFMAs vs spline VALU, per tile) for 1024-, 512- and 256-thread workgroups."""
import ctypes, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, "mentflow_amd", "csrc")
lib = os.path.join(ROOT, "gpurun_out", "libmentflow_diag.so")
os.makedirs(os.path.dirname(lib), exist_ok=True)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DMF_WS_DIAG",
                os.path.join(csrc, "api.hip"), os.path.join(csrc, "kde.hip"), os.path.join(csrc, "flow.hip"), "-o", lib], check=True)
import torch
from mentflow_amd import _lib
_lib.use_library(lib)
import mentflow_amd as mf
dev = torch.device("cuda", 0)
torch.manual_seed(0)
gen = mf.generate.build_generator("nsf", device=dev, input_features=6, output_features=6, hidden_layers=3, hidden_units=64, transforms=1, bins=20)
n = 1 << 21
z = torch.randn(n, 6, device=dev)
h = ctypes.CDLL(lib)
blk = int(os.environ.get("MENTFLOW_FWD_BLOCK", "1024"))
with torch.no_grad():
    for _ in range(3):
        x, lp = gen.sample_and_log_prob(n, z=z)
torch.cuda.synchronize()
raw = (ctypes.c_ulonglong * (256 * 16))()
assert h.mf_debug_ws_read(raw) == 0
a = np.array(raw, dtype=np.float64).reshape(256, 16)
tiles = (n // 32) / 256 / (blk // 64)
print(f"block {blk}: tiles per wave {tiles}")
print(f"  output-block MFMA section : {a[:,0].mean()/tiles:9.0f} cycles per tile   (196 MFMAs = 12544 cycles if alone on the pipe)")
print(f"  spline VALU section       : {a[:,1].mean()/tiles:9.0f} cycles per tile")
print(f"  whole tile loop           : {a[:,2].mean()/tiles:9.0f} cycles per tile (incl. trunk: 112 MFMAs = 7168 cycles)")
print(f"  shader clock during the loop: {a[:,2].mean()/a[:,3].mean()*100:.0f} MHz   loop wall time {a[:,3].mean()/100:.1f} us")
