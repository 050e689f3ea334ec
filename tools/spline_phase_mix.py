#!/usr/bin/env python
"""Development tool: instruction mix per PHASE of the spline (rqs_apply) in the gfx950 ISA of the forward kernel and of the
fused backward (no GPU needed).  The two translation units are compiled with -gline-tables-only, which leaves the code as
shipped and tags every instruction with the source line it came from (.loc); the MF_PHASE("...") markers in rqs_apply
(no-ops in every build) give the line ranges of the phases.  Counted per inlined copy of the spline: full-rate vector,
compare / select, moves (incl. v_permlane / v_accvgpr), transcendental, LDS, s_nop.  Issue-cost model of the guide: 4 cycles per
vector instruction for a lone wave (8 transcendental).  Usage: python tools/spline_phase_mix.py > profiles/r04_spline_phase_mix.txt"""
import collections
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "mentflow_amd", "csrc")
TRANS = ("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt")
JOBS = [("forward  rqs_layer_fwd_kernel<20,3,1024> (MODE 0)", "flow_fwd.hip", [], "rqs_layer_fwd_kernelILi20ELi3ELi1024E", 1),
        ("backward rqs_layer_bwd_fused_kernel<20,3,2> (MODE 1: forward part + adjoint)", "flow_bwd_fused.hip", ["-DMF_FUSED_SAVED=2"],
         "rqs_layer_bwd_fused_kernelILi20ELi3ELi2E", 2)]


def classify(op):
    if op.startswith(TRANS):
        return "trans"
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith(("v_cmp", "v_cndmask")):
        return "cmp/sel"
    if op.startswith(("v_permlane", "v_readlane", "v_writelane", "v_mov", "v_accvgpr")):
        return "move"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_"):
        return "salu"
    return "other"


def phase_lines():
    """[(phase, first line, last line)] of rqs_apply in flow_kernels.inc, from the MF_PHASE markers."""
    src = open(os.path.join(CS, "flow_kernels.inc")).read().split("\n")
    marks = [(i + 1, re.search(r'MF_PHASE\("(\w+)"\)', ln).group(1)) for i, ln in enumerate(src) if re.search(r'^\s*MF_PHASE\("', ln)]
    out = []
    for (l0, name), (l1, _) in zip(marks, marks[1:]):
        if name != "end":
            out.append((name, l0, l1 - 1))
    # helpers inlined into the spline carry their OWN line numbers (a line table has no call sites): listed as rows of their own
    helpers = {"half_pair": "helper: lane-half exchanges (v_permlane32_swap)", "cvt_u32_sat": "helper: float -> fixed point",
               "select_pair": "helper: select trees (knots, derivative logits)", "soft_clip": "helper: soft clip (derivatives)",
               "fast_exp": "helper: exp (derivatives)", "soft_clip_grad": "helper: soft clip gradient"}
    for i, ln in enumerate(src):
        m = re.match(r"^__device__ __forceinline__ \w[\w ]* (\w+)\(", ln) or re.match(r"^__device__ __forceinline__ void (\w+)\(", ln)
        if m and m.group(1) in helpers:
            j = i
            while j < len(src) and src[j] != "}" and not (j > i and src[j].startswith("}")):
                j += 1
            out.append((helpers[m.group(1)], i + 1, j + 1))
    return out


def main():
    phases = phase_lines()
    with tempfile.TemporaryDirectory() as tmp:
        for title, src, flags, key, copies in JOBS:
            out = os.path.join(tmp, "k.s")
            subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-gline-tables-only", *flags,
                            "--cuda-device-only", "-S", os.path.join(CS, src), "-o", out], check=True, stderr=subprocess.DEVNULL)
            lines = open(out).read().split("\n")
            inc_ids = {m.group(1) for ln in lines for m in [re.match(r'\s*\.file\s+(\d+)\s+.*"flow_kernels\.inc"', ln)] if m}
            counts = collections.OrderedDict((ph, collections.Counter()) for ph, _, _ in phases)
            on, cur = False, None
            for ln in lines:
                if re.match(r"^_Z\S*" + key + r"\S*:", ln):
                    on = True
                    continue
                if on and ln.startswith(".Lfunc_end"):
                    break
                if not on:
                    continue
                m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", ln)
                if m:
                    cur = None
                    if m.group(1) in inc_ids:
                        L = int(m.group(2))
                        for ph, l0, l1 in phases:
                            if l0 <= L <= l1:
                                cur = ph
                    continue
                s_ = ln.split(";")[0].strip()
                if cur is None or not s_ or s_.startswith(".") or s_.endswith(":"):
                    continue
                counts[cur][classify(s_.split()[0])] += 1
            print(f"{title}   [{copies} inlined cop{'y' if copies == 1 else 'ies'} of the spline: counts divided by {copies}]")
            tot = collections.Counter()
            for ph in counts:
                c = collections.Counter({k: v / copies for k, v in counts[ph].items()})
                vec = c["valu"] + c["cmp/sel"] + c["move"]
                if vec + c["trans"] == 0:
                    continue
                tot.update(c)
                print(f"    {ph[:52]:52s} vector {vec:6.1f} (cmp/select {c['cmp/sel']:5.1f}, moves {c['move']:5.1f})  transcendental {c['trans']:4.1f}"
                      f"  lds {c['lds']:4.1f}  s_nop {c['nop']:4.1f}   ~{4 * vec + 8 * c['trans'] + 4 * c['nop']:6.0f} issue cycles (lone wave)")
            vec = tot["valu"] + tot["cmp/sel"] + tot["move"]
            print(f"    {'TOTAL (v_rcp of fast_rcp, common.h, not included)':52s} vector {vec:6.1f}  transcendental {tot['trans']:4.1f}  lds {tot['lds']:4.1f}  s_nop {tot['nop']:4.1f}"
                  f"   ~{4 * vec + 8 * tot['trans'] + 4 * tot['nop']:6.0f} issue cycles per feature and wave\n")


if __name__ == "__main__":
    main()
