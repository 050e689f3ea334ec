#!/usr/bin/env python
"""Development tool: instruction mix of the loop bodies of selected kernels in the gfx950 ISA (no GPU needed).

    python tools/isa_loop_mix.py [kernel-name-substring ...]     (default: the KDE backward kernels bench.py books)

Compiles mentflow_amd/csrc/kde.hip (or the file given by ISA_SOURCE) to assembly with hipcc --cuda-device-only -S, splits
every matching kernel into basic blocks and prints, for the blocks that belong to loops, the number of full-rate vector,
transcendental, LDS, vector-memory and scalar instructions.  bench.py's KDE_BWD_MIX are the innermost per-(particle,
projection) blocks of proj_kde1d_bwd_kernel<4, 256> and proj_kde2d_bwd_kernel<4, 1024, 4>."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TRANS = ("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")


def mix(ins):
    c = collections.Counter()
    for ln in ins:
        s = ln.split(";")[0].strip()
        if not s or s.startswith("."):
            continue
        op = s.split()[0]
        if op.startswith(TRANS):
            c["trans"] += 1
        elif op.startswith("v_mfma"):
            c["mfma"] += 1
        elif op.startswith("v_"):
            c["valu"] += 1
        elif op.startswith("ds_"):
            c["lds"] += 1
        elif op.startswith(("global_", "buffer_", "scratch_", "flat_")):
            c["vmem"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
    return dict(c)


def main():
    src = os.environ.get("ISA_SOURCE", os.path.join(ROOT, "mentflow_amd", "csrc", "kde.hip"))
    flags = os.environ.get("ISA_FLAGS", "").split()
    wanted = sys.argv[1:] or ["proj_kde1d_bwd_kernelILi4ELi256E", "proj_kde2d_bwd_kernelILi4ELi1024ELi4E"]
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", *flags, "--cuda-device-only",
                        "-S", src, "-o", out], check=True, stderr=subprocess.DEVNULL)
        lines = open(out).read().split("\n")
    print(f"# {os.path.relpath(src, ROOT)} {' '.join(flags)} -> gfx950 ISA; loop blocks with more than 20 instructions")
    for want in wanted:
        body, on, name = [], False, None
        for ln in lines:
            m = re.match(r"^(_Z\S*" + re.escape(want) + r"\S*):", ln)
            if m and not on:
                on, name = True, m.group(1)
                continue
            if on and ln.startswith(".Lfunc_end"):
                break
            if on:
                body.append(ln)
        if not name:
            print(f"{want}: not found")
            continue
        print(name)
        blk, cur = None, []
        blocks = []
        for ln in body:
            m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", ln)
            if m:
                if blk:
                    blocks.append((blk, cur))
                blk, cur = m.group(1) + " " + (m.group(2) or ""), []
            else:
                cur.append(ln)
        if blk:
            blocks.append((blk, cur))
        for blk, ins in blocks:
            if "Loop" in blk and len(ins) > 20:
                print(f"    {blk[:64]:64s} {mix(ins)}")


if __name__ == "__main__":
    main()
