"""Lint for one compiler hazard in the gfx950 ISA of the kernels (found in round 4, hipcc of ROCm 7.2).

Divergent control flow is lowered to `s_and_saveexec_b64 ... s_cbranch_execz JOIN ... JOIN: s_or_b64 exec, exec, saved`.  Under
heavy register pressure the register allocator's live-range splitting may put its VGPR / AGPR copies (`v_mov`,
`v_accvgpr_write/read`, spill code) at the TOP of the join block, i.e. in front of the `s_or_b64 exec` that re-enables the lanes
the branch had switched off: those lanes keep the stale register.  Seen in rqs_layer_bwd_fused_kernel<20,3,2>: the bias-gradient
sums of lanes whose particle lies beyond the end of the batch were not carried into the accumulator registers (wrong trunk
bias gradients for ragged batches, every other gradient bit-exact).

The check: in every JOIN block (a label that an `s_cbranch_execz` targets), no vector instruction may stand
between the block label and the `s_or_b64 exec, exec, s[..]` of that block.  (Other blocks that end in an exec restore are the
tails of a masked region — their vector instructions belong inside the region.)  Usage: check_exec_regions.py file.s [...]   (exit status 1 if anything is found)
    hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S x.hip -o x.s
"""
import re
import sys

EXEC_RESTORE = re.compile(r"^\s*s_or_b64\s+exec,\s*exec,\s*(s\[\d+:\d+\])")
SAVEEXEC = re.compile(r"^\s*s_(?:and|or|andn2|xor)_saveexec_b64\s+(s\[\d+:\d+\])")
EXEC_BRANCH = re.compile(r"^\s*s_cbranch_execz\s+([.\w$]+)")   # execnz targets are out-of-line region BODIES
LABEL = re.compile(r"^([.\w$]+):")
VECTOR = re.compile(r"^\s*(v_|ds_|global_|buffer_|scratch_|flat_|image_)")


def scan(path):
    """[(function, line, instruction, line of the restore)].  A join block is suspicious only if its restore re-enables the mask
    saved by the very `s_*_saveexec` whose branch targets it: when the compiler merges the end of an inner region into the
    restore of the outer one, the code between the inner join label and that restore legitimately runs under the outer mask."""
    lines = [ln.split(";")[0].rstrip() for ln in open(path)]
    joins = {}                          # label -> set of saved-mask registers of the regions that branch to it
    last_save = None
    for code in lines:
        m = SAVEEXEC.match(code)
        if m:
            last_save = m.group(1)
        m = EXEC_BRANCH.match(code)
        if m and last_save:
            joins.setdefault(m.group(1), set()).add(last_save)
    problems, func, saved, pending = [], None, None, []
    for ln, code in enumerate(lines, 1):
        if not code.strip():
            continue
        m = LABEL.match(code)
        if m:
            if not m.group(1).startswith(".L"):
                func = m.group(1)
            saved, pending = joins.get(m.group(1)), []
            continue
        if code.strip().startswith(".") or saved is None:
            continue
        m = EXEC_RESTORE.match(code)
        if m:
            if m.group(1) in saved:
                problems += [(func, pl, ptxt, ln) for pl, ptxt in pending]
            saved, pending = None, []
        elif VECTOR.match(code) and not code.strip().startswith(("v_writelane", "v_readlane", "v_readfirstlane")):
            pending.append((ln, code.strip()))        # (lane-indexed moves ignore EXEC: SGPR spills to VGPR lanes are safe there)
        elif code.strip().startswith(("s_cbranch", "s_branch", "s_endpgm", "s_barrier", "s_setpc")) or SAVEEXEC.match(code):
            saved, pending = None, []   # end of the prologue window
    return problems


def main(paths):
    bad = 0
    for p in paths:
        for func, ln, txt, at in scan(p):
            print(f"{p}:{ln}: `{txt}` executes before the exec restore at line {at} ({func})")
            bad += 1
    print(f"{bad} vector instruction(s) in front of the exec restore of their own region ({len(paths)} file(s))")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
