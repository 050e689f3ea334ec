"""Static check of the gfx950 ISA hipcc generates for every kernel (no GPU needed: hipcc cross-compiles).

tools/check_exec_regions.py looks for one compiler hazard met in round 4: register-allocator copies placed at the top of a
divergent region's join block, in front of the `s_or_b64 exec` that re-enables the masked-off lanes (those lanes keep a stale
register — the trunk bias gradients of ragged batches were wrong in one instance of the fused backward while every parity test
on the emulator passed, because the emulator compiles the same source with another back end).  The check is cheap, so it runs over
every translation unit of the library at every build."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_exec_regions  # noqa: E402

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _sources():
    import __graft_entry__ as g
    return g.sources()


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_no_vector_instruction_in_front_of_an_exec_restore(tmp_path):
    csrc = os.path.join(ROOT, "mentflow_amd", "csrc")
    procs = []
    for name, src, flags in _sources():
        out = str(tmp_path / (name + ".s"))
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", *flags, "--cuda-device-only", "-S",
               os.path.join(csrc, src), "-o", out]
        procs.append((out, subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)))
    problems = []
    for out, p in procs:
        _, err = p.communicate()
        assert p.returncode == 0, err[-2000:]
        problems += [(os.path.basename(out),) + pr for pr in check_exec_regions.scan(out)]
    assert not problems, "\n".join(f"{f}:{ln}: `{txt}` before the exec restore at line {at} ({fn})" for f, fn, ln, txt, at in problems[:20])


def test_the_lint_catches_the_pattern(tmp_path):
    """The ISA shape that was miscompiled, reduced: copies between the join label and the restore of the region's own mask."""
    bad = tmp_path / "bad.s"
    bad.write_text("""
kernel:
	s_and_saveexec_b64 s[16:17], s[20:21]
	s_cbranch_execz .LBB0_89
	global_load_dword a91, v[8:9], off offset:20
.LBB0_89:
	v_accvgpr_write_b32 a194, v198
	s_mov_b32 s89, s30
	s_or_b64 exec, exec, s[16:17]
	s_endpgm
""")
    good = tmp_path / "good.s"
    good.write_text("""
kernel:
	s_and_saveexec_b64 s[16:17], s[20:21]
	s_cbranch_execz .LBB0_89
	global_load_dword a91, v[8:9], off offset:20
.LBB0_89:
	s_mov_b32 s89, s30
	v_writelane_b32 v42, s31, 8
	s_or_b64 exec, exec, s[16:17]
	v_accvgpr_write_b32 a194, v198
	s_endpgm
""")
    assert len(check_exec_regions.scan(str(bad))) == 1
    assert check_exec_regions.scan(str(good)) == []
