"""SURVEY.md §5 "race detection / sanitizers" — the CPU pass (GPU-side ASan is not available on this pool).

tests/emu/build_sanitize.sh compiles the kernel sources for the host against the fiber emulator with AddressSanitizer
+ UndefinedBehaviorSanitizer (fiber switches annotated, dynamic LDS exactly sized and fenced by PROT_NONE pages) and
links tests/emu/sanitize_main.cpp, which calls every compute entry point of include/mentflow_hip.h on small synthetic
inputs (both backward variants, every window variant of the KDE kernels, NaN / inf / out-of-range rows).  Any
out-of-range LDS or global index, or undefined arithmetic, aborts the program."""
import os
import subprocess

import pytest

from conftest import EMU_DIR, ROOT

BIN = os.path.join(EMU_DIR, "sanitize_emu")
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:detect_stack_use_after_return=0")


def _up_to_date():
    if not os.path.exists(BIN):
        return False
    t = os.path.getmtime(BIN)
    srcs = [os.path.join(ROOT, "mentflow_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "mentflow_amd", "csrc"))
            if f.endswith((".hip", ".h", ".inc"))]
    srcs += [os.path.join(EMU_DIR, f) for f in ("hip_emu.h", "hip_emu.cpp", "sanitize_main.cpp", "build_sanitize.sh")]
    srcs.append(os.path.join(ROOT, "include", "mentflow_hip.h"))
    return all(os.path.getmtime(s) <= t for s in srcs)


@pytest.fixture(scope="module")
def sanitize_binary():
    if not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"):
        pytest.skip("needs the ROCm clang++ with the sanitizer runtimes")
    if not _up_to_date():
        subprocess.run(["bash", os.path.join(EMU_DIR, "build_sanitize.sh")], check=True, capture_output=True, timeout=1500)
    return BIN


def test_every_entry_point_is_clean_under_asan_and_ubsan(sanitize_binary):
    r = subprocess.run([sanitize_binary], env=ENV, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "SANITIZE OK" in r.stdout
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr


def test_lds_guard_catches_an_overflow(sanitize_binary):
    """Self-test of the guard: a kernel writing 64 bytes past its dynamic LDS block must die."""
    r = subprocess.run([sanitize_binary, "--provoke-lds-overflow"], env=ENV, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "NOT caught" not in r.stdout
    assert "AddressSanitizer" in r.stderr
