"""bench.py as the driver runs it: `python bench.py --gpus N` must start its own N ranks (the parent stays a pure
launcher: no torch / HIP in it), print ONE JSON line from rank 0 carrying the backend and the number of ranks the
collective saw, and exit non-zero when a rank fails.  Here: 2 gloo ranks on the host-emulated kernels (test
infrastructure, tiny batch); the production RCCL run is `test_bench_two_ranks_share_gpu` (-m gpu) + the driver's N-GPU run."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _run(extra, env=None, timeout=600):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + extra, env=e, capture_output=True, text=True, timeout=timeout)


def test_launcher_parent_imports_neither_torch_nor_the_library():
    """The launcher path must run before anything that could initialise the GPU: importing bench and calling the
    launcher leaves torch and mentflow_amd out of sys.modules."""
    code = ("import sys, bench; a = bench.parse_args(['--gpus', '2']); "
            "assert 'torch' not in sys.modules and 'mentflow_amd' not in sys.modules, sorted(sys.modules)[:5]; print('ok')")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr


def test_self_launch_two_gloo_ranks_on_the_emulator(emu_library):
    r = _run(["--gpus", "2", "--workload", "c4", "--per-gpu", "192", "--steps", "1", "--warmup", "0", "--repeats", "2",
              "--no-cpu-baseline", "--meas-samples", "2000", "--test-emulator-lib", emu_library])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["ranks_seen"] == 2 and j["backend"] == "gloo" and j["scaling"] == "weak"
    assert j["config"]["global_batch"] == 384 and j["config"]["per_gpu_batch"] == 192
    assert j["timed_regions"]["count"] == 2 and j["value"] > 0
    assert "EMULATED" in j["data"]                      # a test run can never pass for a measurement


def test_strong_scaling_splits_the_global_batch():
    import bench
    a = bench.parse_args(["--gpus", "8", "--scaling", "strong"])
    assert a.scaling == "strong" and bench.WORKLOADS["c4"]["global_batch"] == 16_777_216
    assert bench.WORKLOADS["c4"]["global_batch"] // 8 == bench.WORKLOADS["c4"]["per_gpu"]


def test_failing_rank_fails_the_launcher(emu_library):
    # a workload flag the workers reject: every rank exits non-zero, the launcher must report it
    r = _run(["--gpus", "2", "--per-gpu", "64", "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
              "--test-emulator-lib", os.path.join(ROOT, "tests", "emu", "does_not_exist.so")])
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


@pytest.mark.gpu
def test_bench_two_ranks_share_gpu():
    """One-GPU box rehearsal of the multi-rank bench: both ranks on cuda:0, collectives over gloo (RCCL refuses two
    ranks per device).  With >= 2 GPUs visible the same command takes the production RCCL path."""
    import torch
    two = torch.cuda.device_count() >= 2
    env = {} if two else {"MENTFLOW_SHARE_GPU": "1"}
    r = _run(["--gpus", "2", "--per-gpu", "65536", "--steps", "2", "--warmup", "1", "--repeats", "1", "--no-cpu-baseline",
              "--meas-samples", "50000"], env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == 2 and j["ranks_seen"] == 2
    assert j["backend"] == ("nccl" if two else "gloo")
    assert j["roofline"]["kernel"].startswith("flow_layer")


@pytest.mark.gpu
def test_bench_graph_mode_small_batch():
    """`--graph --fused-adamw`: the hipGraph-replayed step at the reference's 25 000-particle batch prints a valid line."""
    r = _run(["--workload", "c3", "--per-gpu", "25000", "--steps", "20", "--warmup", "3", "--repeats", "2", "--no-cpu-baseline",
              "--meas-samples", "50000", "--graph", "--fused-adamw"], timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == 1 and "hipGraph" in j["config"]["step"] and j["config"]["global_batch"] == 25000
    assert 0.1 < j["ms_per_step"] < 5.0


def test_torchrun_launched_ranks_on_the_emulator(emu_library):
    """The driver's own N > 1 form: `python -m torch.distributed.run ... bench.py --gpus N` (WORLD_SIZE set by the launcher:
    bench.py must run as a worker, not start ranks of its own)."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), BENCH, "--gpus", "2", "--per-gpu", "128", "--steps", "1",
                        "--warmup", "0", "--repeats", "1", "--no-cpu-baseline", "--meas-samples", "2000", "--scaling", "weak",
                        "--test-emulator-lib", emu_library], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["ranks_seen"] == 2 and j["backend"] == "gloo" and j["config"]["global_batch"] == 256
