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


def test_two_gloo_ranks_c5_and_strong_scaling_on_the_emulator(emu_library):
    """(i) workload c5 over 2 ranks: the forward all-reduce carries 100 x 85 x 85 histogram sums (2.9 MB) + the entropy sums;
    (ii) --scaling strong: a fixed global batch divided over the ranks.  Both lines carry every rank's own clock."""
    r = _run(["--gpus", "2", "--workload", "c5", "--per-gpu", "64", "--steps", "1", "--warmup", "0", "--repeats", "1",
              "--no-cpu-baseline", "--meas-samples", "1000", "--test-emulator-lib", emu_library], timeout=1200)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == 2 and j["ranks_seen"] == 2 and j["config"]["global_batch"] == 128
    assert "c5" in j["config"]["workload"] and j["value"] > 0 and j["config"]["final_loss"] == j["config"]["final_loss"]
    pr = j["per_rank_ms_per_step"]
    assert len(pr["ranks"]) == 2 and pr["min"] <= pr["max"] and abs(pr["max"] - j["ms_per_step"]) < 1e-6
    r = _run(["--gpus", "2", "--workload", "c4", "--scaling", "strong", "--global-batch", "256", "--steps", "1", "--warmup", "0",
              "--repeats", "1", "--no-cpu-baseline", "--meas-samples", "2000", "--test-emulator-lib", emu_library])
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["scaling"] == "strong" and j["config"]["global_batch"] == 256 and j["config"]["per_gpu_batch"] == 128
    assert abs(j["value"] - 256 / (j["ms_per_step"] * 1e-3)) < 1e-6 * j["value"]


def test_sigterm_to_the_launcher_stops_its_ranks(emu_library):
    """A driver timeout (SIGTERM to the launcher) must not leave rank processes behind holding their GPUs."""
    import re
    import signal
    import time
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        e.pop(k, None)
    p = subprocess.Popen([sys.executable, BENCH, "--gpus", "2", "--per-gpu", "4096", "--steps", "50", "--warmup", "0",
                          "--repeats", "5", "--no-cpu-baseline", "--meas-samples", "2000", "--test-emulator-lib", emu_library],
                         env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    pids, buf = None, ""
    t0 = time.time()
    while pids is None and time.time() - t0 < 120:
        line = p.stderr.readline()
        buf += line
        m = re.search(r"pids \[(\d+), (\d+)\]", line)
        if m:
            pids = [int(m.group(1)), int(m.group(2))]
    assert pids, buf
    time.sleep(3.0)                                     # let the ranks get going
    p.send_signal(signal.SIGTERM)
    rc = p.wait(timeout=60)
    assert rc != 0
    time.sleep(0.5)
    for pid in pids:
        try:
            os.kill(pid, 0)
            alive = True
        except ProcessLookupError:
            alive = False
        assert not alive, f"rank process {pid} survived the launcher"


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
