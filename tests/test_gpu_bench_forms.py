"""Every form bench.py can be started in, as the driver and a maintainer start it, on a small grid.

Round 5's two mid-round bench failures (a wrong probe in the 2-slab check, a plan missing from a description table) were
found by hand-run bench.py calls, not by the suite; the first multi-GPU SCALE run executes exactly these code paths.  Each
form runs as a subprocess at --grid 128 --steps 5 and must print ONE JSON line with the keys the driver reads; the N > 1
forms must say which transport carried the line, what the other one did, and that A*x / ||b|| were checked before timing.
"""
import json
import os
import subprocess
import sys

import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu

BENCH = os.path.join(REPO, "bench.py")
COMMON = ["--grid", "128", "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-side-workloads"]
DRIVER_KEYS = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
               "dtype", "data", "config", "roofline"]


def run_form(argv, launcher=False, env=None):
    cmd = [sys.executable]
    if launcher:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                "--master-port", "29533"]
    r = subprocess.run(cmd + [BENCH] + argv + COMMON, capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, EC3D_MULTI_WATCHDOG="30", **(env or {})))
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    for k in DRIVER_KEYS:
        assert k in d, k
    assert d["steps"] == 5 and d["dtype"] == "f64" and d["unit"] == "DOF*iters/s" and d["value"] > 0
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and 0 < rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"])
    return d


def test_single_gpu_default_format():
    d = run_form([])
    assert d["n_gpus"] == 1 and d["value"] == pytest.approx(128 ** 3 * 5 / (d["ms_per_step"] * 5e-3), rel=1e-6)
    assert d["spmv_dia"]["bytes_per_row"] == 72 and d["iter_dia"]["bytes_per_dof_iter"] == 264


def test_single_gpu_plain_dia():
    d = run_form(["--format", "dia"])
    assert d["n_gpus"] == 1 and d["roofline"]["byte_model"] == "survey_8d" and d["config"]["band_format"] == "plain DIA"


def test_single_gpu_av_system():
    d = run_form(["--workload", "av", "--refine", "1"])
    assert d["n_gpus"] == 1 and "structured A-V form" in d["config"]["band_format"] and d["config"]["n"] == 792288


def test_launcher_form_with_one_rank():
    d = run_form([], launcher=True)
    assert d["n_gpus"] == 1


@pytest.mark.parametrize("extra", [[], ["--workload", "av", "--refine", "1"]], ids=["cube", "av"])
def test_two_slabs_on_one_card_from_the_plain_form(extra):
    """`python bench.py --gpus 2 --devices 0,0`: the parent starts no rank process (RCCL refuses two ranks on one device, and
    says so in the line) and the in-library form, a fresh child, carries the line -- verified before it was timed."""
    d = run_form(["--gpus", "2", "--devices", "0,0"] + extra)
    assert d["n_gpus"] == 2 and d["transport"] == "in_library"
    assert "refuses two ranks" in d["rccl"]["skipped"] or "cube workloads" in d["rccl"]["skipped"]
    assert "bit for bit" in d["verified"] and d["in_library"]["value"] == d["value"]
    assert "z-slab x2 inside the library" in d["config"]["parallelism"]


def test_plain_form_starts_rank_processes_and_the_in_library_child():
    """`python bench.py --gpus 1 --devices 0`: the parent's whole machinery with the one device this box has -- a fresh RCCL
    rank process (a one-rank communicator: RCCL itself counts it) carries the line, a fresh in-library child the sub-record;
    the parent touches no GPU.  With N devices the same code starts N ranks."""
    d = run_form(["--gpus", "1", "--devices", "0"])
    assert d["n_gpus"] == 1 and d["transport"] == "rccl"
    assert d["rccl"]["nranks"] == 1 and len(d["rccl"]["ranks"]) == 1 and d["rccl"]["ranks"][0]["device"] == 0
    il = d["in_library"]
    assert il["value"] > 0 and il["n_gpus"] == 1 and "inside the library" in il["config"]["parallelism"]
    assert il["value"] == pytest.approx(d["value"], rel=0.5)         # the same slab, the same kernels: the same order of magnitude


@pytest.mark.parametrize("rg", ["1,2", "4,8"])
def test_rank_rehearsal(rg):
    """One rank of a G-rank RCCL job alone on this GPU: the line says what RCCL counted, every stage, and how long the compute
    stream stood at the reduction points and the halo waits."""
    d = run_form(["--rehearse", rg])
    assert d["n_gpus"] == 1 and d["transport"] == "rccl" and "REHEARSAL" in d["config"]["workload"]
    rc = d["rccl"]
    assert rc["nranks"] == 1 and rc["nranks_agreed"] and rc["version"] > 0 and "rccl" in os.path.basename(rc["library"]).lower()
    me = rc["ranks"][0]
    assert set(me["stage_us"]) == {"k1", "k2", "k3", "k4", "k5"} and me["ms_per_step"] > 0
    assert me["reduction_points"]["per_iteration"] == 3 and me["reduction_points"]["us_per_iteration"] > 0
    assert me["halo_waits"]["per_iteration"] == 2 and me["host"]["api_calls_per_iteration"] > 10
    assert me["plan"] in (0, 1, 2, 3, 4, 5) and d["host"]["enqueue_ms_per_iteration"] > 0
