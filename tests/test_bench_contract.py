"""bench.py's contract with the driver: one JSON line with the agreed fields; `--gpus N` from a bare `python bench.py`
starts by itself (in-library multi-GPU) and asks for the devices it needs before touching a GPU."""
import json
import os
import subprocess
import sys

import pytest

from conftest import REPO

BENCH = os.path.join(REPO, "bench.py")
REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline"]


def _run(args, timeout=600, env=None):
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=timeout,
                          env=dict(os.environ, **(env or {})))


def test_more_gpus_than_the_machine_has_is_said_before_any_gpu_call():
    import torch
    have = torch.cuda.device_count()
    want = max(have, 1) + 1                      # at least 2: one GPU is not the multi-GPU path
    r = _run(["--gpus", str(want), "--steps", "2", "--warmup", "1"])
    assert r.returncode != 0
    assert f"needs {want} devices, this machine has {have}" in (r.stdout + r.stderr)
    assert "torch.distributed.run" not in (r.stdout + r.stderr)


@pytest.mark.gpu
@pytest.mark.parametrize("args,ngpu", [(["--grid", "64", "--no-cpu-baseline"], 1),
                                       (["--grid", "64", "--gpus", "2", "--devices", "0,0"], 2),
                                       (["--grid", "64", "--force-dist", "--no-cpu-baseline"], 1)])
def test_one_json_line_with_the_agreed_fields(args, ngpu):
    r = _run(args + ["--steps", "4", "--warmup", "2"], env={"EC3D_MULTI_WATCHDOG": "30"})
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == ngpu and d["steps"] == 4 and d["warmup"] == 2 and d["dtype"] == "f64"
    assert d["unit"] == "DOF*iters/s" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["value"] == pytest.approx(64 ** 3 * 4 / (d["ms_per_step"] * 4e-3), rel=1e-6)
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and rf["unit"] == "GB/s"
    assert rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"])
    assert "traffic" in rf
    assert rf["byte_model"] == "format"                     # dictionary storage: NOT SURVEY 8d's plain-DIA byte model
    if ngpu == 1 and "--force-dist" not in args:           # the north-star SpMV figure travels in the driver-run line
        sd = d["spmv_dia"]
        assert sd["bytes_per_row"] == 72 and sd["ms"] > 0 and sd["frac"] == pytest.approx(sd["GBps"] / 8000.0)
        it = d["iter_dia"]                                  # ... and the whole iteration on SURVEY 8d's exact model
        assert it["byte_model"] == "survey_8d" and it["bytes_per_dof_iter"] == 264 and it["x_update_every"] == 1
        assert it["fusion"] == [0, 0] and set(it["kernels"]) == {"k1", "k2", "k3", "k4", "k5"}
        assert it["frac"] == pytest.approx(264 * 64 ** 3 / (it["ms_per_step"] * 1e-3) / 8e12, rel=1e-6)
    if "--force-dist" in args:                              # one process per GPU: the C++ loop over RCCL says what it costs
        assert d["host"]["enqueue_ms_per_iteration"] > 0 and d["host"]["api_calls_per_iteration"] > 5
    if "--devices" in args or "--force-dist" in args:      # both multi-GPU paths check their transport before timing
        assert "bit for bit" in d["verified"]


@pytest.mark.gpu
def test_headline_line_carries_the_av_and_256_sub_records():
    """The driver's command at the headline size (fewer steps): the 512^3 line with `roofline`, the plain-DIA SpMV figure
    with the placement probe's account, and the two sub-records timed after the headline region on fresh handles -- the
    reference's own A-V system (src/EC3D.f90:408) at 306 x 306 x 72 and the 256^3 cube of BASELINE config 2."""
    r = _run(["--steps", "6", "--warmup", "2", "--no-cpu-baseline"])
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["config"]["grid"] == [512, 512, 512] and "512^3" in d["config"]["workload"]
    assert set(d["kernels"]) == {"k3", "k4", "k5"}                      # three launches, asked of the library
    # K4 runs as an SpMV kernel that computes A*S again (25 B per row in every iteration); the X updates of four iterations
    # are applied by a launch of their own on the same stream (k_x_group: X, 4 P, 4 S read, X written = 80 B per row) that is
    # timed with the K4 of the group's last iteration (ec3d_get_k4_form, ec3d_get_x_interval, ec3d_get_x_groups): the line
    # says so and states what the launches move next to SURVEY 8d's 56 B
    assert d["config"]["x_update_every"] == 4 and d["config"]["k4_as_spmv"] is True and "k_x_group" in d["config"]["x_groups"]
    assert d["kernels"]["k4"]["bytes_per_row"] == 45.0 and d["kernels"]["k4"]["survey_bytes_per_row"] == 56
    assert d["kernels"]["k3"]["bytes_per_row"] == 25 and d["config"]["bytes_per_dof_iter"]["this_format"] == 119.0
    vp = d["config"]["vector_placement"]    # the work vectors' placement search ran at set-up and says what it saw and cost
    assert 2 <= len(vp["candidate_us_per_iteration"]) <= 6 and 0 <= vp["kept"] < len(vp["candidate_us_per_iteration"])
    assert vp["search_ms"] < 20000      # (0.1-0.2 s as a rule; a hipMalloc now and then takes seconds)
    pl = d["spmv_dia"]["placement"]
    assert 1 <= len(pl["candidate_us"]) <= 8 and 0 <= pl["kept"] < len(pl["candidate_us"])
    it = d["iter_dia"]         # SURVEY 8d's 264 B per DOF*iter, plain DIA, five launches, driver-timed (200 iterations)
    assert it["steps"] == 200 and it["x_update_every"] == 1 and it["fusion"] == [0, 0]
    assert 0.55 < it["frac"] < 0.9 and it["frac"] == pytest.approx(264 * 512 ** 3 / (it["ms_per_step"] * 1e-3) / 8e12, rel=1e-6)
    assert d["roofline"]["byte_model"] == "format" 
    for name, n, lo in (("av", 21391776, 2.0e10), ("cube256", 256 ** 3, 2.5e10)):
        s = d[name]
        assert "error" not in s, s
        assert s["n"] == n and s["unit"] == "DOF*iters/s" and s["value"] > lo
        assert s["value"] == pytest.approx(n * s["steps"] / (s["ms_per_step"] * s["steps"] * 1e-3), rel=1e-6)
        assert set(s["kernels"]) == {"k1", "k2", "k3", "k4", "k5"}        # below 32 Mi rows: five launches
        dom = s["dominant"]
        assert dom["frac"] == pytest.approx(dom["achieved"] / 8000.0) and 0.3 < dom["frac"] < 1.2
