"""The whole run of a shipped input without the Fortran program: eddy_currents_3d_amd/host.py (palette source
language, function evaluation, source motion, time loop) driving the device-resident loop, against what the
unmodified reference produced for the same file (tests/golden/g4_*: per-step iter, ||b||, ||x|| and 200 probes
of b and x at the solver call, first three steps)."""
import os

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", ["compare_to_Elmer", "ec_src_move_hole", "LIM"])
def test_shipped_input_runs_like_the_reference(case, tmp_path):
    import eddy_currents_3d_amd as E
    from eddy_currents_3d_amd import host, vxc
    g = load_golden("g4_" + case)
    model = vxc.VxcModel(g["vox"], [str(s) for s in g["names"]], float(str(g["lattice_dim"])),
                         tuple(float(x) for x in g["adj"]))
    probes = g["probes"]
    tol = float(g["tol"])
    seen = []

    def on_rhs(k, s, info):
        b = s.download("B")
        info["bnorm"], info["bprobe"] = float(np.linalg.norm(b)), b[probes]

    def on_solved(k, s, info):
        x = s.download("X")
        info["xnorm"], info["xprobe"] = float(np.linalg.norm(x)), x[probes]
        seen.append(info)

    with E.EC3DSolver() as s:
        log = host.run(model, s, steps=3, out_dir=str(tmp_path), on_rhs=on_rhs, on_solved=on_solved)
        assert s.n == int(g["n"]) and s.info.nnz == int(g["nnz"])
    assert len(log) == 3 and [i.get("output") for i in log] == [None, 1, 2]
    assert sorted(os.listdir(tmp_path)) == ["field_1.vtk", "field_2.vtk", "src_1.vtk", "src_2.vtk"]
    for k, info in enumerate(seen):
        it_ref = int(g["iters"][k])
        print(f"{case} step {k}: iter {info['iter']} / reference {it_ref}; ||b|| {info['bnorm']:.9e} / "
              f"{float(g['bnorm'][k]):.9e}; ||x|| {info['xnorm']:.6e} / {float(g['xnorm'][k]):.6e}")
        # step 0 has no history: the right-hand side is the sources alone and matches to rounding; later
        # steps carry the previous solutions (each within the solver tolerance of the reference's)
        assert info["bnorm"] == pytest.approx(float(g["bnorm"][k]), rel=1e-14 if k == 0 else 10 * tol)
        assert np.abs(info["bprobe"] - g["bprobe"][k]).max() <= (1e-14 if k == 0 else 10 * tol) * np.abs(g["bprobe"][k]).max()
        assert info["xnorm"] == pytest.approx(float(g["xnorm"][k]), rel=10 * tol)
        assert np.abs(info["xprobe"] - g["xprobe"][k]).max() <= 10 * tol * np.abs(g["xprobe"][k]).max()
        # 0.8 M / 0.4 M unknowns at tol 5e-3: the reference's own counts (173/160/80, 288/108/98, 70/36/36)
        assert info["iter"] == it_ref


@pytest.mark.parametrize("slabs", [1, 3], ids=["one-handle", "three-slabs"])
def test_overlapped_output_writes_the_same_files(tmp_path, slabs):
    """Field output beside the next step's solve (host._OutputPipeline over ec3d_vtk_fields_begin / _wait: field
    kernel behind the post-update, bytes swapped on the device, pinned double buffer, a host thread that writes) must
    leave exactly the files the synchronous path writes -- the moving-coil LIM input, 6 steps, 5 outputs, so both
    pinned buffers are reused while the loop runs ahead.  On three slabs behind one multi handle
    (ec3d_multi_vtk_fields_begin / _wait) every slab copies its own part and the writer strings them together."""
    import filecmp
    import eddy_currents_3d_amd as E
    from eddy_currents_3d_amd import host, vxc
    from eddy_currents_3d_amd.vtk import join_parts
    g = load_golden("g4_LIM")
    model = vxc.VxcModel(g["vox"], [str(s) for s in g["names"]], float(str(g["lattice_dim"])),
                         tuple(float(x) for x in g["adj"]))
    logs = {}
    for mode in (True, False):
        d = tmp_path / ("overlap" if mode else "sync")
        seen = []
        with (E.EC3DSolver() if slabs == 1 else E.EC3DMulti(slabs, devices=[0] * slabs)) as s:
            logs[mode] = host.run(model, s, steps=6, out_dir=str(d), overlap_output=mode,
                                  on_fields=lambda N, f, info: seen.append(
                                      (N, float(np.abs(join_parts(f)["A"].astype(np.float32)).max()))))
        # (overlapped: three writer threads, so the callbacks may arrive out of order)
        assert sorted(n for n, _ in seen) == [1, 2, 3, 4, 5] and all(a > 0 for _, a in seen)
    assert [i["iter"] for i in logs[True]] == [i["iter"] for i in logs[False]]
    names = sorted(os.listdir(tmp_path / "sync"))
    assert names == sorted(f"{k}_{n}.vtk" for k in ("field", "src") for n in range(1, 6))
    match, mismatch, errors = filecmp.cmpfiles(tmp_path / "sync", tmp_path / "overlap", names, shallow=False)
    assert sorted(match) == names and not mismatch and not errors


def test_command_line_runs_a_vxc_file(tmp_path, capsys):
    """python -m eddy_currents_3d_amd.run: a ZLIB-compressed .vxc file in, the reference's output files out."""
    from eddy_currents_3d_amd import run, vxc
    g = load_golden("g4_compare_to_Elmer")
    model = vxc.VxcModel(g["vox"], [str(s) for s in g["names"]], float(str(g["lattice_dim"])),
                         tuple(float(x) for x in g["adj"]))
    path = str(tmp_path / "model.vxc")
    vxc.write_vxc(path, model, compression="ZLIB")
    out = str(tmp_path / "vec")
    assert run.main([path, "--steps", "3", "--out", out]) == 0
    text = capsys.readouterr().out
    assert "iter=173" in text and "iter=160" in text and "iter=80" in text     # the reference's counts
    assert sorted(os.listdir(out)) == ["field_1.vtk", "field_2.vtk", "src_1.vtk", "src_2.vtk"]
