"""The Fortran side of the boundary: the iso_c_binding module compiles and links against the library
(CPU), and a Fortran host assembles + solves on the GPU (gpu)."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import REPO, load_golden

FC = shutil.which("amdflang") or "/opt/rocm/bin/amdflang"
PKG = os.path.join(REPO, "eddy_currents_3d_amd")


def build_demo(tmp, prog="ec3d_host_demo"):
    from eddy_currents_3d_amd import build
    build.build()
    exe = os.path.join(tmp, prog)
    subprocess.run([FC, "-c", os.path.join(PKG, "fortran", "ec3d_hip_mod.f90"), "-o", os.path.join(tmp, "mod.o")],
                   check=True, cwd=tmp)
    subprocess.run([FC, "-c", os.path.join(REPO, "examples", prog + ".f90"), "-o",
                    os.path.join(tmp, "demo.o")], check=True, cwd=tmp)
    subprocess.run([FC, os.path.join(tmp, "mod.o"), os.path.join(tmp, "demo.o"), f"-L{PKG}", "-lec3d_hip",
                    f"-Wl,-rpath,{PKG}", "-o", exe], check=True, cwd=tmp)
    return exe


@pytest.mark.skipif(not os.path.exists(FC), reason="no Fortran compiler")
@pytest.mark.parametrize("prog", ["ec3d_host_demo", "ec3d_timeloop_demo", "ec3d_multi_demo"])
def test_fortran_module_compiles_and_links(tmp_path, prog):
    exe = build_demo(str(tmp_path), prog)
    ldd = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    assert "libec3d_hip.so" in ldd


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(FC), reason="no Fortran compiler")
def test_fortran_host_assembles_and_solves(tmp_path):
    exe = build_demo(str(tmp_path))
    g = load_golden("g2_conducting_hole_16x15x14")
    sdz, sdy, sdx = g["geoPHYS"].shape
    n = len(g["irow"]) - 1
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as f:
        np.array([sdx, sdy, sdz, g["valPHYS"].shape[0], int(g["itmax"])], np.int32).tofile(f)
        np.array([float(g["dt"]), float(g["tol"])], np.float64).tofile(f)
        np.asarray(g["delta"], np.float64).tofile(f)
        np.ascontiguousarray(np.asarray(g["BND"], np.float64).T).tofile(f)      # column-major (3,2)
        np.ascontiguousarray(g["geoPHYS"], np.int8).tofile(f)
        np.ascontiguousarray(g["geoPHYS_C"], np.int32).tofile(f)
        np.ascontiguousarray(np.asarray(g["valPHYS"], np.float64).T).tofile(f)  # column-major
        np.array([n], np.int32).tofile(f)
        g["b0"].tofile(f)
        g["xin0"].tofile(f)
    subprocess.run([exe, fin, fout], check=True)
    with open(fout, "rb") as f:
        it = int(np.fromfile(f, np.int32, 1)[0])
        x = np.fromfile(f, np.float64, n)
    assert it == int(g["iters"][0])
    assert np.linalg.norm(x - g["xout0"]) <= 10 * float(g["tol"]) * np.linalg.norm(g["xout0"])


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(FC), reason="no Fortran compiler")
@pytest.mark.parametrize("devices", ["0,0", "0,0,0"])
def test_fortran_host_on_several_slabs(tmp_path, devices):
    """examples/ec3d_multi_demo.f90: the same Fortran host through ec3d_multi_* -- global tables in, whole vectors
    back, the reference's iteration count and solution."""
    exe = build_demo(str(tmp_path), "ec3d_multi_demo")
    g = load_golden("g2_conducting_hole_16x15x14")
    sdz, sdy, sdx = g["geoPHYS"].shape
    n = len(g["irow"]) - 1
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as f:
        np.array([sdx, sdy, sdz, g["valPHYS"].shape[0], int(g["itmax"])], np.int32).tofile(f)
        np.array([float(g["dt"]), float(g["tol"])], np.float64).tofile(f)
        np.asarray(g["delta"], np.float64).tofile(f)
        np.ascontiguousarray(np.asarray(g["BND"], np.float64).T).tofile(f)
        np.ascontiguousarray(g["geoPHYS"], np.int8).tofile(f)
        np.ascontiguousarray(g["geoPHYS_C"], np.int32).tofile(f)
        np.ascontiguousarray(np.asarray(g["valPHYS"], np.float64).T).tofile(f)
        np.array([n], np.int32).tofile(f)
        g["b0"].tofile(f)
        g["xin0"].tofile(f)
    r = subprocess.run([exe, fin, fout], env=dict(os.environ, EC3D_DEMO_DEVICES=devices, EC3D_MULTI_WATCHDOG="30"),
                       capture_output=True, text=True)
    print(r.stdout.strip())
    assert r.returncode == 0, r.stdout + r.stderr
    with open(fout, "rb") as f:
        it = int(np.fromfile(f, np.int32, 1)[0])
        x = np.fromfile(f, np.float64, n)
    assert it == int(g["iters"][0])
    assert np.linalg.norm(x - g["xout0"]) <= 10 * float(g["tol"]) * np.linalg.norm(g["xout0"])


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(FC), reason="no Fortran compiler")
def test_fortran_time_loop_on_device(tmp_path):
    """A Fortran host runs the reference's time loop with resident fields: rhs_step -> solve_resident ->
    post_update per step, VTK vectors at the end; compared with the reference's captured steps and its
    field_N.vtk (moving-coil case G3)."""
    from test_gpu_timeloop import coil_sources
    from test_vtk_output import parse_vectors
    exe = build_demo(str(tmp_path), "ec3d_timeloop_demo")
    g = load_golden("g3_moving_coil_18x16x12")
    sdz, sdy, sdx = g["geoPHYS"].shape
    ncell = sdx * sdy * sdz
    n = len(g["irow"]) - 1
    nsteps = 3                        # steps 0, 1, 2 -> the VTK vectors correspond to field_2.vtk
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as f:
        np.array([sdx, sdy, sdz, g["valPHYS"].shape[0], int(g["itmax"]), 1, nsteps], np.int32).tofile(f)
        np.array([float(g["dt"]), float(g["tol"])], np.float64).tofile(f)
        np.asarray(g["delta"], np.float64).tofile(f)
        np.ascontiguousarray(np.asarray(g["BND"], np.float64).T).tofile(f)
        np.ascontiguousarray(g["geoPHYS"], np.int8).tofile(f)
        np.ascontiguousarray(g["geoPHYS_C"], np.int32).tofile(f)
        np.ascontiguousarray(np.asarray(g["valPHYS"], np.float64).T).tofile(f)
        np.array([n], np.int32).tofile(f)
        for k in range(nsteps):
            idx, val = coil_sources(g, k, True)
            np.array([len(idx)], np.int32).tofile(f)
            idx.astype(np.int32).tofile(f)
            val.astype(np.float64).tofile(f)
    subprocess.run([exe, fin, fout], check=True)
    tol = float(g["tol"])
    with open(fout, "rb") as f:
        for k in range(nsteps):
            it = int(np.fromfile(f, np.int32, 1)[0])
            x = np.fromfile(f, np.float64, n)
            xr = g[f"xout{k}"]
            assert abs(it - int(g["iters"][k])) <= 2
            assert np.linalg.norm(x - xr) <= 10 * tol * np.linalg.norm(xr)
        vec = [np.fromfile(f, np.float32, 3 * ncell).reshape(ncell, 3) for _ in range(4)]
    ref = parse_vectors(g["vtk_field_2"].tobytes(), ncell)
    for got, name in zip(vec, ["Field_A", "Vector_field_eddy", "Vector_field_SOURCE", "Vector_field_B"]):
        scale = np.abs(ref[name]).max()
        assert np.abs(got - ref[name]).max() <= 20 * tol * scale + 1e-30, name


def build_c_demo(tmp):
    from eddy_currents_3d_amd import build
    build.build()
    exe = os.path.join(tmp, "ec3d_c_demo")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(REPO, "include"),
                    os.path.join(REPO, "examples", "ec3d_c_demo.c"), f"-L{PKG}", "-lec3d_hip", f"-Wl,-rpath,{PKG}",
                    "-lm", "-o", exe], check=True, cwd=tmp)
    return exe


def test_c_host_compiles_and_links(tmp_path):
    exe = build_c_demo(str(tmp_path))
    assert "libec3d_hip.so" in subprocess.run(["ldd", exe], capture_output=True, text=True).stdout


@pytest.mark.gpu
def test_c_host_solves(tmp_path):
    """Plain C against the C ABI: assemble on the device, solve for a known x, check with the library's SpMV."""
    exe = build_c_demo(str(tmp_path))
    r = subprocess.run([exe, "40"], capture_output=True, text=True)
    print(r.stdout.strip())
    assert r.returncode == 0, r.stdout + r.stderr
