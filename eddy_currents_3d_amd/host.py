"""Python host for a whole run: what the reference's main program does around the solve.

The library covers assembly, the per-step right-hand side, the solve, the post-update and the field output
(include/ec3d_hip.h); this module supplies the rest of /root/reference/src/EC3D.f90's time loop for a
``.vxc`` model, so a file the reference runs can be run on the GPU without the Fortran program:

* the palette's source mini-language (src/vxc2data.f90:416-560, :836-890): ``SRCx=F`` / ``SRCy=F`` materials,
  ``FUNC`` definitions with named parameters, source velocities ``Vsx/Vsy/Vsz`` given as numbers or functions;
* the expression evaluator the reference uses for them (src/m_fparser.f90:76-100, :195-240: ``sind cosd tgd
  sh ch th cth lg ln impls impl2 pos int nint floor ceil atg`` ...);
* per step (src/EC3D.f90:241-367): function values at time T (times mu0), motion of the source cells
  (``motion_calc`` / ``new_m``, :1052-1114: accumulated distance in cells, rounded, clamped two cells off the
  box), the list of (unknown id, value) pairs handed to ``ec3d_rhs_step``;
* the loop itself (:137-152, :404-455): solve, post-update, ``field_N.vtk`` every ``jump`` (default: every
  step but the first), ``T += dt`` until ``T >= stop``.

Both output files of the reference are written: ``field_N.vtk`` (fields from the device) and ``src_N.vtk`` (the
source cells as hexahedra).  ``SRCz`` is parsed but ignored: the reference stores its axis as 'D', never matches it against 'Z' again
(src/vxc2data.f90:489, :694) and stops in its time loop (src/EC3D.f90:327) -- there is no behaviour to mirror.
"""
from __future__ import annotations

import ast
import math
import os

import numpy as np

from . import vxc
from .vtk import write_field_vtk, write_src_vtk

MU0 = 0.12566370964050292e-05  # src/EC3D.f90:255


def _nint(x: float) -> int:
    """Fortran NINT: nearest integer, halves away from zero."""
    return int(math.floor(abs(x) + 0.5)) * (1 if x >= 0 else -1)


_FUNCS = {
    "ABS": abs, "EXP": math.exp, "LG": math.log10, "LN": math.log, "SQRT": math.sqrt,
    "SH": math.sinh, "CH": math.cosh, "TH": lambda x: math.sinh(x) / math.cosh(x),
    "CTH": lambda x: math.cosh(x) / math.sinh(x),
    "SIND": lambda x: math.sin(math.radians(x)), "COSD": lambda x: math.cos(math.radians(x)),
    "TGD": lambda x: math.tan(math.radians(x)),
    "SIN": math.sin, "COS": math.cos, "TG": math.tan, "ASIN": math.asin, "ACOS": math.acos,
    "IMPLS": lambda x: 1.0 if x > 0.0 else 0.0, "IMPL2": lambda x: 1.0 if x >= 0.0 else -1.0,
    "POS": lambda x: x if x > 0.0 else 0.0, "INT": lambda x: float(math.trunc(x)),
    "NINT": lambda x: float(_nint(x)), "FLOOR": lambda x: float(math.floor(x)),
    "CEIL": lambda x: float(math.ceil(x)), "ATG": math.atan,
}


class Expression:
    """One FUNC right-hand side (upper-cased, as the reference sees it after ``Upp``)."""

    def __init__(self, text: str):
        self.text = text
        self.tree = ast.parse(text.replace("^", "**"), mode="eval").body

    def __call__(self, variables: dict) -> float:
        def ev(n):
            if isinstance(n, ast.Constant) and isinstance(n.value, (int, float)):
                return float(n.value)
            if isinstance(n, ast.Name):
                return float(variables[n.id])
            if isinstance(n, ast.UnaryOp) and isinstance(n.op, (ast.USub, ast.UAdd)):
                v = ev(n.operand)
                return -v if isinstance(n.op, ast.USub) else v
            if isinstance(n, ast.BinOp):
                a, b = ev(n.left), ev(n.right)
                if isinstance(n.op, ast.Add): return a + b
                if isinstance(n.op, ast.Sub): return a - b
                if isinstance(n.op, ast.Mult): return a * b
                if isinstance(n.op, ast.Div): return a / b
                if isinstance(n.op, ast.Pow): return a ** b
            if isinstance(n, ast.Call) and isinstance(n.func, ast.Name) and n.func.id in _FUNCS and len(n.args) == 1:
                return float(_FUNCS[n.func.id](ev(n.args[0])))
            raise ValueError(f"unsupported construct in source expression {self.text!r}")
        return ev(self.tree)


class _Func:
    def __init__(self, name):
        self.name, self.expr, self.argnames, self.argvals = name, None, [], []

    def value(self, T):
        if self.expr is None:
            raise ValueError(f"source function {self.name} is used but has no FUNC definition")
        v = {a: (T if a == "T" else x) for a, x in zip(self.argnames, self.argvals)}
        return self.expr(v)


class SourceProgram:
    """The independent sources of a model: which unknowns they act on and with what value at time T."""

    def __init__(self, model: vxc.VxcModel, tables: dict):
        vox = model.vox
        self.sdz, self.sdy, self.sdx = vox.shape
        self.ncells = vox.size
        self.delta = np.asarray(tables["delta"], np.float64)
        self.dt = float(tables["dt"])
        nsub = tables["nsub"]
        consts = dict(PI=math.pi, E=0.27182818284590451e+001, MU0=MU0, E0=vxc.E0, DT=self.dt, DX=self.delta[0],
                      DY=self.delta[1], DZ=self.delta[2], TIME=float(tables["time"] or 0.0), NX=self.sdx,
                      NY=self.sdy, NZ=self.sdz)
        self.funs = []    # one per SRC keyword: dict(mat, axis, f, vel=[const or None]*3, mech=[_Func or None]*3, move)
        defs = {}         # name -> _Func (shared by every use of the name, as nameFun / nameVmech matching does)

        def func(name):
            return defs.setdefault(name, _Func(name))

        lines = [nm.replace("=", " ").split() for nm in model.names]
        for kp, w in enumerate(lines, start=1):
            W = [x.upper() for x in w]
            for i in range(1, len(W)):
                if W[i][:1] == "D" and kp <= nsub and not W[i].startswith("DIR"):
                    if i + 2 < len(W) and "SRC" in W[i + 2]:            # src/vxc2data.f90:478
                        for j in range(i + 2, len(W) - 1):
                            ax = {"SRCX": 0, "SRCY": 1}.get(W[j])
                            if ax is None:
                                continue                                # SRCZ: see the module docstring
                            f = dict(mat=kp, axis=ax, f=func(W[j + 1]), vel=[0.0] * 3, mech=[None] * 3, move=[0] * 3)
                            for n in range(1, 7):                       # calcVmech (:844-888): Vs? name/number pairs
                                if j + 1 + n + 1 > len(W) - 1:
                                    continue
                                key = W[j + 1 + n]
                                d = 0 if "VSX" in key else 1 if "VSY" in key else 2 if "VSZ" in key else None
                                if d is None:
                                    continue
                                val = w[j + 1 + n + 1]
                                f["move"][d] = 1
                                if val[:1].upper().isalpha():           # a function name
                                    f["mech"][d] = func(val.upper())
                                else:                                   # a number or a quoted expression
                                    f["vel"][d] = vxc.evaluate(val, consts)
                            self.funs.append(f)
                    break
                if "FUNC" in W[i] and i + 2 < len(W):                   # :497-548
                    fn = func(W[i + 1])
                    fn.expr = Expression(W[i + 2])
                    fn.argnames, fn.argvals = [], []
                    for a, b in zip(W[i + 3::2], w[i + 4::2]):
                        fn.argnames.append(a[:8])
                        fn.argvals.append(0.0 if b.strip("'\"").upper() == "T" else vxc.evaluate(b, consts))
                    break
        flat = vox.reshape(-1)
        for f in self.funs:   # cells of the material, in the order the reference's LIFO list yields them
            f["nodes"] = (np.flatnonzero(flat == f["mat"])[::-1] + 1).astype(np.int64)   # 1-based cell ids
            f["distance"] = np.zeros(3)
            f["shift"] = np.array([f["vel"][d] * self.dt / self.delta[d] if f["move"][d] and f["mech"][d] is None
                                   else 0.0 for d in range(3)])
        self.moving = any(any(f["move"]) for f in self.funs)           # flag_move, src/EC3D.f90:158-187
        self.movestop = [1, 1, 1]

    def _moved(self, f):
        """new_m for all nodes of one function (src/EC3D.f90:1064-1114)."""
        sd = (self.sdx, self.sdy, self.sdz)
        m = f["nodes"] - 1
        pos = [m % self.sdx + 1, (m // self.sdx) % self.sdy + 1, m // (self.sdx * self.sdy) + 1]
        new = []
        for d in (2, 1, 0):   # the reference tests z, then y, then x
            p = pos[d] + int(f["length"][d])
            hi, lo = sd[d] - 2, 2
            clamped = (p > hi) | (p < lo)
            p = np.clip(p, lo, hi)
            # movestop(d) after the reference's node-by-node pass: a clamped node clears it, a later node that
            # is inside the range sets it again
            inside = ~clamped & ((p < hi) | (p > lo))
            if clamped.any():
                last = int(np.flatnonzero(clamped)[-1])
                self.movestop[d] = 1 if inside[last + 1:].any() else 0
            elif self.movestop[d] == 0 and inside.any():
                self.movestop[d] = 1
            new.append(p)
        Lz, jy, ix = new
        return ix + self.sdx * (jy - 1) + self.sdx * self.sdy * (Lz - 1)

    def step(self, T: float):
        """(src_index, src_value, moving) for ec3d_rhs_step at time T; advances the motion state."""
        idx, val = [], []
        self.groups = []   # what src_N.vtk shows: per function (axis, cells of one component, value)
        mech_cache = {}
        for f in self.funs:
            a = f["f"].value(T) * MU0
            nodes = f["nodes"]
            if self.moving:
                for d in range(3):                                      # motion_calc, :1052-1062
                    if f["mech"][d] is None:
                        f["distance"][d] += self.movestop[0] * f["shift"][d]
                    else:
                        fn = f["mech"][d]
                        if fn.name not in mech_cache:
                            mech_cache[fn.name] = fn.value(T)
                        f["distance"][d] += mech_cache[fn.name] * self.dt / self.delta[d]
                f["length"] = [_nint(x) for x in f["distance"]]
                nodes = self._moved(f)
            idx.append(nodes + f["axis"] * self.ncells)
            val.append(np.full(len(nodes), a))
            self.groups.append((f["axis"], nodes, a))
        if not idx:
            return np.zeros(0, np.int32), np.zeros(0), self.moving
        return np.concatenate(idx).astype(np.int32), np.concatenate(val), self.moving


class _OutputPipeline:
    """field_N.vtk / src_N.vtk of output step N written on a host thread while step N+1 is solved.

    The reference writes the files inside its time loop (src/EC3D.f90:436-444) and the solver waits; here
    ``ec3d_vtk_fields_begin`` puts the field kernel behind the post-update on the solver's stream and the copy of
    the four vectors (already in the file's big-endian byte order) into one of two pinned buffers on a side stream,
    the loop goes on, and this thread waits for the copy, hands the views to ``on_fields`` and writes the bytes as
    they are.  As many outputs are in flight as the library has pinned buffers (three; one writer thread each, since
    one write() stream fills the page cache at 3-4 GB/s and a 9 M-cell step is 566 MB): before output N is started,
    output N-3 must have left its buffer.  src_N.vtk needs nothing from the device and goes through threads of its own."""

    def __init__(self, solver, dims, delta, out_dir, on_fields, on_written=None):
        import queue
        import threading
        self.solver, self.dims, self.delta, self.out_dir, self.on_fields = solver, dims, delta, out_dir, on_fields
        self.on_written = on_written
        self.jobs = queue.Queue()
        self.done = []                  # one threading.Event per started output, in order
        self.error = None
        self._Event = threading.Event
        # src_N.vtk needs nothing from the device: it is formatted and written on a thread of its own from the moment
        # the step's sources are known (81 MB of corner coordinates per step on config 5)
        from concurrent.futures import ThreadPoolExecutor
        self.src_pool = ThreadPoolExecutor(max_workers=2, thread_name_prefix="ec3d-src")
        from .solver import VTK_SLOTS
        self.slots = VTK_SLOTS
        self.threads = [threading.Thread(target=self._work, name=f"ec3d-output-{i}", daemon=True)
                        for i in range(self.slots)]
        for th in self.threads:
            th.start()

    def start(self, N, groups, write, info):
        if len(self.done) >= self.slots:
            self.done[-self.slots].wait()   # the buffer this output is going to use is free again
        self._raise()
        slot = self.solver.vtk_fields_begin(self.delta, big_endian=True)
        ev = self._Event()
        self.done.append(ev)
        src = None
        if write and groups:                                                # src/EC3D.f90:446  CALL writeVtk_src
            sdx, sdy, sdz = self.dims
            src = self.src_pool.submit(write_src_vtk, os.path.join(self.out_dir, f"src_{N}.vtk"), sdx, sdy, sdz,
                                       self.delta, groups)
        self.jobs.put((slot, N, src, write, info, ev))

    def _work(self):
        sdx, sdy, sdz = self.dims
        while True:
            job = self.jobs.get()
            if job is None:
                return
            slot, N, src, write, info, ev = job
            try:
                f = self.solver.vtk_fields_wait(slot, big_endian=True)      # views of the pinned buffer
                if self.on_fields is not None:
                    self.on_fields(N, f, info)
                if write:
                    paths = [os.path.join(self.out_dir, f"field_{N}.vtk")]
                    write_field_vtk(paths[0], sdx, sdy, sdz, self.delta, f)
                    if src is not None:
                        src.result()
                        paths.append(os.path.join(self.out_dir, f"src_{N}.vtk"))
                    if self.on_written is not None:
                        self.on_written(N, paths)
            except BaseException as e:  # reported by the loop's thread at its next output, or at the end
                self.error = e
            finally:
                ev.set()

    def _raise(self):
        if self.error is not None:
            e, self.error = self.error, None
            raise e

    def finish(self):
        for _ in self.threads:
            self.jobs.put(None)
        for th in self.threads:
            th.join()
        self.src_pool.shutdown(wait=True)
        self._raise()


class _SourcesAhead:
    """The source program of the NEXT time steps evaluated while the GPU solves this one.  Sources and their motion
    are functions of time alone (src/EC3D.f90:245-340 never reads the solution), so step k + 1's (unknown id, value)
    list -- 317 088 moving cells on config 5, 18 ms of numpy per step -- does not have to wait for solve k: a thread
    walks the same sequence T = 0, T + DT, ... (same floating-point accumulation as the loop) at most two steps ahead."""

    def __init__(self, prog, T, DT, Time, steps):
        import queue
        import threading
        self.q = queue.Queue(maxsize=2)
        self.error = None
        self.stop = threading.Event()           # the loop ended early (an error, `steps`): do not wait on a full queue

        def put(item):
            while not self.stop.is_set():
                try:
                    self.q.put(item, timeout=0.2)
                    return True
                except queue.Full:
                    pass
            return False

        def work():
            try:
                t, k = T, 0
                while True:
                    idx, val, moving = prog.step(t)
                    if not put((t, idx, val, moving, list(prog.groups))):
                        return
                    k += 1
                    t = t + DT
                    if not t < Time or (steps is not None and k >= steps):
                        break
            except BaseException as e:
                self.error = e
                put(None)
        self.thread = threading.Thread(target=work, name="ec3d-sources", daemon=True)
        self.thread.start()

    def close(self):
        self.stop.set()
        self.thread.join()

    def next(self, T):
        item = self.q.get()
        if item is None:
            raise self.error
        assert item[0] == T
        return item[1:]


def run(model: vxc.VxcModel, solver, steps: int | None = None, out_dir: str | None = None, on_step=None,
        on_rhs=None, on_solved=None, write_output=None, overlap_output: bool = True, on_fields=None, on_written=None):
    """The reference's run of ``model`` on ``solver`` (an EC3DSolver): assemble, then step until T >= stop (or
    ``steps`` steps).  Returns a list of per-step dicts (T, iter).  ``out_dir``: write ``field_N.vtk`` there at
    the reference's output cadence.  Hooks, all ``(k, solver, info)``: ``on_rhs`` when Jaf (B) of step k is
    built -- what the reference passes to its solver --, ``on_solved`` when Uaf (X) holds the solver's result,
    ``on_step`` after the post-update.  ``solver`` may be an EC3DMulti (N GPUs behind one handle) as well.
    ``write_output(N) -> bool`` (default: always) says whether output step N's files go to disk; the fields are
    computed on the device and brought to the host either way.

    ``overlap_output`` (default; one handle or the slabs of an EC3DMulti, whose ``fields`` are then per-slab lists,
    vtk.join_parts): the field output of step N runs beside step N+1 -- field
    kernel and device-to-host copy asynchronously, formatting-free writing on a host thread (_OutputPipeline); the
    files are the same bytes.  ``on_fields(N, fields, info)`` is then called on one of the writer threads with views of
    the pinned buffer (valid during the call; calls for different N may overlap and arrive out of order).  Without overlap the fields are fetched synchronously after the post-update,
    ``on_fields`` is called in the loop and ``info["fields"]`` holds them when they are not written.
    ``on_written(N, paths)``: after output step N's files are complete (a run of hundreds of 500 MB files may want
    to move them away)."""
    t = vxc.domain_tables(model)
    if t["dt"] is None or t["time"] is None:
        raise ValueError("the model has no 'tran stop=... step=...' line")
    solver.assemble(t["geoPHYS"], t["geoPHYS_C"], t["valPHYS"], t["BND"], t["delta"], t["dt"])
    n = solver.n
    solver.upload("X", np.zeros(n))
    solver.upload("B", np.zeros(n))
    prog = SourceProgram(model, t)
    sdz, sdy, sdx = model.vox.shape
    conducting = t["ncells0"] > 0
    DT, Time = float(t["dt"]), float(t["time"])
    DTT = float(t["jump"] or 0.0)                        # src/vxc2data.f90:191-195: unset -> 0 -> every step
    Nout = _nint(DTT / DT)
    T, Ntime, Nprint, Npoint = 0.0, 0, Nout, 0           # src/EC3D.f90:137-144
    if out_dir:
        os.makedirs(out_dir, exist_ok=True)
    pipe = None
    if out_dir and overlap_output and hasattr(solver, "vtk_fields_begin"):
        pipe = _OutputPipeline(solver, (sdx, sdy, sdz), t["delta"], out_dir, on_fields, on_written)
    log = []
    try:
        log = _time_loop(solver, t, prog, (sdx, sdy, sdz), conducting, out_dir, pipe, steps, on_step, on_rhs, on_solved,
                         write_output, on_fields, on_written, DT, Time, Nout, T, Ntime, Nprint, Npoint)
    finally:
        if pipe is not None:
            pipe.finish()
    return log


def _time_loop(solver, t, prog, dims, conducting, out_dir, pipe, steps, on_step, on_rhs, on_solved, write_output,
               on_fields, on_written, DT, Time, Nout, T, Ntime, Nprint, Npoint):
    sdx, sdy, sdz = dims
    log = []
    ahead = _SourcesAhead(prog, T, DT, Time, steps)
    try:
        while True:
            idx, val, moving, groups = ahead.next(T)
            info = dict(T=T, nsrc=len(idx))
            solver.rhs_step(idx, val, moving=moving)
            if on_rhs is not None:
                on_rhs(len(log), solver, info)
            it, _ = solver.solve_resident(t["tol"], t["itmax"])
            info["iter"] = it
            if on_solved is not None:
                on_solved(len(log), solver, info)
            solver.post_update()
            if Ntime >= Nprint and Ntime != 0:               # :437-446
                Nprint = Ntime + Nout
                Npoint += 1
                if out_dir and pipe is not None:
                    pipe.start(Npoint, groups, write_output is None or bool(write_output(Npoint)), info)
                elif out_dir:
                    f = solver.vtk_fields(t["delta"], sdx * sdy * sdz, conducting)
                    if on_fields is not None:
                        on_fields(Npoint, f, info)
                    if write_output is None or write_output(Npoint):
                        paths = [os.path.join(out_dir, f"field_{Npoint}.vtk")]
                        write_field_vtk(paths[0], sdx, sdy, sdz, t["delta"], f)
                        if groups:                           # :446  CALL writeVtk_src
                            paths.append(os.path.join(out_dir, f"src_{Npoint}.vtk"))
                            write_src_vtk(paths[1], sdx, sdy, sdz, t["delta"], groups)
                        if on_written is not None:
                            on_written(Npoint, paths)
                    else:
                        info["fields"] = f
                info["output"] = Npoint
            log.append(info)
            if on_step is not None:
                on_step(len(log) - 1, solver, info)
            Ntime += 1
            T = T + DT
            if not T < Time or (steps is not None and len(log) >= steps):
                break
    finally:
        ahead.close()
    return log


class _SlabOutputPipeline:
    """field_N.vtk of a one-process-per-GPU run without gathering anything: a z-slab's cells are consecutive bytes of
    every vector in the file (src/utilites.f90:222-289 writes cell by cell, z outermost), so each rank puts its own part
    where it belongs (``vtk.field_vtk_layout``, ``os.pwrite``) and rank 0 adds the text, the POINTS block and
    src_N.vtk.  Per rank the same pipeline as on one GPU: ``ec3d_vtk_fields_begin`` on the slab's handle behind the
    X halo exchange, three pinned buffers, a writer thread each; ``finish`` joins them -- the caller's barrier after it
    is what makes the files complete on every rank."""

    def __init__(self, s, dims, delta, conducting, out_dir, rank):
        import queue
        import threading
        from .solver import VTK_SLOTS
        from .vtk import field_vtk_layout
        self.s, self.dims, self.delta, self.out_dir, self.rank = s, dims, delta, out_dir, rank
        sdx, sdy, sdz = dims
        self.text, self.data, self.size = field_vtk_layout(sdx, sdy, sdz, delta, conducting)
        self.cell0 = s.k0 * sdx * sdy
        self.jobs, self.done, self.error = queue.Queue(), [], None
        self._Event = threading.Event
        self.slots = VTK_SLOTS
        self.threads = [threading.Thread(target=self._work, name=f"ec3d-slab-output-{i}", daemon=True)
                        for i in range(self.slots)]
        for th in self.threads:
            th.start()

    def start(self, N, groups):
        if len(self.done) >= self.slots:
            self.done[-self.slots].wait()
        self._raise()
        ops = self.s.ops
        with ops.context():
            self.s.exchange("X")                                   # the curl reads the neighbours' planes
            slot = ops.local.vtk_fields_begin(self.delta, big_endian=True)
        ev = self._Event()
        self.done.append(ev)
        self.jobs.put((slot, N, [(ax, np.array(nodes), a) for ax, nodes, a in groups] if self.rank == 0 else None, ev))

    def _work(self):
        sdx, sdy, sdz = self.dims
        while True:
            job = self.jobs.get()
            if job is None:
                return
            slot, N, groups, ev = job
            try:
                f = self.s.ops.local.vtk_fields_wait(slot, big_endian=True)
                fd = os.open(os.path.join(self.out_dir, f"field_{N}.vtk"), os.O_WRONLY | os.O_CREAT, 0o644)
                try:
                    if self.rank == 0:
                        os.ftruncate(fd, self.size)
                        for off, b in self.text:
                            _pwrite_all(fd, b, off)
                    for k, off in self.data.items():
                        part = f[k]
                        if part is None:                           # a slab without conductor: zeros in the file
                            part = np.zeros(f["A"].shape, ">f4")
                        _pwrite_all(fd, part, off + 12 * self.cell0)
                finally:
                    os.close(fd)
                if groups:                                         # rank 0: src/EC3D.f90:446  CALL writeVtk_src
                    write_src_vtk(os.path.join(self.out_dir, f"src_{N}.vtk"), sdx, sdy, sdz, self.delta, groups)
            except BaseException as e:
                self.error = e
            finally:
                ev.set()

    def _raise(self):
        if self.error is not None:
            e, self.error = self.error, None
            raise e

    def finish(self):
        for _ in self.threads:
            self.jobs.put(None)
        for th in self.threads:
            th.join()
        self._raise()


def _pwrite_all(fd, data, offset):
    mv = memoryview(np.ascontiguousarray(data) if isinstance(data, np.ndarray) else data).cast("B")
    while len(mv):
        k = os.pwrite(fd, mv, offset)
        mv, offset = mv[k:], offset + k


def run_slabs(model: vxc.VxcModel, rank: int, world: int, device: int = 0, steps: int | None = None,
              out_dir: str | None = None, on_step=None, overlap_output: bool = True):
    """The same run on ``world`` GPUs, one process per GPU (torch.distributed initialised by the caller):
    z-slabs of the A-V system (eddy_currents_3d_amd/dist.py), every rank evaluates the (tiny) source program
    itself and keeps its part of the fields resident.  Output: every rank writes its own cells into field_N.vtk
    beside the next step's solve (_SlabOutputPipeline; the files are complete when this function returns, after a
    barrier); ``overlap_output=False``: the fields gathered on rank 0, which writes them inside the loop.  Returns the
    per-step log (identical on all ranks)."""
    from .dist import HipAVSlabOps, SlabSolver, slab_bounds
    t = vxc.domain_tables(model)
    if t["dt"] is None or t["time"] is None:
        raise ValueError("the model has no 'tran stop=... step=...' line")
    sdz, sdy, sdx = model.vox.shape
    k0, k1 = slab_bounds(sdz, rank, world)
    ops = HipAVSlabOps(t["geoPHYS"], t["geoPHYS_C"], t["valPHYS"], t["BND"], t["delta"], t["dt"], k0, k1, world,
                       device=device)
    s = SlabSolver(ops, rank, world, k0, k1)
    n = 3 * model.vox.size + t["ncells0"]
    ops.set_vector_global("X", np.zeros(n))
    ops.set_vector_global("B", np.zeros(n))
    prog = SourceProgram(model, t)
    conducting = t["ncells0"] > 0
    DT, Time = float(t["dt"]), float(t["time"])
    Nout = _nint(float(t["jump"] or 0.0) / DT)
    T, Ntime, Nprint, Npoint = 0.0, 0, Nout, 0
    if out_dir and rank == 0:
        os.makedirs(out_dir, exist_ok=True)
    pipe = None
    if out_dir and overlap_output:
        if world > 1:
            s.dist.barrier()                                 # the directory exists before anyone opens a file in it
        pipe = _SlabOutputPipeline(s, (sdx, sdy, sdz), t["delta"], conducting, out_dir, rank)
    log = []
    try:
        while True:
            idx, val, moving = prog.step(T)
            s.rhs_step(idx, val, moving=moving)
            it = s.solve(t["tol"], t["itmax"])
            s.post_update()
            info = dict(T=T, iter=it)
            if Ntime >= Nprint and Ntime != 0:
                Nprint = Ntime + Nout
                Npoint += 1
                info["output"] = Npoint
                if pipe is not None:
                    pipe.start(Npoint, prog.groups)
                elif out_dir:
                    f = s.vtk_fields(t["delta"], conducting)      # gathered on rank 0
                    if rank == 0:
                        write_field_vtk(os.path.join(out_dir, f"field_{Npoint}.vtk"), sdx, sdy, sdz, t["delta"], f)
                        if prog.groups:
                            write_src_vtk(os.path.join(out_dir, f"src_{Npoint}.vtk"), sdx, sdy, sdz, t["delta"],
                                          prog.groups)
            log.append(info)
            if on_step is not None:
                on_step(len(log) - 1, s, info)
            Ntime += 1
            T = T + DT
            if not T < Time or (steps is not None and len(log) >= steps):
                break
        if pipe is not None:
            pipe.finish()
            pipe = None
            if world > 1:
                s.dist.barrier()                             # every rank's part of every file is in place
    finally:
        if pipe is not None:                                 # an error above: stop the writers, keep the error
            try:
                pipe.finish()
            except BaseException:
                pass
        ops.close()
    return log
