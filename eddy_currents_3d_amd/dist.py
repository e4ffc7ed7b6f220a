"""z-slab multi-GPU BiCGSTAB-with-restart: one process per GPU, torch.distributed (nccl = RCCL over
xGMI on the GPU box, gloo in the CPU tests) between the stages of an iteration.

The reference is serial; this is new design.  Rank g of G owns the z-planes
[k0, k1) of the structured grid, i.e. one contiguous index range (k is the slowest index,
/root/reference/src/EC3D.f90:506-510).  Per iteration (src/solvers.f90:24-50):

    halo(P) -> K1 -> gather -> K2 -> halo(S) -> K3 -> gather -> K4 -> gather -> K5

* halo(v): the first/last owned plane of v goes to the z-neighbours' ghost planes (send/recv pairs,
  nearest neighbours only; the planes are contiguous, so tensors are sent in place, no packing);
* gather: all_gather of the 8 per-rank partial sums (64 B).  Every rank then adds the G values in
  the same fixed order inside the next kernel, so all ranks take bit-identical decisions
  (alpha, omega, beta, exits, restart) without any further synchronisation;
* all scalars stay on the device; the host only polls the stop flag every few iterations.

The orchestration below is backend-agnostic: `ops` supplies the per-stage local compute.  The
product ops (:class:`HipSlabOps`) drive libec3d_hip.so through the C ABI and raise when it is
missing; the CPU tests inject a numpy stand-in to exercise exactly this file over gloo.
"""
from __future__ import annotations

import numpy as np

RESID, SETUP, K1, K2, K3, K4, K5, K1_INT, K1_BND, K3_INT, K3_BND, K2_BND, K2_INT, K5_BND, K5_INT = range(15)
NSLOT = 8
# the communication/compute schedule, shared by every driver below
BEGIN_PLAN = (("halo", "X"), ("step", RESID), ("gather",), ("step", SETUP))
# three global reduction points per iteration: K3 (A*S) is launched before ||S|| is known and K4 takes
# the ||S|| exit, so S.S travels with AS.S and AS.AS (results unchanged)
ITER_PLAN = (("halo", "P"), ("step", K1), ("gather",), ("step", K2), ("halo", "S"),
             ("step", K3), ("gather",), ("step", K4), ("gather",), ("step", K5))
# the same with the halo exchange hidden behind the interior planes: K1/K3 run as an interior launch
# (planes 1 .. np-2, no halo needed) while the planes travel, then a boundary launch (planes 0, np-1).
# The sequence of collectives is identical to ITER_PLAN, so ranks may mix the two plans.
ITER_PLAN_OVERLAP = (("halo_start", "P"), ("step", K1_INT), ("halo_wait", "P"), ("step", K1_BND), ("gather",),
                     ("step", K2), ("halo_start", "S"), ("step", K3_INT), ("halo_wait", "S"), ("step", K3_BND),
                     ("gather",), ("step", K4), ("gather",), ("step", K5))


# the same idea from the producers' side, for any slab and storage format (A-V slabs): K2 (makes S) and K5
# (makes P) run their boundary tiles first, the exchange starts, the interior tiles run while the planes
# travel; the consumer (K3 / next K1) waits for the exchange.  P is exchanged once before the first iteration.
BEGIN_PLAN_VSPLIT = BEGIN_PLAN + (("halo", "P"),)
ITER_PLAN_VSPLIT = (("halo_wait", "P"), ("step", K1), ("gather",), ("step", K2_BND), ("halo_start", "S"),
                    ("step", K2_INT), ("halo_wait", "S"), ("step", K3), ("gather",), ("step", K4), ("gather",),
                    ("step", K5_BND), ("halo_start", "P"), ("step", K5_INT))


def unsplit_stage(stage):
    """A rank whose slab is all boundary runs the whole kernel where the others run *_BND and nothing where
    they run *_INT: same order of exchanges and collectives on every rank."""
    return {K2_BND: K2, K5_BND: K5, K2_INT: None, K5_INT: None}.get(stage, stage)


def slab_bounds(sdz: int, rank: int, world: int):
    """Planes [k0, k1) of rank `rank`: as even as possible, lower ranks take the remainder."""
    base, rem = divmod(sdz, world)
    k0 = rank * base + min(rank, rem)
    return k0, k0 + base + (1 if rank < rem else 0)


def share_unique_ids(ids, rank: int, world: int, device: int):
    """Rank 0's two RCCL unique ids (128 bytes each) handed to every rank of the torch.distributed job that is already
    running -- as ONE uint8 tensor on the backend's own kind of memory (this rank's GPU under nccl, the host under gloo),
    so no object is pickled through a side channel.  Returns [id_halo, id_sum] on every rank."""
    import torch
    import torch.distributed as dist
    dev = torch.device("cuda", device) if dist.get_backend() == "nccl" else torch.device("cpu")
    if rank == 0:
        if len(ids[0]) != 128 or len(ids[1]) != 128:
            raise ValueError("an RCCL unique id is 128 bytes")
        t = torch.frombuffer(bytearray(ids[0] + ids[1]), dtype=torch.uint8).to(dev)
    else:
        t = torch.zeros(256, dtype=torch.uint8, device=dev)
    dist.broadcast(t, src=0)
    raw = bytes(t.cpu().numpy().tobytes())
    return [raw[:128], raw[128:]]


def rccl_rank(rank: int, world: int, device: int, dictionary=None, structured=None, rehearse=None):
    """This process's slab of a `world`-rank job as an EC3DMulti driven from C++ over RCCL (include/ec3d_hip.h section 2c,
    ec3d_multi_create_rank): rank 0 makes the two RCCL unique ids, torch.distributed -- already initialised by the
    launcher's form of this job -- hands them to every rank (share_unique_ids).  Without a process group (a one-rank job
    or a rehearsal started by hand) the ids stay local; a job of several ranks without one is refused: its ranks could
    not meet."""
    from .solver import EC3DMulti
    have_group = False
    try:
        import torch.distributed as dist
        have_group = dist.is_available() and dist.is_initialized() and dist.get_world_size() == world
    except ImportError:
        pass
    if world > 1 and not have_group:
        raise RuntimeError(f"rccl_rank: a job of {world} ranks needs torch.distributed initialised with that world size "
                           f"(it carries rank 0's RCCL ids to the others)")
    ids = [None, None]
    if rank == 0:
        ids = [EC3DMulti.rccl_unique_id(), EC3DMulti.rccl_unique_id()]
    if world > 1:
        ids = share_unique_ids(ids, rank, world, device)
    return EC3DMulti.for_rank(rank, world, device, ids[0], ids[1], dictionary=dictionary, structured=structured,
                              rehearse=rehearse)


class HipSlabOps:
    """One z-slab on one MI355X through the C ABI (include/ec3d_hip.h §2b)."""

    VEC = dict(X=0, B=1, R=2, R0=3, P=4, AP=5, S=6, AS=7)

    def __init__(self, sdx, sdy, sdz, k0, k1, world, device=0, delta=(0.00333,) * 3, bnd=-0.95,
                 dictionary=None):
        import torch
        from .solver import EC3DSolver
        self.torch = torch
        self.device = torch.device("cuda", device)
        self.stream = torch.cuda.Stream(device=self.device)
        self.local = EC3DSolver(device=device, dictionary=dictionary)
        self.local.set_stream(self.stream.cuda_stream)
        self.local.assemble_poisson(sdx, sdy, sdz, delta, bnd, slab=(k0, k1))
        self.k0, self.k1 = k0, k1
        lay = self.local.vector_layout()
        self.n, self.ghost, self.kdz = lay["n"], lay["ghost"], sdx * sdy
        self.len = 2 * lay["ghost"] + lay["n_pad"]
        with torch.cuda.stream(self.stream):
            self.store = torch.zeros(8 * self.len, dtype=torch.float64, device=self.device)
            self.lsum = torch.zeros(NSLOT, dtype=torch.float64, device=self.device)
            self.gsum = torch.zeros(world * NSLOT, dtype=torch.float64, device=self.device)
        self.stream.synchronize()
        self.local.adopt_vectors(self.store.data_ptr())
        self.local.dist_configure(world, self.lsum.data_ptr(), self.gsum.data_ptr())

    def context(self):
        return self.torch.cuda.stream(self.stream)

    def close(self):
        """Detach the library from the torch-owned stream and vectors BEFORE torch frees them."""
        local = getattr(self, "local", None)
        if local is not None and getattr(local, "h", None) and local.h.value:
            try:
                self.stream.synchronize()
                local.set_stream(None)
            finally:
                local.close()
        self.local = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _base(self, name):
        return self.VEC[name] * self.len + self.ghost

    def owned(self, name):
        b = self._base(name)
        return self.store[b:b + self.n]

    def halo_views(self, name):
        b, n, p = self._base(name), self.n, self.kdz
        return (self.store[b:b + p], self.store[b - p:b],          # lo: send first plane, recv ghost
                self.store[b + n - p:b + n], self.store[b + n:b + n + p])  # hi

    def halo_pairs(self, name):
        """[(direction, send view, recv view)]: direction -1 = lower z-neighbour, +1 = upper."""
        lo_s, lo_r, hi_s, hi_r = self.halo_views(name)
        return [(-1, lo_s, lo_r), (+1, hi_s, hi_r)]

    def export_owned(self, name, out):
        """Write this slab's part of vector `name` into the global array `out` (reference numbering)."""
        out[self.k0 * self.kdz:self.k1 * self.kdz] = self.get_vector(name)

    def set_vector(self, name, host_array):
        with self.context():
            self.owned(name).copy_(self.torch.from_numpy(np.ascontiguousarray(host_array, np.float64)))

    def get_vector(self, name):
        with self.context():
            return self.owned(name).cpu().numpy()

    def step(self, stage, it=0, tol=0.0):
        self.local.dist_step(stage, it, tol)

    def can_overlap(self):
        return self.local.can_overlap()

    def enable_vsplit(self):
        """Tell the library which rows are sent in a halo exchange (the send views of halo_pairs), so K2/K5
        can run boundary-first.  Returns False when there is nothing to split."""
        b0 = self._base("P")
        ranges = []
        for _, send, recv in self.halo_pairs("P"):
            # the rows sent must be final before the exchange starts; the rows RECEIVED must not be written
            # after it (halo rows of an extended slab are swept like any other row): both go first
            for view in (send, recv):
                if view.numel():
                    lo = view.storage_offset() - b0
                    ranges.append((lo, lo + view.numel()))
        if not ranges:
            return False
        return self.local.dist_set_boundary_rows(ranges)

    def read_state(self):
        return self.local.read_state()

    def stop_flag_probe(self):
        """Non-blocking look at the stop flag: enqueue a copy into pinned memory behind the work issued so
        far; returns a function that waits for that copy only and gives the flag (-1 = still running)."""
        torch = self.torch
        if not hasattr(self, "_pin"):
            self._pin = torch.zeros(2, dtype=torch.int32).pin_memory()
            self._pin_ev = [torch.cuda.Event(), torch.cuda.Event()]
            self._pin_i = 0
        i = self._pin_i & 1
        self._pin_i += 1
        self.local.read_state_async(self._pin.data_ptr() + 4 * i)
        self._pin_ev[i].record(self.stream)

        def result():
            self._pin_ev[i].synchronize()
            v = int(self._pin[i])
            return -1 if v == 2147483647 else v
        return result

    def timed(self, fn):
        """Run fn() on the stream between two events; returns a closure giving ms after a sync."""
        e0, e1 = self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)
        e0.record(self.stream)
        fn()
        e1.record(self.stream)
        return lambda: e0.elapsed_time(e1)

    def synchronize(self):
        self.stream.synchronize()

    def device_synchronize(self):
        """Everything enqueued on the device, whatever the stream (InProcessSlabs copies between slabs on
        the caller's current stream)."""
        self.torch.cuda.synchronize(self.device)


class HipAVSlabOps(HipSlabOps):
    """One z-slab of the full A-V system [Ax | Ay | Az | U] on one MI355X (ec3d_assemble_slab).

    The handle holds the owned planes [k0, k1) plus H = 2 halo planes on every interior side (the
    one-sided A-U stencils of /root/reference/src/EC3D.f90:697-706 reach two cells) for each of the
    three A components, and the U unknowns of all those planes.  Rows of halo planes are inert and
    masked out of the dot products; their vector entries are overwritten by the halo exchange, which
    moves 4 contiguous ranges per neighbour (the nearest plane of Ax, Ay, Az and two planes of U)."""

    H = 2
    # every rank of an A-V job uses the exchange order of ITER_PLAN_VSPLIT (a property of the job, not of
    # the slab: ranks must agree on it)
    producer_side_overlap = True

    def __init__(self, geoPHYS, geoPHYS_C, valPHYS, BND, delta, dt, k0, k1, world, device=0, dictionary=None,
                 structured=None):
        import torch
        from .solver import EC3DSolver
        self.torch = torch
        sdz, sdy, sdx = geoPHYS.shape
        H = self.H
        if k1 - k0 < H:
            raise ValueError("a slab needs at least two planes")
        e0, e1 = max(0, k0 - H), min(sdz, k1 + H)
        self.k0, self.k1, self.e0, self.e1 = k0, k1, e0, e1
        self.kdz = sdx * sdy
        kdz = self.kdz
        cond = np.asarray(geoPHYS_C) != 0                        # [k, j, i]
        per_plane = cond.reshape(sdz, -1).sum(axis=1)
        self.nC_global = sdz * kdz
        self.nU_global = int(per_plane.sum())
        self.u_global0 = int(per_plane[:e0].sum())               # global U index of my first (extended) U
        nC = (e1 - e0) * kdz
        self.nC = nC
        cond_ext = cond[e0:e1]
        geoC_ext = np.zeros(cond_ext.shape, np.int32)
        self.nU = int(cond_ext.sum())
        geoC_ext[cond_ext] = 3 * nC + 1 + np.arange(self.nU, dtype=np.int32)
        u_in = lambda a, b: int(per_plane[a:b].sum())            # conducting cells in global planes [a, b)
        self.device = torch.device("cuda", device)
        self.stream = torch.cuda.Stream(device=self.device)
        self.local = EC3DSolver(device=device, dictionary=dictionary, structured=structured)
        self.local.set_stream(self.stream.cuda_stream)
        self.local.assemble_slab(sdz, e0, e1, k0, k1, np.asarray(geoPHYS)[e0:e1], geoC_ext, valPHYS, BND, delta, dt)
        lay = self.local.vector_layout()
        self.n, self.ghost = lay["n"], lay["ghost"]          # device rows
        self.n_ref = 3 * nC + self.nU                         # local unknowns, reference order [Ax|Ay|Az|U]
        # device row of every local unknown: identity for bands + tail; the structured form embeds U in the
        # grid and may pad the planes
        rm = self.local.row_map().astype(np.int64)
        assert len(rm) == self.n_ref
        self.structured = self.n != self.n_ref
        self.len = 2 * lay["ghost"] + lay["n_pad"]
        with torch.cuda.stream(self.stream):
            self.store = torch.zeros(8 * self.len, dtype=torch.float64, device=self.device)
            self.lsum = torch.zeros(NSLOT, dtype=torch.float64, device=self.device)
            self.gsum = torch.zeros(world * NSLOT, dtype=torch.float64, device=self.device)
            self._rm = torch.from_numpy(rm).to(self.device)
        self.stream.synchronize()
        self.local.adopt_vectors(self.store.data_ptr())
        self.local.dist_configure(world, self.lsum.data_ptr(), self.gsum.data_ptr())
        # contiguous (send, recv) DEVICE row ranges per neighbour
        p0, p1 = k0 - e0, k1 - e0
        self._ranges = []
        # planes exchanged per block: the 7-point stencil and the U rows read A one plane away; only the
        # one-sided A-U stencils reach two planes, and they read U
        if self.structured:   # four grid-shaped blocks of nCd rows, `pitch` rows per plane
            nCd = self.n // 4
            pitch = nCd // (e1 - e0)
            blocks = [(d * nCd, pitch, 1 if d < 3 else H) for d in range(4)]
        else:                 # three grid-shaped blocks; the U block is compact (handled below)
            blocks = [(d * nC, kdz, 1) for d in range(3)]
        if e0 < k0:   # lower neighbour exists: send my first owned planes, receive my lower halo planes
            for base, pl, h in blocks:
                self._ranges.append((-1, (base + p0 * pl, base + (p0 + h) * pl), (base + (p0 - h) * pl, base + p0 * pl)))
            if not self.structured:
                ulo = u_in(e0, k0)
                self._ranges.append((-1, (3 * nC + ulo, 3 * nC + ulo + u_in(k0, k0 + H)), (3 * nC, 3 * nC + ulo)))
        if k1 < e1:   # upper neighbour
            for base, pl, h in blocks:
                self._ranges.append((+1, (base + (p1 - h) * pl, base + p1 * pl), (base + p1 * pl, base + (p1 + h) * pl)))
            if not self.structured:
                uown_end = u_in(e0, k1)
                self._ranges.append((+1, (3 * nC + uown_end - u_in(k1 - H, k1), 3 * nC + uown_end),
                                     (3 * nC + uown_end, 3 * nC + self.nU)))
        # owned rows in LOCAL reference numbering (what get_vector returns)
        self._own = [(d * nC + p0 * kdz, d * nC + p1 * kdz) for d in range(3)] + \
                    [(3 * nC + u_in(e0, k0), 3 * nC + u_in(e0, k1))]

    def set_vector(self, name, host_array):
        """host_array: the slab's unknowns in local reference order [Ax | Ay | Az | U] (held planes)."""
        with self.context():
            src = self.torch.from_numpy(np.ascontiguousarray(host_array, np.float64)).to(self.device)
            b = self._base(name)
            self.store[b:b + self.n].index_copy_(0, self._rm, src)

    def get_vector(self, name):
        with self.context():
            b = self._base(name)
            return self.store[b:b + self.n].index_select(0, self._rm).cpu().numpy()

    def halo_pairs(self, name):
        b = self._base(name)
        return [(d, self.store[b + s0:b + s1], self.store[b + r0:b + r1]) for d, (s0, s1), (r0, r1) in self._ranges]

    def _global_index(self):
        """global (reference) index of every local row of the extended slab"""
        nC, kdz = self.nC, self.kdz
        cell = self.e0 * kdz + np.arange(nC)
        return np.concatenate([d * self.nC_global + cell for d in range(3)] +
                              [3 * self.nC_global + self.u_global0 + np.arange(self.nU)])

    # ---- the time loop around the solve (src/EC3D.f90:275-404, :412-433) on this slab -------------
    def rhs_step(self, src_index, src_value, moving=False):
        """Jaf of this step on the held planes.  src_index: 1-based GLOBAL ids of the A unknowns the host's
        source functions act on (what a single-device ec3d_rhs_step takes); sources outside the extended
        slab are dropped, the rest renumbered locally.  Needs current X halo planes (the U-row right-hand
        sides read A one plane away): the drivers exchange X first."""
        g0 = np.asarray(src_index, np.int64) - 1
        val = np.asarray(src_value, np.float64)
        d, cell = np.divmod(g0, self.nC_global)
        plane = cell // self.kdz
        keep = (plane >= self.e0) & (plane < self.e1)
        local = d[keep] * self.nC + (cell[keep] - self.e0 * self.kdz) + 1
        with self.context():
            self.local.rhs_step(local.astype(np.int32), val[keep], moving=moving)

    def post_update(self):
        with self.context():
            self.local.post_update()

    def vtk_fields(self, delta, conducting=True):
        """The four float32 point vectors of field_N.vtk for the OWNED planes (ec3d_vtk_fields); needs
        current X halo planes for the curl.  conducting: the problem has conductors somewhere (the eddy
        field exists in the file even where this slab holds none)."""
        ncell = (self.k1 - self.k0) * self.kdz
        with self.context():
            f = self.local.vtk_fields(delta, ncell, conducting, zero_eddy=True)
        return f

    def set_vector_global(self, name, global_vec):
        """Fill owned AND halo entries from a global vector in the reference's numbering."""
        self.set_vector(name, np.asarray(global_vec, np.float64)[self._global_index()])

    def export_owned(self, name, out):
        v = self.get_vector(name)
        gi = self._global_index()
        for lo, hi in self._own:
            out[gi[lo:hi]] = v[lo:hi]


class SlabSolver:
    """Backend-agnostic distributed loop (see module docstring)."""

    def __init__(self, ops, rank: int, world: int, k0: int, k1: int):
        self.ops, self.rank, self.world, self.k0, self.k1 = ops, rank, world, k0, k1
        self.n_local = ops.n
        self._p2p_cache = {}
        self._pending = {}
        self.dist = None
        self.host_staged = False
        self.begin_plan = BEGIN_PLAN
        self.split_ok = True
        if getattr(ops, "can_overlap", lambda: False)():
            self.iter_plan = ITER_PLAN_OVERLAP           # single-component slab: K1/K3 interior + boundary
        elif world > 1 and getattr(ops, "producer_side_overlap", False):
            # any other slab (A-V): K2/K5 boundary first.  The ORDER of exchanges is what all ranks share; a
            # rank with nothing to split runs the whole kernels in that order (unsplit_stage)
            self.iter_plan, self.begin_plan = ITER_PLAN_VSPLIT, BEGIN_PLAN_VSPLIT
            self.split_ok = bool(ops.enable_vsplit())
        else:
            self.iter_plan = ITER_PLAN
        if world > 1:
            import torch
            import torch.distributed as dist
            self.dist = dist
            self.torch = torch
            # RCCL moves device memory directly; gloo (CPU tests, single-GPU rehearsals) cannot send
            # or gather GPU tensors, so device data is staged through the host for it
            self.host_staged = dist.get_backend() == "gloo" and getattr(ops.lsum, "is_cuda", False)
        else:
            try:  # a 1-rank process group (rehearsal): still go through the collective
                import torch.distributed as dist
                if dist.is_available() and dist.is_initialized():
                    self.dist = dist
            except ImportError:
                pass

    @classmethod
    def poisson_cube(cls, N, rank, world, device=0, dictionary=None):
        k0, k1 = slab_bounds(N, rank, world)
        return cls(HipSlabOps(N, N, N, k0, k1, world, device=device, dictionary=dictionary), rank, world, k0, k1)

    @property
    def local(self):
        return self.ops.local

    # ---- communication -----------------------------------------------------------------------
    def exchange(self, name, start_only=False):
        """Nearest-neighbour halo planes of vector `name` (no-op for a single rank).  start_only: enqueue
        the transfers and return; exchange_wait(name) makes the compute stream wait for them (RCCL runs
        them on its own stream, so kernels launched in between overlap with the transfer)."""
        if self.world == 1:
            return
        d = self.dist
        p2p = self._p2p_cache.get(name)
        if p2p is None:  # the views stay valid for the life of the vectors: build the op list once
            p2p = []
            for direction, send, recv in self.ops.halo_pairs(name):
                peer = self.rank + direction
                if 0 <= peer < self.world and send.numel() > 0:
                    p2p += [d.P2POp(d.isend, send, peer), d.P2POp(d.irecv, recv, peer)]
            self._p2p_cache[name] = p2p
        if not p2p:
            return
        if self.host_staged:  # gloo has no GPU send/recv: stage the planes through pinned-free host copies
            pairs = [(p2p[i].tensor, p2p[i + 1].tensor, p2p[i].peer) for i in range(0, len(p2p), 2)]
            sends = [t.cpu() for t, _, _ in pairs]
            recvs = [self.torch.empty(r.shape, dtype=r.dtype) for _, r, _ in pairs]
            ops = []
            for (_, _, peer), sc, rc in zip(pairs, sends, recvs):
                ops += [d.P2POp(d.isend, sc, peer), d.P2POp(d.irecv, rc, peer)]
            for req in d.batch_isend_irecv(ops):
                req.wait()
            for (_, r, _), rc in zip(pairs, recvs):
                r.copy_(rc)
            return  # complete already; exchange_wait() finds nothing pending
        reqs = d.batch_isend_irecv(p2p)
        if start_only:
            self._pending[name] = reqs
            return
        for req in reqs:
            req.wait()

    def exchange_wait(self, name):
        for req in self._pending.pop(name, ()):
            req.wait()

    def _drain_exchanges(self):
        """An exit leaves the exchange started by the last producer stage un-waited: join it before the
        vectors are touched again."""
        for name in list(self._pending):
            self.exchange_wait(name)

    def gather(self):
        """gsum[g*8 + slot] <- rank g's lsum[slot]; summed in rank order inside the next kernel."""
        if self.dist is None:
            self.ops.gsum.copy_(self.ops.lsum)
        elif self.host_staged:
            l_cpu = self.ops.lsum.cpu()
            g_cpu = self.torch.empty(self.world * NSLOT, dtype=l_cpu.dtype)
            self.dist.all_gather_into_tensor(g_cpu, l_cpu)
            self.ops.gsum.copy_(g_cpu)
        else:
            self.dist.all_gather_into_tensor(self.ops.gsum, self.ops.lsum)

    # ---- algorithm ---------------------------------------------------------------------------
    def set_rhs(self, b_local, x_local):
        self.ops.set_vector("B", b_local)
        self.ops.set_vector("X", x_local)

    def begin(self, tol):
        """R = B - A X, R0 = P = R, Bnorm, rr0 (src/solvers.f90:14-23)."""
        with self.ops.context():
            self._run(self.begin_plan, 0, tol, None)

    def _run(self, plan, it, tol, timers):
        ops = self.ops
        for op in plan:
            if op[0] == "halo":
                self.exchange(op[1])
            elif op[0] == "halo_start":
                self.exchange(op[1], start_only=True)
            elif op[0] == "halo_wait":
                self.exchange_wait(op[1])
            elif op[0] == "gather":
                self.gather()
            else:
                stage = op[1] if self.split_ok else unsplit_stage(op[1])
                if stage is None:
                    continue
                if timers is None:
                    ops.step(stage, it, tol)
                else:
                    timers.setdefault(stage, []).append(ops.timed(lambda st=stage: ops.step(st, it, tol)))

    def iteration(self, it, timers=None):
        self._run(self.iter_plan, it, 0.0, timers)

    def solve(self, tol, itmax, poll=8):
        """One reference solve on the resident slab of b/x.  Returns iter (identical on all ranks)."""
        total = max(0, itmax + 1)                      # src/solvers.f90:25-29
        self.begin(tol)
        it = 0
        probe = getattr(self.ops, "stop_flag_probe", None)
        with self.ops.context():
            pending = None   # look at chunk c-1 only after chunk c is enqueued: the GPU never waits for the host
            while it < total:
                for _ in range(min(poll, total - it)):
                    it += 1
                    self.iteration(it)
                if probe is None:
                    stop_iter = self.ops.read_state()[0]
                else:
                    pending, prev = probe(), pending
                    stop_iter = prev() if prev is not None else -1
                # every rank sees the same flag at the same chunk (the exit decision is computed from the
                # same gathered sums everywhere), so all ranks leave the loop together; iterations enqueued
                # past the exit return at once and touch nothing
                if stop_iter >= 0:
                    break
            self._drain_exchanges()
        stop_iter, _, _ = self.ops.read_state()
        return stop_iter if stop_iter >= 0 else total

    # ---- the time loop around the solve (A-V slabs) ---------------------------------------------
    def rhs_step(self, src_index, src_value, moving=False):
        """Build this step's right-hand side on every rank (src/EC3D.f90:275-404): same arguments on all
        ranks (global source ids and values); b and x stay resident."""
        with self.ops.context():
            self.exchange("X")
        self.ops.rhs_step(src_index, src_value, moving)

    def post_update(self):
        """src/EC3D.f90:412-433 on the owned planes (and their halo copies)."""
        self.ops.post_update()

    def vtk_fields(self, delta, conducting=True):
        """Field vectors of the whole grid on rank 0 (None elsewhere): every rank computes its owned planes
        on its GPU (X halo refreshed first), the float32 slabs are gathered in rank order."""
        with self.ops.context():
            self.exchange("X")
        mine = self.ops.vtk_fields(delta, conducting)
        if self.world == 1:
            return mine
        parts = [None] * self.world if self.rank == 0 else None
        self.dist.gather_object(mine, parts, dst=0)
        if self.rank != 0:
            return None
        return {k: (None if parts[0][k] is None else np.concatenate([p[k] for p in parts])) for k in parts[0]}

    # ---- bench "steps": exits disabled ---------------------------------------------------------
    def iterate_begin(self):
        self.begin(-1.0)
        self.ops.synchronize()

    def iterate(self, first_iter, count, per_kernel=False):
        timers = {} if per_kernel else None
        with self.ops.context():
            for it in range(first_iter, first_iter + count):
                self.iteration(it, timers)
        if not per_kernel:
            return None
        with self.ops.context():
            self._drain_exchanges()
        self.ops.synchronize()
        names = {K1: "k1", K2: "k2", K3: "k3", K4: "k4", K5: "k5", K1_INT: "k1", K1_BND: "k1", K3_INT: "k3",
                 K3_BND: "k3", K2_BND: "k2", K2_INT: "k2", K5_BND: "k5", K5_INT: "k5"}
        out = {}
        for st, ts in timers.items():  # split launches add up to the kernel they stand for
            out[names[st]] = out.get(names[st], 0.0) + float(np.mean([t() for t in ts]))
        return out

    def gather_x(self):
        """Global solution on every rank (testing / small problems)."""
        x = self.ops.get_vector("X")
        if self.world == 1:
            return x
        import torch
        parts = [None] * self.world
        self.dist.all_gather_object(parts, x)
        return np.concatenate(parts)


class InProcessSlabs:
    """The same schedule over several slabs held by ONE process (all on one GPU, or numpy stand-ins):
    halo planes are copied tensor to tensor and the per-slab sums concatenated, so the slab kernels,
    the ghost-plane layout and the rank-ordered reduction can be validated without a second GPU."""

    def __init__(self, ops_list, overlap=None, vsplit=False):
        self.ops_list = ops_list
        self.world = len(ops_list)
        can = all(getattr(o, "can_overlap", lambda: False)() for o in ops_list)
        self.begin_plan = BEGIN_PLAN
        self.iter_plan = ITER_PLAN_OVERLAP if (can if overlap is None else overlap) else ITER_PLAN
        self.split_ok = [True] * self.world
        if vsplit:  # K2/K5 boundary-first (what SlabSolver picks for A-V slabs)
            self.split_ok = [bool(o.enable_vsplit()) for o in ops_list]
            self.iter_plan, self.begin_plan = ITER_PLAN_VSPLIT, BEGIN_PLAN_VSPLIT

    def _halo(self, name):
        pairs = [o.halo_pairs(name) for o in self.ops_list]
        for g in range(self.world - 1):
            up = [(s_, r_) for d_, s_, r_ in pairs[g] if d_ == +1]        # slab g talking to g+1
            down = [(s_, r_) for d_, s_, r_ in pairs[g + 1] if d_ == -1]  # slab g+1 talking to g
            assert len(up) == len(down)
            for (s_up, r_up), (s_dn, r_dn) in zip(up, down):
                r_up.copy_(s_dn)   # my upper ghost <- next slab's first planes
                r_dn.copy_(s_up)   # next slab's lower ghost <- my last planes

    def _gather(self):
        for o in self.ops_list:
            for g, src in enumerate(self.ops_list):
                o.gsum[g * NSLOT:(g + 1) * NSLOT].copy_(src.lsum)

    def _run(self, plan, it, tol):
        for op in plan:
            if op[0] in ("halo", "halo_start"):
                self._sync()
                self._halo(op[1])
                self._sync()
            elif op[0] == "halo_wait":
                pass
            elif op[0] == "gather":
                self._sync()
                self._gather()
                self._sync()
            else:
                for o, ok in zip(self.ops_list, self.split_ok):
                    stage = op[1] if ok else unsplit_stage(op[1])
                    if stage is None:
                        continue
                    with o.context():
                        o.step(stage, it, tol)

    def _sync(self):
        for o in self.ops_list:
            o.synchronize()
        # the slab-to-slab copies of _halo/_gather run on the caller's current stream, which the slabs' own
        # (non-blocking) streams do not wait for
        ds = getattr(self.ops_list[0], "device_synchronize", None)
        if ds is not None:
            ds()

    def solve(self, tol, itmax):
        total = max(0, itmax + 1)
        self._run(self.begin_plan, 0, tol)
        for it in range(1, total + 1):
            self._run(self.iter_plan, it, 0.0)
            states = [o.read_state()[0] for o in self.ops_list]
            assert len(set(states)) == 1, f"slabs disagree on the stop flag: {states}"
            if states[0] >= 0:
                return states[0]
        return total

    def rhs_step(self, src_index, src_value, moving=False):
        self._sync()
        self._halo("X")
        self._sync()
        for o in self.ops_list:
            o.rhs_step(src_index, src_value, moving)

    def post_update(self):
        for o in self.ops_list:
            o.post_update()

    def vtk_fields(self, delta, conducting=True):
        self._sync()
        self._halo("X")
        self._sync()
        parts = [o.vtk_fields(delta, conducting) for o in self.ops_list]
        return {k: (None if parts[0][k] is None else np.concatenate([p[k] for p in parts])) for k in parts[0]}

    def vector(self, name, n_global):
        """Global vector `name` in the reference's numbering, owned parts of every slab."""
        out = np.zeros(n_global)
        self._sync()
        for o in self.ops_list:
            o.export_owned(name, out)
        return out

    def x(self, n_global=None):
        """Global solution in the reference's numbering."""
        if n_global is None:
            return np.concatenate([o.get_vector("X") for o in self.ops_list])
        out = np.zeros(n_global)
        for o in self.ops_list:
            o.export_owned("X", out)
        return out
