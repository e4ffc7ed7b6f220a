"""numpy stand-in for the per-slab device ops of eddy_currents_3d_amd/dist.py (TEST DOUBLE).

Mirrors, stage by stage, what libec3d_hip.so does for one z-slab (ec3d_dist_step), on CPU tensors,
so that SlabSolver / InProcessSlabs can be exercised over gloo without a GPU.  The local operator is
cut out of the oracle's global CSR (oracle/ec3d_oracle.c: oracle_poisson_csr)."""
import contextlib

import numpy as np
import torch

from eddy_currents_3d_amd.dist import (K1, K1_BND, K1_INT, K2, K2_BND, K2_INT, K3, K3_BND, K3_INT, K4, K5, K5_BND,
                                       K5_INT, NSLOT, RESID, SETUP)

BB, RR_INIT, D1, SS, D2, D3, RR, RR0N = range(8)
VEC = dict(X=0, B=1, R=2, R0=3, P=4, AP=5, S=6, AS=7)


class NumpySlabOps:
    def __init__(self, oracle, sdx, sdy, sdz, k0, k1, world):
        self.oracle = oracle
        self.kdz = sdx * sdy
        self.n = (k1 - k0) * self.kdz
        self.ghost = self.kdz
        self.len = self.n + 2 * self.ghost
        valA, irow, jcol = oracle.poisson_csr(sdx, sdy, sdz)
        r0, r1 = k0 * self.kdz, k1 * self.kdz
        p0, p1 = irow[r0] - 1, irow[r1] - 1
        self.valA = np.ascontiguousarray(valA[p0:p1])
        self.irow = np.ascontiguousarray(irow[r0:r1 + 1] - p0).astype(np.int32)
        self.jcol = np.ascontiguousarray(jcol[p0:p1] - r0 + self.ghost).astype(np.int32)  # ext index
        self.store = torch.zeros(8 * self.len, dtype=torch.float64)
        self.np_store = self.store.numpy()
        self.lsum = torch.zeros(NSLOT, dtype=torch.float64)
        self.gsum = torch.zeros(world * NSLOT, dtype=torch.float64)
        self.world = world
        self.overlap = True
        self.producer_side_overlap = False   # True: the A-V slabs' exchange order (ITER_PLAN_VSPLIT)
        self._ss_bnd = 0.0
        self.st = dict(stop_iter=-1, stop_kind=0, rr0=[0.0, 0.0], alpha=0.0, omega=0.0, bnorm=0.0, tol=0.0)

    def context(self):
        return contextlib.nullcontext()

    def synchronize(self):
        pass

    def _ext(self, name):
        b = VEC[name] * self.len
        return self.np_store[b:b + self.len]

    def _own(self, name):
        b = VEC[name] * self.len + self.ghost
        return self.np_store[b:b + self.n]

    def owned(self, name):
        b = VEC[name] * self.len + self.ghost
        return self.store[b:b + self.n]

    def halo_views(self, name):
        b, n, p = VEC[name] * self.len + self.ghost, self.n, self.kdz
        s = self.store
        return s[b:b + p], s[b - p:b], s[b + n - p:b + n], s[b + n:b + n + p]

    def halo_pairs(self, name):
        lo_s, lo_r, hi_s, hi_r = self.halo_views(name)
        return [(-1, lo_s, lo_r), (+1, hi_s, hi_r)]

    def set_vector(self, name, a):
        self._own(name)[:] = a

    def get_vector(self, name):
        return self._own(name).copy()

    def _g(self, slot):
        return float(sum(self.gsum[g * NSLOT + slot].item() for g in range(self.world)))

    def _spmv(self, name):
        return self.oracle.spmv_csr(self.valA, self.irow, self.jcol, self._ext(name))

    def read_state(self):
        return self.st["stop_iter"], self.st["stop_kind"], self.st["bnorm"]

    def can_overlap(self):
        return self.overlap and self.n >= 4 * self.kdz

    def enable_vsplit(self):
        """K2/K5 boundary rows (first and last plane) first; False when there is no interior row."""
        return self.n > 2 * self.kdz

    def _rows(self, boundary):
        m = np.zeros(self.n, bool)
        m[:self.kdz] = True
        m[self.n - self.kdz:] = True
        return m if boundary else ~m

    def _split(self, src, dst, interior):
        """rows of planes 1 .. np-2 (interior) or planes 0 and np-1 of dst = A * src, with the halos as
        they are at this moment"""
        y = self._spmv(src)
        p = self.kdz
        rows = slice(p, self.n - p) if interior else None
        if interior:
            self._own(dst)[rows] = y[rows]
        else:
            self._own(dst)[:p] = y[:p]
            self._own(dst)[self.n - p:] = y[self.n - p:]

    def step(self, stage, it=0, tol=0.0):
        st, L = self.st, self.lsum
        stopped_before = st["stop_iter"] >= 0 and st["stop_iter"] < it
        stopped_now = st["stop_iter"] >= 0 and st["stop_iter"] <= it
        if stage == RESID:
            r = self._own("B") - self._spmv("X")
            self._own("R")[:] = r; self._own("R0")[:] = r; self._own("P")[:] = r
            L[BB] = float(self._own("B") @ self._own("B")); L[RR_INIT] = float(r @ r)
        elif stage == SETUP:
            st.update(bnorm=np.sqrt(self._g(BB)), tol=tol, rr0=[0.0, self._g(RR_INIT)], alpha=0.0, omega=0.0,
                      stop_kind=0)
            st["stop_iter"] = 0 if st["bnorm"] == 0.0 else -1
        elif stage == K1:
            if stopped_before: return
            ap = self._spmv("P")
            self._own("AP")[:] = ap
            L[D1] = float(ap @ self._own("R0"))
        elif stage in (K1_INT, K1_BND):
            if stopped_before: return
            self._split("P", "AP", stage == K1_INT)
            if stage == K1_BND:
                L[D1] = float(self._own("AP") @ self._own("R0"))
        elif stage in (K3_INT, K3_BND):
            if stopped_before: return
            self._split("S", "AS", stage == K3_INT)
            if stage == K3_BND:
                a = self._own("AS")
                L[D2] = float(a @ self._own("S")); L[D3] = float(a @ a)
        elif stage == K2:
            if stopped_before: return
            st["alpha"] = st["rr0"][it & 1] / self._g(D1)
            s = self._own("R") - st["alpha"] * self._own("AP")
            self._own("S")[:] = s
            L[SS] = float(s @ s)
        elif stage in (K2_BND, K2_INT):
            if stopped_before: return
            st["alpha"] = st["rr0"][it & 1] / self._g(D1)
            m = self._rows(stage == K2_BND)
            s = self._own("R")[m] - st["alpha"] * self._own("AP")[m]
            self._own("S")[m] = s
            if stage == K2_BND:
                self._ss_bnd = float(s @ s)
            else:
                L[SS] = self._ss_bnd + float(s @ s)
        elif stage in (K5_BND, K5_INT):
            if stopped_now: return
            rr, rr0n = self._g(RR), self._g(RR0N)
            if np.sqrt(rr) / st["bnorm"] < st["tol"]:
                st["stop_kind"], st["stop_iter"] = 2, it
                return
            beta = (st["alpha"] / st["omega"]) * rr0n / st["rr0"][it & 1]
            restart = abs(rr0n) / st["bnorm"] < st["tol"]
            st["rr0"][(it + 1) & 1] = rr if restart else rr0n
            m = self._rows(stage == K5_BND)
            if restart:
                self._own("R0")[m] = self._own("R")[m]; self._own("P")[m] = self._own("R")[m]
            else:
                self._own("P")[m] = self._own("R")[m] + beta * (self._own("P")[m] - st["omega"] * self._own("AP")[m])
        elif stage == K3:
            if stopped_before: return
            a = self._spmv("S")
            self._own("AS")[:] = a
            L[D2] = float(a @ self._own("S")); L[D3] = float(a @ a)
        elif stage == K4:
            if stopped_before: return
            if np.sqrt(self._g(SS)) / st["bnorm"] < st["tol"]:
                self._own("X")[:] += st["alpha"] * self._own("P")
                st["stop_kind"], st["stop_iter"] = 1, it
                return
            st["omega"] = self._g(D2) / self._g(D3)
            self._own("X")[:] = (self._own("X") + st["alpha"] * self._own("P")) + st["omega"] * self._own("S")
            r = self._own("S") - st["omega"] * self._own("AS")
            self._own("R")[:] = r
            L[RR] = float(r @ r); L[RR0N] = float(r @ self._own("R0"))
        elif stage == K5:
            if stopped_now: return
            rr, rr0n = self._g(RR), self._g(RR0N)
            if np.sqrt(rr) / st["bnorm"] < st["tol"]:
                st["stop_kind"], st["stop_iter"] = 2, it
                return
            beta = (st["alpha"] / st["omega"]) * rr0n / st["rr0"][it & 1]
            restart = abs(rr0n) / st["bnorm"] < st["tol"]
            st["rr0"][(it + 1) & 1] = rr if restart else rr0n
            if restart:
                self._own("R0")[:] = self._own("R"); self._own("P")[:] = self._own("R")
            else:
                self._own("P")[:] = self._own("R") + beta * (self._own("P") - st["omega"] * self._own("AP"))
