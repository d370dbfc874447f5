"""CPU stand-ins for the GPU engines, built on the oracle - TEST DOUBLES ONLY.

They let the multi-rank driver logic (sweep sharding; domain decomposition with halo exchange and the
all-gathered step control) run under ``gloo`` with world_size 2 in a container without a GPU.  The product
never imports this module."""
import math

import numpy as np
import torch

from oracle import oracle as orc

A = [[], [1 / 5], [3 / 40, 9 / 40], [44 / 45, -56 / 15, 32 / 9], [19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729],
     [9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656]]
B = [35 / 384, 0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84]
E = [-71 / 57600, 0, 71 / 16695, -71 / 1920, 17253 / 339200, -22 / 525, 1 / 40]


class OracleSweepEngine:
    def __init__(self, base_parms, instances):
        self.N = int(base_parms["N"])
        self.P = [orc.params_from_dict(base_parms | inst) for inst in instances]

    def integrate_rk45(self, y0, t_span, first_step, rtol, atol, max_attempts):
        from marlpde_amd.LHeureux_model import RK45Result
        ys, res = [], []
        for P, y in zip(self.P, y0):
            yf, st, *_ = orc.rk45(P, self.N, y, t_span[0], t_span[1], first_step, rtol, atol, max_attempts=max_attempts)
            ys.append(yf)
            res.append(RK45Result(st))
        return np.array(ys).reshape(len(self.P), 5 * self.N), res

    def integrate_radau(self, y0, t_span, first_step, rtol, atol, max_attempts):
        from marlpde_amd.LHeureux_model import RK45Result
        ys, res = [], []
        for P, y in zip(self.P, y0):
            yf, st, *_ = orc.radau(P, self.N, y, t_span[0], t_span[1], first_step, rtol, atol, max_attempts=max_attempts)
            ys.append(yf)
            res.append(RK45Result(st))
        return np.array(ys).reshape(len(self.P), 5 * self.N), res

    def close(self):
        pass


class OracleSlabEngine:
    """Same interface as marlpde_amd.domain.HipSlabEngine, arithmetic by the oracle's full-grid RHS."""

    def __init__(self, pde_parms, N_global, begin, end, halo):
        self.P = orc.params_from_dict(pde_parms)
        self.N, self.halo = N_global, halo
        self.hl = halo if begin > 0 else 0
        self.hr = halo if end < N_global else 0
        self.goff, self.n_own = begin - self.hl, end - begin
        self.n_buf = self.hl + self.n_own + self.hr
        self.Y = [np.zeros((5, self.n_buf)), np.zeros((5, self.n_buf))]
        self.F = [np.zeros((5, self.n_buf)), np.zeros((5, self.n_buf))]
        self.filler = np.array([pde_parms[k] for k in ("CA0", "CC0", "cCa0", "cCO30", "Phi0")])[:, None]
        self.c = None

    def new_tensor(self, n):
        return torch.zeros(n, dtype=torch.float64)

    def own(self):
        return slice(self.hl, self.hl + self.n_own)

    def _which(self, w):
        return w if w >= 0 else (self.c["cur"] ^ 1 if w == -1 else self.c["cur"])

    def _rhs_buf(self, ybuf):
        """RHS on every buffer cell via the oracle's full-grid routine (cells next to a missing neighbour are
        garbage - exactly the cells the fused kernel invalidates, one layer per evaluation)."""
        g = np.tile(self.filler, (1, self.N))
        g[:, self.goff:self.goff + self.n_buf] = ybuf
        with np.errstate(all="ignore"):
            r = orc.rhs(self.P, self.N, g.ravel()).reshape(5, self.N)
        return r[:, self.goff:self.goff + self.n_buf]

    def _record(self, ybuf_owned, sumsq):
        g = np.tile(ybuf_owned[:, :1], (1, self.N))
        g[:, :self.n_own] = ybuf_owned
        ev = orc.events(self.P, self.N, g.ravel())
        return np.array([sumsq, ev[0], ev[1], ev[2], ev[5], ev[3] + 1, ev[4] + 1, ev[6]])

    def load(self, y_owned):
        self.Y[0][:, self.own()] = y_owned.numpy().reshape(5, self.n_own)

    def store(self, y_owned):
        y_owned.copy_(torch.from_numpy(self.Y[self.c["cur"]][:, self.own()].copy().ravel()))

    def pack(self, which, send_lo, send_hi):
        w, h, o = self._which(which), self.halo, self.own()
        lo = np.stack([self.Y[w][:, o][:, :h], self.F[w][:, o][:, :h]])
        hi = np.stack([self.Y[w][:, o][:, -h:], self.F[w][:, o][:, -h:]])
        send_lo.copy_(torch.from_numpy(lo.ravel().copy()))
        send_hi.copy_(torch.from_numpy(hi.ravel().copy()))

    def unpack(self, which, recv_lo, recv_hi):
        w, h = self._which(which), self.halo
        if self.hl:
            s = recv_lo.numpy().reshape(2, 5, h)
            self.Y[w][:, :h], self.F[w][:, :h] = s[0], s[1]
        if self.hr:
            s = recv_hi.numpy().reshape(2, 5, h)
            self.Y[w][:, -h:], self.F[w][:, -h:] = s[0], s[1]

    def rhs0(self):
        self.F[0][:, self.own()] = self._rhs_buf(self.Y[0])[:, self.own()]

    def monitors(self, rec):
        rec.copy_(torch.from_numpy(self._record(self.Y[0][:, self.own()], 0.0)))

    # -- controller: scipy/integrate/_ivp/rk.py:111-176 restated once more, for the test double only
    def _prepare(self):
        c = self.c
        min_step = 10 * abs(np.nextafter(c["t"], np.inf) - c["t"])
        if not c["rejected"]:
            c["h_abs"] = max(c["h_abs"], min_step)
        elif c["h_abs"] < min_step:
            c["status"] = -1
            return
        if c["max_attempts"] > 0 and c["attempts"] >= c["max_attempts"]:
            c["status"] = 2
            return
        t_new = c["t"] + c["h_abs"]
        if t_new - c["t1"] > 0:
            t_new = c["t1"]
        c["t_new"], c["h"] = t_new, t_new - c["t"]
        c["h_abs"] = abs(c["h"])
        c["attempts"] += 1

    @staticmethod
    def _combine(recs, nrec):
        r = recs.numpy().reshape(nrec, 8)
        return np.concatenate([[r[:, 0].sum()], r[:, 1:5].min(axis=0), r[:, 5:].max(axis=0)])

    def init_control(self, recs, nrec, t0, t1, first_step, rtol, atol, max_attempts):
        self.c = dict(t=t0, t1=t1, h_abs=first_step, rtol=rtol, atol=atol, max_attempts=max_attempts, attempts=0,
                      rejected=False, cur=0, status=0 if t0 == t1 else 1, n_acc=0, n_rej=0, nfev=1)
        if self.c["status"] == 1:
            self._prepare()

    def attempt(self, rec):
        c = self.c
        if c["status"] != 1:
            return
        cur, h, o = c["cur"], c["h"], self.own()
        y, K = self.Y[cur], [self.F[cur]]
        for s in range(1, 6):
            K.append(self._rhs_buf(y + sum(a * k for a, k in zip(A[s], K)) * h))
        yn = y + h * sum(b * k for b, k in zip(B, K))
        K.append(self._rhs_buf(yn))
        err = h * sum(e * k for e, k in zip(E, K))
        scale = c["atol"] + np.maximum(np.abs(y), np.abs(yn)) * c["rtol"]
        self.Y[cur ^ 1][:, o], self.F[cur ^ 1][:, o] = yn[:, o], K[6][:, o]
        rec.copy_(torch.from_numpy(self._record(yn[:, o], float(np.sum((err[:, o] / scale[:, o]) ** 2)))))

    def control(self, recs, nrec):
        c = self.c
        if c["status"] != 1:
            return
        r = self._combine(recs, nrec)
        norm = math.sqrt(r[0]) / math.sqrt(5 * self.N)
        c["nfev"] += 6
        with np.errstate(all="ignore"):
            f = 0.9 * norm ** -0.2 if norm != 0 else np.inf      # NaN norm -> NaN factor -> "max(0.2, nan)" = 0.2
        if norm < 1:
            factor = 10.0 if norm == 0 else min(10.0, f)
            if c["rejected"]:
                factor = min(1.0, factor)
            c["h_abs"] *= factor
            c["rejected"] = False
            c["n_acc"] += 1
            c["t"] = c["t_new"]
            c["cur"] ^= 1
            if c["t"] - c["t1"] >= 0:
                c["status"] = 0
        else:
            c["h_abs"] *= f if f > 0.2 else 0.2
            c["rejected"] = True
            c["n_rej"] += 1
        if c["status"] == 1:
            self._prepare()

    def status(self):
        from marlpde_amd._abi import MarlStats
        st = MarlStats()
        st.status, st.n_accepted, st.n_rejected, st.nfev, st.t, st.h_next = (
            self.c["status"], self.c["n_acc"], self.c["n_rej"], self.c["nfev"], self.c["t"], self.c["h_abs"])
        return st

    def close(self):
        pass
