"""TEST INFRASTRUCTURE — numpy stand-in for one shard (same step interface as HipShardBackend in
sbm-bp_amd/distributed.py) so the sharding plan, the halo exchange and the convergence driver of
ShardedBP can be exercised on CPU (gloo, world_size 2) without a GPU. It restates the
marginal-gather sweep of csrc/kernels.h (k_sweep_psi) with vectorised numpy; never shipped."""
import numpy as np
import torch


class State:
    def __init__(self):
        self.maxdiff, self.conv_iter, self.sweep_idx, self.stop, self.last_exact = 0.0, -1, 0, 0, 1


class NumpyShardBackend:
    def __init__(self, plan, Q, dc):
        self.plan, self.Q, self.dc = plan, Q, dc
        n_tab = plan.n_own + plan.n_halo
        self.psi = torch.zeros((2, n_tab, Q), dtype=torch.float64)
        self.red = torch.zeros(8192, dtype=torch.float64)
        self.ncomp = Q - 1  # compressed halo payload, as the HIP backend's default
        self.sendbuf = torch.zeros((max(1, len(plan.send_idx_chunked)), self.ncomp), dtype=torch.float64)
        self.recvbuf = torch.zeros((max(1, plan.n_halo), self.ncomp), dtype=torch.float64)
        self.send_views = [self.sendbuf[int(plan.send_off_c[c]):int(plan.send_off_c[c + 1])] for c in range(plan.n_chunks)]
        self.recv_views = [self.recvbuf[int(plan.stage_off_c[c]):int(plan.stage_off_c[c + 1])] for c in range(plan.n_chunks)]
        self.M = [np.zeros((plan.n_edges, Q)), np.zeros((plan.n_edges, Q))]
        self.cur = self.pcur = 0
        self.st = State()
        self.src = np.repeat(np.arange(plan.n_own), plan.deg)
        self.armed = -1.0

    # -- state -----------------------------------------------------------------------------------
    def set_state(self, psi, msg):
        self.psi[self.pcur][:self.plan.n_own] = torch.from_numpy(np.asarray(psi, dtype=np.float64))
        self.M[0][:] = msg  # (psi^0, m^-1): sweep 0 reads the buffer it then overwrites
        self.M[1][:] = msg

    def get_state(self):
        return self.psi[self.pcur][:self.plan.n_own].numpy().copy(), self.M[self.cur].copy()

    def set_params(self, cab, na, beta):
        self.cab = np.asarray(cab, dtype=np.float64)
        self.W = self.cab ** beta if self.dc == 0 else self.cab
        self.eta = np.asarray(na, dtype=np.float64) / self.plan.n_global
        self.beta = beta

    # -- steps -------------------------------------------------------------------------------------
    def begin(self, crit):
        self.crit = crit
        self.exact = False
        self.prev_hint = 0.0
        self.st = State()

    def set_schedule(self, field_mix, check_every=1):
        assert field_mix == 1.0 or True  # the stand-in has no field relaxation (not exercised by the CPU tests)

    def read_buffer(self, j):
        return (self.pcur + j) & 1

    def recv_view(self, j, c):
        return self.recv_views[c]

    def pack(self, j, c):
        off, n = int(self.plan.send_off_cp[c, 0]), int(self.plan.send_counts_cp[c].sum())
        if n:
            idx = torch.from_numpy(self.plan.send_idx_chunked[off:off + n])
            self.sendbuf[off:off + n] = self.psi[self.read_buffer(j)][idx][:, :self.ncomp]

    def unpack(self, j):
        nh = self.plan.n_halo
        if nh:
            tab = self.psi[self.read_buffer(j)]
            dst = torch.from_numpy(self.plan.n_own + self.plan.stage_to_halo)
            tab[dst, :self.ncomp] = self.recvbuf[:nh]
            tab[dst, self.ncomp] = torch.clamp(1.0 - self.recvbuf[:nh].sum(1), min=0.0)

    def _g(self):
        return self.plan.deg.astype(np.float64) if self.dc else np.ones(self.plan.n_own)

    def field_partial(self, j):
        p = self.psi[self.read_buffer(j)][:self.plan.n_own].numpy()
        self.red[:self.Q] = torch.from_numpy((self._g()[:, None] * p).sum(0))
        self.red[self.Q] = 0.0

    def sweep_chunk(self, j, c):
        if self.st.stop:
            return
        pl = self.plan
        if c == 0:
            self._S, self._md = np.zeros(self.Q), 0.0
        r0, r1 = int(pl.chunk_row[c]), int(pl.chunk_row[c + 1])
        e0, e1 = int(pl.row_ptr[r0]), int(pl.row_ptr[r1])
        Mio = self.M[((self.cur + j) & 1) ^ 1][e0:e1]
        pold = self.psi[(self.pcur + j) & 1].numpy()
        pnew = self.psi[((self.pcur + j) & 1) ^ 1].numpy()
        src = self.src[e0:e1] - r0
        deg = pl.deg[r0:r1]
        bo = Mio @ self.W
        inc = pold[pl.nbr_local[e0:e1].astype(np.int64)] / bo
        inc /= inc.sum(1, keepdims=True)
        logb = np.log(inc @ self.W)
        logA = np.zeros((r1 - r0, self.Q))
        np.add.at(logA, src, logb)
        if self.dc == 0:
            logA += (np.log(self.eta) - self.beta * self.hN)[None, :]
        else:
            logA += np.log(self.eta)[None, :] - deg[:, None] * self.hN[None, :]
        A = np.exp(logA - logA.max(1, keepdims=True)) if r1 > r0 else logA
        psi_new = A / A.sum(1, keepdims=True) if r1 > r0 else A
        cav = logA[src] - logb
        cav = np.exp(cav - cav.max(1, keepdims=True)) if e1 > e0 else cav
        new = cav / cav.sum(1, keepdims=True) if e1 > e0 else cav
        if e1 > e0:
            ref = self.M[(self.cur + j) & 1][e0:e1] if getattr(self, "exact", False) else Mio  # exact 1-step difference, or the 2-step hint
            self._md = max(self._md, float(np.abs(new - ref).max()))
        Mio[:] = new
        pnew[r0:r1] = psi_new
        g = deg.astype(np.float64) if self.dc else np.ones(r1 - r0)
        self._S += (g[:, None] * psi_new).sum(0)

    def sweep_fold(self):
        if self.st.stop:
            return
        self.red[:self.Q] = torch.from_numpy(self._S)
        self.red[self.Q] = self._md

    def finalize(self, mode, n_rows):
        if mode == 0 and self.st.stop:
            return
        rows = self.red[32:32 + n_rows * (self.Q + 1)].numpy().reshape(n_rows, self.Q + 1)
        S = np.zeros(self.Q)
        for r in range(n_rows):  # fixed order, as k_finalize
            S += rows[r, :self.Q]
        h = self.cab.T @ S
        self.hN = h / self.plan.n_global
        if mode == 0:
            md = float(rows[:, self.Q].max())
            self.st.maxdiff = md
            self.st.last_exact = int(self.exact)
            if not self.exact:  # a 2-step hint can only arm the exact criterion (k_finalize: scale from the measured rate)
                scale, prev = 8.0, getattr(self, "prev_hint", 0.0)
                if prev > 0.0 and 0.0 < md < prev:
                    r = md / prev
                    scale = min(64.0, max(8.0, 1.5 * (1.0 + 1.0 / r) / r))
                self.prev_hint = md
                if md < scale * self.crit:
                    self.exact = True
            elif md < self.crit and self.st.conv_iter < 0:
                self.st.conv_iter, self.st.stop = self.st.sweep_idx, 1
            self.st.sweep_idx += 1

    def msgdiff_partial(self):
        self.red[0] = float(np.abs(self.M[0] - self.M[1]).max()) if self.plan.n_edges else 0.0

    def rowsums_partial(self):
        Q = self.Q
        p = self.psi[self.pcur][:self.plan.n_own].numpy()
        out = np.zeros(2 * Q + Q * Q)
        out[:Q] = p.sum(0)
        out[Q:2 * Q] = (self.plan.deg[:, None] * p).sum(0)
        for a in range(Q):
            out[2 * Q + a * Q:2 * Q + (a + 1) * Q] = p[self.true_conf == a].sum(0)
        self.red[:len(out)] = torch.from_numpy(out)

    def poll(self):
        import copy
        return copy.copy(self.st)

    def commit(self, n):
        self.cur = (self.cur + n) & 1
        self.pcur = (self.pcur + n) & 1

    def sync(self):
        pass

    def init_from_global(self, psi_global, msg_global, true_conf_global):
        p = self.plan
        self.set_state(psi_global[p.row0:p.row0 + p.n_own], msg_global[p.edge0:p.edge0 + p.n_edges])
        self.true_conf = np.asarray(true_conf_global)[p.row0:p.row0 + p.n_own]
