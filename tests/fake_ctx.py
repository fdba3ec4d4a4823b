"""CPU stand-in for mixmogam_amd._lib.Context, for the -m "not gpu" tests of the HOST logic only
(REML grid/secant search, scan preparation, MLMM stepping, chunked drivers).  Test infrastructure:
every method is a few lines of numpy/scipy; nothing in the product imports this."""
import numpy as np
from scipy import linalg, stats

from mixmogam_amd import _lib


class FakeGeno(_lib.Geno):
    def __init__(self, ctx, snps):
        self.ctx = ctx
        self.data = np.ascontiguousarray(snps, dtype=np.int8)
        self.M, self.N = self.data.shape
        self.h = None

    def upload_packed(self, packed, bits=1, m0=0, lut=None):
        vals = _lib.unpack_genotypes(packed, self.N, bits)
        if lut is not None:
            vals = np.asarray(lut, dtype=np.int8)[vals]
        self.data[m0:m0 + len(vals)] = vals
        return self

    def download(self, m0=0, rows=None):
        rows = self.M - m0 if rows is None else rows
        return self.data[m0:m0 + rows].copy()

    def download_rows(self, idx):
        return self.data[np.asarray(idx)].copy()

    def snp_stats(self):
        return self.data.mean(1), self.data.std(1)

    def matvec(self, V):
        return np.atleast_2d(V) @ self.data.T.astype(np.float64)

    def close(self):
        pass


class FakeAcc(object):
    def __init__(self, n):
        self.c, self.n = np.zeros((n, n)), 0

    def allreduce(self, comm):
        if comm is not None:
            self.c = comm.allreduce(self.c, "sum").reshape(self.c.shape)
            self.n = int(comm.allreduce(np.array([self.n], dtype=np.int64), "sum")[0])

    def add_grm(self, g):
        s = g.data.astype(np.float64)
        sd = s.std(1)
        if np.any(sd == 0):
            raise ValueError("monomorphic SNP (std == 0)")
        z = (s - s.mean(1, keepdims=True)) / sd[:, None]
        self.c += z.T @ z
        self.n += g.M

    def add(self, g, scale=None, shift=None):
        s = g.data.astype(np.float64)
        x = 2 * s - 1 if scale is None else s * np.asarray(scale)[:, None] + np.asarray(shift)[:, None]
        self.c += x.T @ x
        self.n += g.M

    def fetch(self):
        return self.c.copy(), self.n

    def close(self):
        pass


class FakeRot(object):
    def __init__(self, V):
        self.V, self.T, self.M = V, None, 0

    def load(self, g):
        self.T = self.V @ g.data.T.astype(np.float64)
        self.M = g.M
        return self

    def fetch(self, m0=0, rows=None):
        return self.T[:, m0:(self.M if rows is None else m0 + rows)].copy()

    def close(self):
        pass


class FakeReml(object):
    """numpy stand-in for _lib.Reml (mmg_reml_*): the four likelihood sums from H = K + delta I by dense solves."""

    def __init__(self, ctx, K, X, y):
        self.ctx, self.K, self.X, self.y = ctx, np.asarray(K, float), np.asarray(X, float), np.asarray(y, float).reshape(-1)
        xtx = self.X.T @ self.X
        xty = self.X.T @ self.y
        self.sse = float(self.y @ self.y - xty @ np.linalg.solve(xtx, xty))
        self.logdet_xtx = np.linalg.slogdet(xtx)[1]

    def _point(self, delta):
        H = self.K + delta * np.eye(len(self.K))
        Hi = np.linalg.inv(H)
        HiX, Hiy = Hi @ self.X, Hi @ self.y
        a = self.X.T @ HiX
        beta = np.linalg.solve(a, self.X.T @ Hiy)
        Py = Hiy - HiX @ beta
        P = Hi - HiX @ np.linalg.solve(a, HiX.T)
        return (float(self.y @ Py), np.linalg.slogdet(H)[1] + np.linalg.slogdet(a)[1] - self.logdet_xtx,
                float(Py @ Py), float(np.trace(P)), beta, Py, P)

    def band_factor(self, deltas):
        """mmg_reml_band_factor keeps factors; the stand-in has none to keep (its sums cost the same either way)."""
        self.factored = [float(d) for d in np.asarray(deltas).reshape(-1)]

    def sums(self, deltas):
        pts = [self._point(d) for d in np.asarray(deltas).reshape(-1)]
        return tuple(np.array([p[k] for p in pts]) for k in range(4)) + (self.sse,)

    def sums_ml(self, deltas, route="auto"):
        out = []
        for d in np.asarray(deltas).reshape(-1):
            H = self.K + d * np.eye(len(self.K))
            p = self._point(d)
            out.append((p[0], p[2], np.linalg.slogdet(H)[1], float(np.trace(np.linalg.inv(H)))))
        return tuple(np.array([o[k] for o in out]) for k in range(4))

    def scan_model(self, delta, ndigits=0):
        s1, _s2, _s3, _s4, beta, Py, P = self._point(delta)
        self.ctx.scan_set_model(P, Py)
        return s1, beta

    def uses_band(self, route="auto"):
        return False

    def linv(self, delta):
        """L^-1 of K + delta I = L L' (mmg_reml_linv_fetch)."""
        L = np.linalg.cholesky(self.K + delta * np.eye(len(self.K)))
        return np.linalg.inv(L)

    def linv_apply(self, delta, V, trans=False):
        H = self.linv(delta)
        return (H.T if trans else H) @ np.asarray(V, dtype=np.float64)

    def perm_plan(self, delta, Ys, h0_rss, centre_snps=True, centre_H=False):
        H = self.linv(delta)
        if centre_H:
            H = H - H.mean(axis=0, keepdims=True)
        return self.ctx.perm_plan(H, Ys, h0_rss, centre_snps=centre_snps)

    def close(self):
        pass


class FakeContext(object):
    device = 0

    def reml(self, K, X, y):
        return FakeReml(self, K, X, y)

    def geno(self, snps=None, M=None, N=None):
        return FakeGeno(self, snps if snps is not None else np.zeros((M, N), dtype=np.int8))

    def kinship_ibs_counts(self, g, comm=None):
        x = 2 * g.data.astype(np.int64) - 1
        c = x.T @ x
        return comm.allreduce(c, "sum") if comm is not None else c     # comm: a torch_coll.TorchCollectives here

    def kinship_indicator_counts(self, g, thr):
        u = (g.data >= thr).astype(np.int64)
        return u.T @ u

    def kinship_affine(self, g, scale=None, shift=None):
        a = FakeAcc(g.N)
        a.add(g, scale, shift)
        return a.c

    def kinship_accumulator(self, n):
        return FakeAcc(n)

    def eigh(self, A, vectors=True):
        vals, vecs = linalg.eigh(np.asarray(A, dtype=np.float64))
        return vals, (vecs.T.copy() if vectors else None)

    def dgemm(self, A, B, ta=False, tb=False):
        return (A.T if ta else A) @ (B.T if tb else B)

    def scan_set_model(self, A, w, ndigits=0):
        self.A, self.w = np.asarray(A, dtype=np.float64), np.asarray(w, dtype=np.float64).reshape(-1)

    def scan(self, g, h0_rss, df2, fetch=True, stats=False, out=None):
        S = g.data.astype(np.float64)
        dot = S @ self.w
        den = np.einsum('ij,ij->i', S @ self.A, S)
        dd = (S * S) @ np.diag(self.A)
        ok = (den > 1e-7 * dd) & (den > 0)
        rss = np.where(ok, h0_rss - dot * dot / np.where(ok, den, 1.0), h0_rss)
        F = (h0_rss / rss - 1.0) * df2
        res = {"rss": rss, "f_stats": F, "ps": self.f_sf(F, df2)}
        if stats:
            res.update(dot=dot, den=den, sum=S.sum(1))
        return res

    def f_sf(self, F, df2):
        return stats.f.sf(np.asarray(F, dtype=np.float64), 1, df2)

    def perm_plan(self, H, Ys, h0_rss, centre_snps=True):
        ctx = self

        class _Plan(object):
            def run(self, g, comm=None, after_scan_HtQ=None):
                return ctx.perm(g, H, Ys, h0_rss, comm=comm, centre_snps=centre_snps)

            def close(self):
                pass
        return _Plan()

    def rot(self, evecs_rows, M_cap):
        return FakeRot(np.asarray(evecs_rows, dtype=np.float64))

    def scan_multi(self, rot, d, omega, G, h0_rss, df2, want=("rss", "f_stats", "ps")):
        T = rot.T                                                    # N x M
        h0 = np.asarray(h0_rss, dtype=np.float64)[:, None]
        aq = np.asarray(d) @ (T * T)
        den = aq - sum((np.asarray(G)[:, c, :] @ T) ** 2 for c in range(np.asarray(G).shape[1]))
        dot = np.asarray(omega) @ T
        ok = (den > 1e-7 * aq) & (den > 0)
        rss = np.where(ok, h0 - dot * dot / np.where(ok, den, 1.0), h0)
        F = (h0 / rss - 1.0) * df2
        return {"rss": rss, "f_stats": F, "ps": self.f_sf(F, df2)}

    def perm(self, g, H, Ys, h0_rss, ndigits=0, comm=None, after_scan_HtQ=None, centre_snps=True):
        S = g.data.astype(np.float64)
        if centre_snps:
            S = S - S.mean(1, keepdims=True)
        T = S @ np.asarray(H).T
        tt = np.einsum('ij,ij->i', T, T)
        G = T @ np.asarray(Ys)
        ok = tt > 1e-12 * max(tt.max(), 1e-300)
        stat = np.where(ok[:, None], G * G / np.where(ok, tt, 1.0)[:, None], 0.0)
        mx = stat.max(0) if len(S) else np.zeros(np.asarray(Ys).shape[1])
        if comm is not None:
            mx = comm.allreduce(mx, "max")
        return np.minimum(h0_rss, np.einsum('ij,ij->j', Ys, Ys) - mx)
