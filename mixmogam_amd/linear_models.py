"""EMMAX mixed-model association -- the reference's linear_models.py call surface for the hot
path (emmax, get_emma_reml_estimates, emmax_perm_test, LinearMixedModel), with the O(N^3) and
O(N^2 M) arithmetic on the MI355X through libmixmogam_hip.so.

Line references are into /root/reference/linear_models.py.  What runs where:
  device  eigh of K and of S(K+I)S (:594,:613); A = Mp Mp^T (:1303 folded into the closed form);
          the per-SNP scan (:1316-1349) incl. F and p-values; the permutation GEMM (:1157-1164).
  host    O(N q), O(N^2 q) and O(N * grid) glue in float64: REML grid + secant search
          (:796-891), H_sqrt_inv scaling (:898), QR of the N x q null design (:1300).
Arithmetic is float64 end to end (the "double-promoted" reading of the reference, SURVEY 7/8c:
its literal 'single' storage limits its own p-values to ~1e-3 relative).
"""
import time
import os
import warnings

import numpy as np
from scipy import linalg, optimize

from . import _lib, kinship


def _col(x, n=None):
    a = np.asarray(x, dtype=np.float64).reshape(-1)
    if n is not None and len(a) != n:
        raise ValueError("expected length %d, got %d" % (n, len(a)))
    return a


class LinearModel(object):
    """linear_models.py:81 -- the K-free twin of the EMMAX scan: fast_f_test is the same
    device scan with H = I, i.e. A = I - QQ' and w = (I - QQ') y."""

    def __init__(self, Y=None, ctx=None):
        self.n = len(Y)
        self.Y = _col(Y).reshape(self.n, 1)
        self.X = np.ones((self.n, 1))
        self.p = 1
        self.beta_est = None
        self.cofactors = []
        self._ctx = ctx

    @property
    def ctx(self):
        if self._ctx is None:
            self._ctx = _lib.get_context()
        return self._ctx

    def add_factor(self, x, lin_depend_thres=1e-8):
        """:98-113."""
        new_x = _col(x, self.n)
        (beta, rss, rank, sigma) = linalg.lstsq(self.X, new_x)
        if float(np.sum((new_x - self.X @ beta) ** 2)) < lin_depend_thres:
            warnings.warn('A factor was found to be linearly dependent on the factors already in the X '
                          'matrix.  Hence skipping it!')
            return False
        self.X = np.hstack([self.X, new_x.reshape(self.n, 1)])
        self.cofactors.append(x)
        self.p += 1
        return True

    def get_rss(self, dtype='double'):
        """:178-184."""
        (betas, _r, r, s) = linalg.lstsq(self.X, self.Y.reshape(-1))
        return float(np.sum((self.Y.reshape(-1) - self.X @ betas) ** 2))

    def get_ll(self, rss=None, dtype='double'):
        """:187-193."""
        if not rss:
            rss = self.get_rss(dtype)
        return (-self.n / 2) * (1 + np.log(2 * np.pi) + rss / self.n)

    def fast_f_test(self, snps, verbose=True, Z=None, with_betas=False, ndigits=0):
        """:196-257 -- per-SNP F test of y ~ X + snp.  M = I - QQ' (:215-218) is idempotent, so
        the closed form of the scan applies with A = M and w = M y.
        with_betas (:220-221,:233-239): the reference regresses the RESIDUAL r = y - X b0 on [X, s] per SNP and reports those
        coefficients -- in closed form b_snp = (s.r) / (s'Ms) and, for the covariates, -(X'X)^-1 X's b_snp (r is
        orthogonal to X); a SNP that leaves the design rank deficient keeps h0_betas and rss = h0_rss (:236-239).
        Z is accepted and, as in the reference (:196, never read in the body), ignored."""
        ctx = self.ctx
        y = self.Y.reshape(-1)
        (h0_betas, _r, h0_rank, h0_s) = linalg.lstsq(self.X, y)          # :210
        r = y - self.X @ h0_betas
        h0_rss = float(r @ r)
        (Q, R) = linalg.qr(self.X, mode='economic')                       # :214
        n_p = self.n - (self.X.shape[1] + 1)
        own = not isinstance(snps, _lib.Geno)
        g = ctx.geno(kinship._as_snp_matrix(snps)) if own else snps
        res = None
        try:
            if LINEAR_MODEL_LOW_RANK and ndigits == 0 and isinstance(ctx, _lib.Context):
                # M = I - QQ' has rank-q structure: s'Ms = sum s^2 - |Q's|^2 and s.My = s.r are 1 + q dot products per SNP --
                # one bandwidth-bound pass over the store in fp64 (mmg_geno_matvec) instead of an N x N matrix built on the
                # host, quantised and run through the quadratic-form GEMM (94 -> 30 ms at N = 5000 x M = 200,000, upload
                # included).  The rank rule of the scan's finalize kernel (den <= 1e-7 of the diagonal form: rss = h0_rss,
                # :236-239) is kept against sum s^2.
                mean, sd = g.snp_stats()
                ssq = self.n * (sd * sd + mean * mean)
                dots = g.matvec(np.vstack([r[None, :], Q.T]))             # [1 + q x M]
                den = ssq - np.einsum('cm,cm->m', dots[1:], dots[1:])
                ok_den = (den > 1e-7 * ssq) & (den > 0.0)
                rss_v = np.where(ok_den, h0_rss - dots[0] ** 2 / np.where(ok_den, den, 1.0), h0_rss)
                f_v = (h0_rss / rss_v - 1.0) * n_p
                out = {'rss': rss_v, 'f_stats': f_v, 'ps': ctx.f_sf(f_v, n_p), 'dot': dots[0], 'den': den}
            else:
                A = np.eye(self.n) - Q @ Q.T                              # :218
                ctx.scan_set_model(A, r, ndigits)
                out = ctx.scan(g, h0_rss, n_p, stats=with_betas)
            res = {'ps': out['ps'], 'f_stats': out['f_stats'], 'rss': out['rss'], 'var_perc': 1 - out['rss'] / h0_rss,
                   'h0_rss': np.array([h0_rss]), 'h0_betas': [float(b) for b in h0_betas]}
            if with_betas:
                ok = out['rss'] != h0_rss
                b_snp = np.where(ok, out['dot'] / np.where(ok, out['den'], 1.0), 0.0)
                Cs = g.matvec(linalg.solve_triangular(R, Q.T))           # (X'X)^-1 X' s for every SNP: q x M
                res['betas'] = [([float(-Cs[k, j] * b_snp[j]) for k in range(Cs.shape[0])] + [float(b_snp[j])]) if ok[j]
                                else list(res['h0_betas']) for j in range(g.M)]
        finally:
            if own:
                g.close()
        return res


# LinearModel.fast_f_test: dot products against Q and the residual instead of the N x N projection through the scan GEMM
LINEAR_MODEL_LOW_RANK = True
# Where get_estimates would compute eig_R itself, take the spectral sums from eig_L instead (no second eigh).
REML_SUMS_FROM_EIG_L = True
# above this many individuals emmax_f_test takes the eigendecomposition-free route when nothing needs H_sqrt_inv.
# Mandatory beyond N = 46,340 (rocSOLVER's dsyevd indexes with 32 bits: N^2 < 2^31).  Round 4: the route wins from the
# smallest N at which the band reduction of K exists at all (csrc/reml_band.hip: Cholesky-QR panels on own kernels, the
# secant search on an interpolant of the sums, own blocked Cholesky for the scan model) -- tools/reml_vs_eigh.py, emmax()
# on resident genotypes, eigen route / this route: N = 300 9.9 / 5.6 ms, 1000 91 / 17 ms, 2000 100 / 48 ms, 5000 324 /
# 126 ms, 8192 957 / 297 ms (profiles/r4_reml_vs_eigh.txt); round 3 had the crossover at 5000.  Below N = 256
# mmg_reml_sums takes one Cholesky factorisation per delta (no band to reduce to) -- since round 5 on this library's own kernels
# like everything else of the route up to N = 8192 (csrc/reml_chol.hip: reml_own_route), so the smallest data sets take it
# too: a process that calls emmax() on the bundled 199 accessions never loads rocSOLVER / rocBLAS (10-23 s of a first call on a
# cold box, profiles/r4_small_configs.txt).  MMG_EIGEN_FREE_MIN_N overrides (256: round 4).
EIGEN_FREE_MIN_N = int(os.environ.get("MMG_EIGEN_FREE_MIN_N", 15))
EIGH_MAX_N = 46340
# emmax_f_test builds the scan model on the device from K and delta (mmg_reml_scan_model) when nothing needs H itself.
DEVICE_SCAN_MODEL = True


def perm_h_from_cholesky(ctx):
    """Whether the callers that need `H_sqrt_inv` and were not handed one take H = L^-1 of K + delta I = L L' from the device
    (round 5) instead of diag((lambda + delta)^-1/2) U' from eigh(K): the default; MMG_PERM_H=eigen keeps the literal matrix
    (identical in distribution, not draw by draw: the shuffled residuals live in the basis of the H that is used)."""
    return os.environ.get('MMG_PERM_H', '') != 'eigen' and hasattr(ctx, 'reml')


class _SpectralSumsR(object):
    """The four sums of the EMMA likelihood from the eigen-pairs of S(K+I)S (linear_models.py:794-810)."""

    def __init__(self, eig_R, y, p):
        self.eig_vals = np.asarray(eig_R['values'], dtype=np.float64)
        assert len(self.eig_vals) == p, 'Number of eigenvalues is incorrect.'
        etas = np.asarray(eig_R['vectors']) @ y                          # :794
        self.sq_etas = etas * etas
        self.sum_sq_etas = float(np.sum(self.sq_etas))

    def at(self, deltas):
        lambdas = self.eig_vals[:, None] + deltas[None, :]
        s1 = np.sum(self.sq_etas[:, None] / lambdas, axis=0)
        s3 = np.sum(self.sq_etas[:, None] / (lambdas * lambdas), axis=0)
        return s1, np.sum(np.log(lambdas), axis=0), s3, np.sum(1 / lambdas, axis=0)


class _SpectralSumsL(object):
    """The same sums from eig_L alone: rotate y and X once (O(N^2 q)), then O(N q^2) per delta."""

    def __init__(self, eig_L, X, y, rot=None):
        self.lam = np.asarray(eig_L['values'], dtype=np.float64)
        if rot is not None:                                              # (U y, U X) supplied by the caller
            self.yt, self.Xt = rot
        else:
            U = np.asarray(eig_L['vectors'], dtype=np.float64)           # rows are eigenvectors
            self.yt = U @ y
            self.Xt = U @ X
        XtX = X.T @ X
        self.logdet_xtx = np.linalg.slogdet(XtX)[1]
        xty = X.T @ y
        self.sum_sq_etas = float(y @ y - xty @ np.linalg.solve(XtX, xty))   # |Sy|^2
        q = self.Xt.shape[1]
        # the per-delta contractions as plain products with d' (m x N): einsum(..., optimize=True) spent 0.4 ms per call on
        # its path search -- half of the per-SNP cost of the exact-EMMA refinement (emma_num), which builds one of these per SNP
        self._xx = (self.Xt[:, :, None] * self.Xt[:, None, :]).reshape(len(self.Xt), q * q)
        self._xy = self.Xt * self.yt[:, None]

    def at(self, deltas):
        d = 1.0 / (self.lam[:, None] + deltas[None, :])                  # N x m
        Xt, yt = self.Xt, self.yt
        q = Xt.shape[1]
        a = (d.T @ self._xx).reshape(-1, q, q)                           # X'H^-1 X
        a2 = ((d * d).T @ self._xx).reshape(-1, q, q)                    # X'H^-2 X
        b = d.T @ self._xy                                               # X'H^-1 y
        c = (yt * yt) @ d                                                # y'H^-1 y
        x = np.linalg.solve(a, b[:, :, None])[:, :, 0]                   # m x q
        s1 = c - np.einsum('mi,mi->m', b, x)
        z = d * (yt[:, None] - Xt @ x.T)                                 # P y in the eigenbasis
        s3 = np.einsum('nm,nm->m', z, z)
        s2 = np.sum(np.log(self.lam[:, None] + deltas[None, :]), axis=0) + np.linalg.slogdet(a)[1] - self.logdet_xtx
        s4 = np.sum(d, axis=0) - np.trace(np.linalg.solve(a, a2), axis1=1, axis2=2)
        return s1, s2, s3, s4


class _SpectralSumsChol(object):
    """The same four sums with NO eigendecomposition, on the device (_lib.Reml / mmg_reml_sums).  For N beyond
    rocSOLVER's syevd index range (N > 46,340), where eigh falls back to block Jacobi (6.6 min at N = 50,000), this is
    the cheaper route.  Two forms (mmg_reml_sums_ex):
      band  K is reduced once to an orthogonally similar band matrix (csrc/reml_band.hip: 6.5 s at N = 50,000) and
            every delta of the ~57 the likelihood search asks for costs a banded factorisation (0.3 s per call for any
            number of deltas).  Every rank does the same arithmetic on the same K: no exchange at all.
      chol  one Cholesky factorisation + triangular inverse of K + delta I per delta (csrc/reml_chol.hip, 1.6 s each at
            N = 50,000); coll: the grid values are dealt out to the ranks (every rank holds K), the sums all-gathered."""

    # Chebyshev nodes of the local model of the sums over a bracket of the likelihood search (prepare_interval)
    INTERP_NODES = 16
    INTERP_MARGIN = 0.3                                                  # in log(delta), either side of the bracket
    # Round 5: the search FACTORS once.  The grid the search starts from (:814-830: 51 or 101 values equispaced in
    # log(delta)) goes to the device refined to a spacing of ~0.1 with FINE_PAD extra nodes beyond either end -- up to
    # FINE_MAX variance ratios, one workgroup each: the factor sweep takes 4 ms at N = 5000 for 227 as for 51
    # (mmg_reml_band_factor); the substitutions and the trace recurrence (3.6 ms) run on the grid, and on the ~40 refined nodes
    # around the bracket if there is one -- and the model over the bracket is the polynomial through the FINE_STENCIL nearest nodes around the
    # question, which therefore always sits in the central cell of its stencil.  The sums are analytic in u = log(delta) for
    # |Im u| < pi: the error of that polynomial is at most M_r (h / r)^20 (0.5 * 1.5 * ... * 9.5)^2 = 1.2e-18 M_3 for
    # h = 0.1, r = 3 (M_r: the sum's size on the circle of radius r) -- below the rounding of the evaluations (1e-13), as the
    # 16 Chebyshev nodes were.  MMG_REML_FINE_GRID=0: the grid as it is asked for, then the 16 nodes (two full device calls).
    FINE_GRID = os.environ.get("MMG_REML_FINE_GRID", "1") != "0"
    FINE_STEP = 0.1
    FINE_STENCIL = 20
    FINE_PAD = 13                                                        # >= FINE_STENCIL / 2 + INTERP_MARGIN / FINE_STEP
    FINE_MAX = 256

    def __init__(self, reml, coll=None, route="auto"):
        self.reml, self.coll, self.route = reml, coll, route
        self.band = reml.uses_band(route) if hasattr(reml, "uses_band") else False
        self.sum_sq_etas = None
        self.n_factorisations = 0                                        # deltas evaluated on this rank
        self.n_calls = 0                                                 # device calls (each a latency chain of N steps)
        self._memo = {}
        self._interp = None
        self._fine = None                                                # refined grid: (first node, spacing, [4 arrays of sums], log nodes, nodes, evaluated?)

    def _fine_plan(self, deltas):
        """(R, u0, h) when `deltas` is a grid equispaced in log(delta) that can be refined to ~FINE_STEP within FINE_MAX
        variance ratios per call, else None."""
        if not (self.FINE_GRID and self.band and len(deltas) >= 8 and np.all(deltas > 0)):
            return None
        u = np.log(deltas)
        steps = np.diff(u)
        h = float(steps.mean())
        if not (h > 0 and np.max(np.abs(steps - h)) <= 1e-9 * max(1.0, abs(h))):
            return None
        R = max(1, int(round(h / self.FINE_STEP)))
        while R > 1 and (len(deltas) - 1) * R + 1 + 2 * self.FINE_PAD > self.FINE_MAX:
            R -= 1
        if h / R > 1.5 * self.FINE_STEP or (len(deltas) - 1) * R + 1 + 2 * self.FINE_PAD > self.FINE_MAX:
            return None
        return R, float(u[0]), h / R

    def _at_fine(self, deltas, plan):
        """The grid of the search together with its refinement; returns the sums at `deltas`.  With mmg_reml_band_factor
        (Reml.band_factor) the refined grid is FACTORED in one sweep -- as long for 227 variance ratios as for 51 -- and the sums
        are taken on the caller's grid only (substitutions + trace recurrence from the kept factors); the refined nodes around
        a bracket follow in prepare_interval, again from the kept factors.  A search whose optimum is a grid point thus costs
        what it always did, one with a bracket 4.1 + 3.6 + 3.6 ms at N = 5000 instead of two full calls of 7.5.  Without the
        entry point: every refined node in one call."""
        R, u0, hf = plan
        pad, m = self.FINE_PAD, len(deltas)
        k = np.arange(-pad, (m - 1) * R + pad + 1)
        fine = np.exp(u0 + hf * k)
        own = pad + R * np.arange(m)                                     # the caller's values, bit for bit, at their places
        fine[own] = deltas
        have = np.zeros(len(fine), dtype=bool)
        arrays = [np.full(len(fine), np.nan) for _ in range(4)]
        if hasattr(self.reml, "band_factor"):
            self.reml.band_factor(fine)
            self.n_factorisations += len(fine) - m                       # (_at_device counts the m it is asked for)
            vals = self._at(fine[own])
            for a, v in zip(arrays, vals[:4]):
                a[own] = v
            have[own] = True
        else:
            vals = self._at(fine)
            arrays = [np.asarray(v, dtype=np.float64) for v in vals[:4]]
            have[:] = True
            vals = tuple(a[own] for a in arrays)
        self._fine = (u0 - pad * hf, hf, arrays, np.log(fine), fine, have)
        return tuple(np.asarray(v) for v in vals[:4])

    def prepare_interval(self, d_lo, d_hi):
        """The secant search of get_estimates (:847) asks for the sums at one delta after another inside the bracket
        [d_lo, d_hi] -- on the band route every such question is a device call whose cost is a latency chain of N steps
        (9 ms at N = 5000), the same for 1 or 51 deltas.  So ONE call evaluates the four sums at 16 Chebyshev nodes of
        log(delta) over the bracket (+- 0.3), and the search runs on the interpolant: the sums are analytic in
        u = log(delta) in the strip |Im u| < pi (their poles sit at delta = -lambda_i), the interval's half width is 0.5,
        so the Bernstein ellipse through the nearest pole has rho = 12.6 and 16 nodes leave 12.6^-16 = 2e-18 -- below the
        rounding noise of the evaluations themselves (1e-13).  Questions outside the interval, and the final
        evaluation of the likelihood at the optimum (at_exact), go to the device."""
        if not self.band:
            return                                                       # the Cholesky route deals independent deltas over ranks
        if self._fine is not None:
            u_first, hf, vals, u, fine, have = self._fine
            lo, hi = np.log(d_lo) - self.INTERP_MARGIN, np.log(d_hi) + self.INTERP_MARGIN
            half = self.FINE_STENCIL // 2
            if lo >= u[half - 1] and hi <= u[len(u) - half]:             # every question in there has a full stencil around it
                first = max(int(np.searchsorted(u, lo, side='right')) - 1 - (half - 1), 0)
                last = min(int(np.searchsorted(u, hi, side='right')) - 1 + half, len(u) - 1)
                missing = np.nonzero(~have[first:last + 1])[0] + first
                if len(missing):                                         # the refined nodes around the bracket, from the kept factors
                    got = self._at(fine[missing])
                    for a, v in zip(vals, got[:4]):
                        a[missing] = v
                    have[missing] = True
                self._interp = (lo, hi, None, None, None)
                return
        n = self.INTERP_NODES
        lo, hi = np.log(d_lo) - self.INTERP_MARGIN, np.log(d_hi) + self.INTERP_MARGIN
        k = np.arange(n)
        x = np.cos(np.pi * (2 * k + 1) / (2 * n))                        # Chebyshev points of the first kind on [-1, 1]
        w = (-1.0) ** k * np.sin(np.pi * (2 * k + 1) / (2 * n))          # their barycentric weights
        u = 0.5 * (lo + hi) + 0.5 * (hi - lo) * x
        vals = self._at(np.exp(u))
        self._interp = (lo, hi, u, w, [np.asarray(v, dtype=np.float64) for v in vals])

    _FINE_W = None

    def _from_fine(self, delta):
        """The polynomial through the FINE_STENCIL nodes of the refined grid around log(delta), in barycentric form (equispaced
        nodes: w_j = (-1)^j C(n - 1, j))."""
        _u_first, hf, vals, u, _fine, _have = self._fine
        n = self.FINE_STENCIL
        if _SpectralSumsChol._FINE_W is None or len(_SpectralSumsChol._FINE_W) != n:
            from scipy.special import comb
            _SpectralSumsChol._FINE_W = np.array([(-1.0) ** j * comb(n - 1, j, exact=True) for j in range(n)], dtype=np.float64)
        w = _SpectralSumsChol._FINE_W
        t = np.log(delta)
        c = int(np.searchsorted(u, t, side='right')) - 1                 # u[c] <= t < u[c + 1]
        a = min(max(c - (n // 2 - 1), 0), len(u) - n)
        d = t - u[a:a + n]
        hit = np.nonzero(d == 0.0)[0]
        if len(hit):
            return tuple(np.array([v[a + hit[0]]]) for v in vals)
        cw = w / d
        return tuple(np.array([float(cw @ v[a:a + n] / cw.sum())]) for v in vals)

    def _from_model(self, delta):
        lo, hi, u, w, vals = self._interp
        if u is None:
            return self._from_fine(delta)
        t = np.log(delta)
        d = t - u
        hit = np.nonzero(d == 0.0)[0]
        if len(hit):
            return tuple(np.array([v[hit[0]]]) for v in vals)
        c = w / d
        return tuple(np.array([float(c @ v / c.sum())]) for v in vals)

    def at(self, deltas):
        deltas = np.asarray(deltas, dtype=np.float64).reshape(-1)
        if len(deltas) == 1:                                             # rell(opt), vg, ... ask for the same delta again
            key = float(deltas[0])
            if key in self._memo:
                return self._memo[key]
            if self._interp is not None and key > 0.0 and self._interp[0] <= np.log(key) <= self._interp[1]:
                return self._from_model(key)
            self._at(deltas)
            return self._memo[key]
        plan = self._fine_plan(deltas) if self._fine is None else None
        if plan is not None:
            return self._at_fine(deltas, plan)
        return self._at(deltas)

    def at_ml(self, deltas):
        """(s1, s3, log|K + delta I|, tr (K + delta I)^-1) per delta: the sums of the ML likelihood (:634-649, :821-826)."""
        deltas = np.asarray(deltas, dtype=np.float64).reshape(-1)
        self.n_calls += 1
        self.n_factorisations += len(deltas)
        return self.reml.sums_ml(deltas, self.route) if self.route != "auto" else self.reml.sums_ml(deltas)

    # The likelihood and vg at the optimum of the search (:882-896): from the interpolant when the optimum lies inside the
    # prepared bracket (2e-18 of interpolation error under 1e-13 of evaluation noise: the third device call bought nothing
    # but its 9 ms), from the device otherwise.  False: always from the device.
    FINAL_FROM_MODEL = True

    def at_exact(self, delta):
        """The sums at one delta as the search's final evaluation wants them: a remembered device value, the bracket's
        model (FINAL_FROM_MODEL), or a device call."""
        key = float(delta)
        if key in self._memo:
            return self._memo[key]
        if (self.FINAL_FROM_MODEL and self._interp is not None and key > 0.0
                and self._interp[0] <= np.log(key) <= self._interp[1]):
            return self._from_model(key)
        self._at(np.array([key], dtype=np.float64))
        return self._memo[key]

    def _remember(self, deltas, vals):
        """Every delta that went to the device is remembered: the optimum of a search without an interior bracket is a
        grid point, whose likelihood and vg (:882-896) are then already known."""
        for k, d in enumerate(deltas):
            self._memo.setdefault(float(d), tuple(np.array([v[k]]) for v in vals[:4]))
        return vals

    def _at(self, deltas):
        return self._remember(deltas, self._at_device(deltas))

    def _at_device(self, deltas):
        coll = self.coll
        self.n_calls += 1
        if coll is not None and coll.world > 1 and len(deltas) >= coll.world and not self.band:
            mine = np.arange(coll.rank, len(deltas), coll.world)
            # A factorisation may fail on SOME ranks only (an indefinite K: the smallest deltas sit on the low ranks).
            # Every rank must still enter the all-gather, so the failure travels as a flag row of the gathered block
            # and all ranks raise together afterwards (advisor r2: a rank that left early hung the others).
            failure = None
            try:
                part = self.reml.sums(deltas[mine])
            except _lib.MixmogamHipError as e:
                failure, part = e, None
            self.n_factorisations += len(mine)
            count = -(-len(deltas) // coll.world)
            blk = np.full((5, count), np.nan)
            blk[4, :] = 0.0 if failure is None else 1.0
            if part is not None:
                for k in range(4):
                    blk[k, :len(mine)] = part[k]
            allb = np.asarray(coll.allgather(blk.reshape(-1))).reshape(coll.world, 5, count)
            failed = [r for r in range(coll.world) if allb[r, 4, 0] != 0.0]
            if failed:
                if failure is not None:
                    raise failure
                raise _lib.MixmogamHipError("K + delta*I is not positive definite on rank(s) %s of the shared REML grid"
                                            % failed)
            out = [np.empty(len(deltas)) for _ in range(4)]
            for r in range(coll.world):
                idx = np.arange(r, len(deltas), coll.world)
                for k in range(4):
                    out[k][idx] = allb[r, k, :len(idx)]
            self.sum_sq_etas = part[4]
            return tuple(out)
        s1, s2, s3, s4, sse = self.reml.sums(deltas, self.route) if self.route != "auto" else self.reml.sums(deltas)
        self.n_factorisations += len(deltas)
        self.sum_sq_etas = sse
        return s1, s2, s3, s4


class _LazyIdentity(object):
    """The identity random effect of :574 (`random_effects[0] = ('normal', I)`): an N x N matrix nothing on the EMMAX path
    reads (only _get_eigen_R_ adds it to K), built when someone does -- np.eye(5000) is 7 ms of a 0.12 s emmax() call."""

    def __init__(self, n):
        self.n, self._m = n, None
        self.shape = (n, n)

    def __array__(self, dtype=None, copy=None):
        if self._m is None:
            self._m = np.eye(self.n)
        return self._m if dtype is None else self._m.astype(dtype, copy=False)

    def __getitem__(self, idx):
        return np.asarray(self)[idx]

    def __len__(self):
        return self.n


class LinearMixedModel(object):
    """linear_models.py:554 (and the parts of LinearModel :81 it inherits on this path)."""

    def __init__(self, Y=None, dtype='double', ctx=None):
        self.n = len(Y)
        self.Y = _col(Y).reshape(self.n, 1)                              # :566-567
        self.y_var = np.var(self.Y, ddof=1)
        self.X = np.ones((self.n, 1))                                    # :568 intercept
        self.p = 1
        self.beta_est = None
        self.cofactors = []
        self.random_effects = [('normal', _LazyIdentity(self.n))]       # :574
        self._ctx = ctx

    @property
    def ctx(self):
        if self._ctx is None:
            self._ctx = _lib.get_context()
        return self._ctx

    # ------------------------------------------------------------------ model building
    def add_random_effect(self, cov_matrix=None, effect_type='normal'):
        if effect_type != 'normal':
            raise Exception('Currently, only Normal random effects are allowed.')
        self.random_effects.append((effect_type, kinship.scale_k(cov_matrix)))   # :580

    def set_random_effect(self, cov_matrix_list, effect_types=None):
        self.random_effects = [('normal', _LazyIdentity(self.n))]
        for cov_matrix in cov_matrix_list:
            self.add_random_effect(cov_matrix=kinship.scale_k(cov_matrix))       # :586

    def add_factor(self, x, lin_depend_thres=1e-8):
        """:98-113 -- reject a cofactor that is linearly dependent on X."""
        new_x = _col(x, self.n)
        (beta, rss, rank, sigma) = linalg.lstsq(self.X, new_x)
        if float(np.sum((new_x - self.X @ beta) ** 2)) < lin_depend_thres:
            warnings.warn('A factor was found to be linearly dependent on the factors already in the X '
                          'matrix.  Hence skipping it!')
            return False
        self.X = np.hstack([self.X, new_x.reshape(self.n, 1)])
        self.cofactors.append(x)
        self.p += 1
        return True

    def set_factors(self, factors, include_intercept=True):
        """:116-130."""
        cols = [_col(f, self.n).reshape(self.n, 1) for f in factors]
        if include_intercept:
            self.X = np.hstack([np.ones((self.n, 1))] + cols)
            self.p = 1 + len(cols)
        else:
            self.X = np.hstack(cols)
            self.p = len(cols)

    # ------------------------------------------------------------------ eigen decompositions
    def _get_eigen_L_(self, K=None, dtype='double'):
        """:589-596 -- eigh(K) on the device; 'vectors' holds the eigenvectors as ROWS."""
        if K is None:
            K = self.random_effects[1][1]
        evals, evecs_rows = self.ctx.eigh(np.asarray(K, dtype=np.float64))
        return {'values': evals, 'vectors': evecs_rows}

    def _get_eigen_R_(self, X=None, K=None, hat_matrix=None, dtype='double'):
        """:600-615 -- eigh(S (K+I) S), S = I - X (X'X)^+ X'; drop the q null values; -1.

        S B S is formed as B - Q(Q'B) - (BQ)Q' + Q(Q'BQ)Q' with Q an orthonormal basis of X
        (O(N^2 q) instead of two N^3 products); the eigendecomposition runs on the device."""
        if X is None:
            X = self.X
        X = np.asarray(X, dtype=np.float64)
        q = X.shape[1]
        if K is None:
            K = self.random_effects[1][1]
        B = np.asarray(K, dtype=np.float64) + np.asarray(self.random_effects[0][1])
        Q = linalg.orth(X)
        QtB = Q.T @ B
        M = B - Q @ QtB - (B @ Q) @ Q.T + Q @ (QtB @ Q) @ Q.T
        M = 0.5 * (M + M.T)
        evals, evecs_rows = self.ctx.eigh(M)
        return {'values': evals[q:] - 1.0, 'vectors': evecs_rows[q:]}

    # ------------------------------------------------------------------ likelihoods (:618-649)
    def _rell_(self, delta, eig_vals, sq_etas):
        num_eig_vals = len(eig_vals)
        c_1 = 0.5 * num_eig_vals * (np.log(num_eig_vals / (2.0 * np.pi)) - 1)
        v = eig_vals + delta
        return c_1 - 0.5 * (num_eig_vals * np.log(np.sum(sq_etas.flatten() / v)) + np.sum(np.log(v)))

    def _redll_(self, delta, eig_vals, sq_etas):
        num_eig_vals = len(eig_vals)
        v1 = eig_vals + delta
        v2 = sq_etas.flatten() / v1
        return num_eig_vals * np.sum(v2 / v1) / np.sum(v2) - np.sum(1.0 / v1)

    def _ll_(self, delta, eig_vals, eig_vals_L, sq_etas):
        n = self.n
        c_1 = 0.5 * n * (np.log(n / (2.0 * np.pi)) - 1)
        v1 = eig_vals + delta
        v2 = eig_vals_L + delta
        return c_1 - 0.5 * (n * np.log(np.sum(sq_etas.flatten() / v1)) + np.sum(np.log(v2)))

    def _dll_(self, delta, eig_vals, eig_vals_L, sq_etas):
        v1 = eig_vals + delta
        v2 = sq_etas.flatten() / v1
        v3 = eig_vals_L + delta
        return self.n * np.sum(v2 / v1) / np.sum(v2) - np.sum(1.0 / v3)

    def get_REML(self, ngrids=100, llim=-10, ulim=10, esp=1e-6, eig_L=None, eig_R=None):
        """:653-668."""
        if not eig_L:
            eig_L = self._get_eigen_L_(self.random_effects[1][1])
        res = self.get_estimates(eig_L, ngrids=ngrids, llim=llim, ulim=ulim, esp=esp, method='REML', eig_R=eig_R)
        res['eig_L'] = eig_L
        return res

    def get_ML(self, ngrids=100, llim=-10, ulim=10, esp=1e-6, eig_L=None, eig_R=None):
        """:672-683 (the H=None branch)."""
        if not eig_L:
            eig_L = self._get_eigen_L_(self.random_effects[1][1])
        return self.get_estimates(eig_L, ngrids=ngrids, llim=llim, ulim=ulim, esp=esp, method='ML', eig_R=eig_R)

    def get_estimates(self, eig_L, K=None, xs=None, ngrids=50, llim=-10, ulim=10, esp=1e-6,
                      return_pvalue=False, return_f_stat=False, method='REML', verbose=False,
                      dtype='double', eig_R=None, rss_0=None, return_H=True, _rot=None, use_eig_R=False, _sums=None):
        """:771-927 -- EMMA variance-component estimates (Kang et al. 2008).
        use_eig_R: take the likelihood sums from the caller's eig_R even when xs is None (the reference's `:787`
        test would recompute it there); used by callers that already hold eig_R and by the route-comparison tests."""
        if xs is not None:
            xs = np.asarray(xs, dtype=np.float64).reshape(self.n, -1)
            X = np.hstack([self.X, xs])
        else:
            X = self.X
        q = X.shape[1]
        n = self.n
        p = n - q
        m = ngrids + 1
        y = self.Y.reshape(-1)
        log_deltas = (np.arange(m, dtype=np.float64) / ngrids) * (ulim - llim) + llim
        deltas = np.exp(log_deltas)
        eig_vals_L = np.asarray(eig_L['values'], dtype=np.float64) if eig_L is not None else None
        # The likelihood only needs four sums over the spectrum of S(K+delta I)S -- s1 = sum eta^2/(xi+delta),
        # s3 = sum eta^2/(xi+delta)^2, s2 = sum log(xi+delta), s4 = sum 1/(xi+delta) -- and sum eta^2.  The
        # reference gets them from a second N^3 eigendecomposition (eig_R, :787-799); they are also
        #   s1 = y'Py, s3 = |Py|^2, s2 = log|H| + log|X'H^-1 X| - log|X'X|, s4 = tr H^-1 - tr[(X'H^-1 X)^-1 X'H^-2 X]
        # with H = K + delta I and P = H^-1 - H^-1 X (X'H^-1 X)^-1 X'H^-1, i.e. O(N q^2) per delta from eig_L alone
        # (_SpectralSums).  Where the reference would compute eig_R itself (:787) the second eigh is skipped;
        # a caller-supplied eig_R that the reference would use (xs given) is used as is.
        if _sums is not None:
            sums = _sums                                                 # e.g. _SpectralSumsChol: no eigen-pairs at all
            if method != 'REML' and not hasattr(sums, 'at_ml'):
                raise NotImplementedError("these sums evaluate the restricted likelihood only")
        elif eig_R and (xs is not None or use_eig_R):
            sums = _SpectralSumsR(eig_R, y, p)
        elif K is not None or not REML_SUMS_FROM_EIG_L:
            sums = _SpectralSumsR(self._get_eigen_R_(X=X, K=K), y, p)   # :787 (quirk kept)
        else:
            sums = _SpectralSumsL(eig_L, X, y, rot=_rot)
        # the ML likelihood sums log(lambda + delta), 1 / (lambda + delta) over the spectrum of K itself (:634-649): from the
        # eigenvalues, or -- on the eigendecomposition-free route -- as log|K + delta I| and tr (K + delta I)^-1
        if eig_vals_L is not None:
            def h_terms(dd):
                xis = eig_vals_L[:, None] + np.asarray(dd, dtype=np.float64).reshape(1, -1)
                return np.sum(np.log(xis), axis=0), np.sum(1 / xis, axis=0)
        elif method == 'ML':
            def h_terms(dd):
                return sums.at_ml(dd)[2:]
        if method == 'ML' and eig_vals_L is None:
            s1, s3, _ld, _tr = sums.at_ml(deltas)
            s2 = s4 = None
        else:
            s1, s2, s3, s4 = sums.at(deltas)
        if method == 'REML':
            lls = 0.5 * (p * (np.log(p / (2.0 * np.pi)) - 1 - np.log(s1)) - s2)        # :807
            dlls = 0.5 * (p * s3 / s1 - s4)
        elif method == 'ML':
            s2, s4 = h_terms(deltas)
            lls = 0.5 * (n * (np.log(n / (2.0 * np.pi)) - 1 - np.log(s1)) - s2)        # :821
            dlls = 0.5 * (n * s3 / s1 - s4)
        else:
            raise ValueError(method)

        def redll(delta):                                                # _redll_ (:627-631)
            a1, _a2, a3, a4 = sums.at(np.array([delta], dtype=np.float64))
            return float(p * a3[0] / a1[0] - a4[0])

        exact = getattr(sums, 'at_exact', None) or (lambda d: sums.at(np.array([d], dtype=np.float64)))

        def rell(delta):                                                 # _rell_ (:618-623)
            a1, a2, _a3, _a4 = exact(delta)
            return float(0.5 * p * (np.log(p / (2.0 * np.pi)) - 1) - 0.5 * (p * np.log(a1[0]) + a2[0]))

        def ml_point(delta):                                             # (s1, s3, sum log, sum 1/.) at one delta
            dd = np.array([delta], dtype=np.float64)
            if eig_vals_L is None:
                a1, a3, ld, tr = sums.at_ml(dd)
                return a1[0], a3[0], ld[0], tr[0]
            a1, _a2, a3, _a4 = sums.at(dd)
            ld, tr = h_terms(dd)
            return a1[0], a3[0], ld[0], tr[0]

        def dll(delta):                                                  # _dll_ (:643-649)
            a1, a3, _ld, tr = ml_point(delta)
            return float(n * a3 / a1 - tr)

        def ll(delta):                                                   # _ll_ (:634-640)
            a1, _a3, ld, _tr = ml_point(delta)
            return float(0.5 * n * (np.log(n / (2.0 * np.pi)) - 1) - 0.5 * (n * np.log(a1) + ld))

        max_ll_i = int(np.argmax(lls))
        max_ll = lls[max_ll_i]
        zero_intervals = []
        last_dll, last_ll = dlls[0], lls[0]
        for i in range(1, len(dlls)):                                    # :832-836
            if dlls[i] < 0 and last_dll > 0:
                zero_intervals.append(((lls[i] + last_ll) * 0.5, i))
            last_ll, last_dll = lls[i], dlls[i]
        if len(zero_intervals) > 0:
            opt_ll, opt_i = max(zero_intervals)
            opt_delta = 0.5 * (deltas[opt_i - 1] + deltas[opt_i])
            if hasattr(sums, 'prepare_interval') and method == 'REML':
                sums.prepare_interval(deltas[opt_i - 1], deltas[opt_i])   # band route: the search runs on a local model
            try:
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    new_opt_delta = optimize.newton(redll if method == 'REML' else dll, opt_delta, tol=esp,
                                                    maxiter=100)          # :847 (secant)
            except Exception:
                new_opt_delta = opt_delta
            if opt_i > 1 and deltas[opt_i - 1] - esp < new_opt_delta < deltas[opt_i] + esp:
                opt_delta = new_opt_delta
            elif opt_i == 1 and 0.0 < new_opt_delta < deltas[opt_i] + esp:
                opt_delta = new_opt_delta
            elif opt_i == len(deltas) - 1 and new_opt_delta > deltas[opt_i - 1] - esp \
                    and not np.isinf(new_opt_delta):
                opt_delta = new_opt_delta
            opt_ll = rell(opt_delta) if method == 'REML' else ll(opt_delta)   # :882
            if opt_ll < max_ll:
                opt_delta = deltas[max_ll_i]                             # :886-887
        else:
            opt_delta = deltas[max_ll_i]
            opt_ll = max_ll
        # :894-896 -- the reference's (p,1)/(p,) broadcast makes vg = sum(sq_etas) *
        # sum(1/(lambda+delta)) / p ("BUG NEEDS TO BE FIXED HERE!!!" in its own words); the value
        # is reported as is so that results are identical; nothing on the scan path uses it.
        if method == 'ML' and eig_vals_L is None:
            # :894 needs sum 1 / (xi + delta) over the spectrum of S(K + delta I)S = s4 of the REML sums
            s4_opt = sums.at(np.array([opt_delta], dtype=np.float64))[3][0]   # (also fills sums.sum_sq_etas)
            opt_vg = sums.sum_sq_etas * s4_opt / p
        else:
            opt_vg = sums.sum_sq_etas * exact(opt_delta)[3][0] / p
        opt_ve = opt_vg * opt_delta
        if isinstance(sums, _SpectralSumsChol):
            # no H_sqrt_inv without eigenvectors: the GLS estimate and the Mahalanobis RSS (= y'Py = s1) come with
            # the scan model (LinearMixedModel.scan_model_eigen_free)
            return {'max_ll': opt_ll, 'delta': opt_delta, 've': opt_ve, 'vg': opt_vg, 'H_sqrt_inv': None,
                    'pseudo_heritability': 1.0 / (1 + opt_delta), 'n_factorisations': sums.n_factorisations,
                    'n_device_calls': sums.n_calls}
        # :898-907.  H_sqrt_inv = diag((lambda+delta)^-1/2) U'; its products with X and y are row scalings of the
        # rotated U'X, U'y that _SpectralSumsL already holds (O(N q) instead of O(N^2 q)); the N x N matrix itself
        # is only formed when the caller wants it (return_H; the exact-EMMA loop does not).
        wts = 1.0 / np.sqrt(eig_vals_L + opt_delta)
        H_sqrt_inv = wts[:, None] * np.asarray(eig_L['vectors']) if return_H else None
        if isinstance(sums, _SpectralSumsL):
            X_t = wts[:, None] * sums.Xt
            Y_t = wts * sums.yt
        else:
            if H_sqrt_inv is None:
                H_sqrt_inv = wts[:, None] * np.asarray(eig_L['vectors'])
            X_t = H_sqrt_inv @ X
            Y_t = H_sqrt_inv @ y
        (beta_est, _res, rank, sigma) = linalg.lstsq(X_t, Y_t)
        mahalanobis_rss = float(np.sum((Y_t - X_t @ beta_est) ** 2))
        residuals = y - X @ beta_est
        rss = float(residuals @ residuals)
        res_dict = {'max_ll': opt_ll, 'delta': opt_delta, 'beta': beta_est.reshape(-1, 1), 've': opt_ve,
                    'vg': opt_vg, 'rss': rss, 'mahalanobis_rss': np.array([mahalanobis_rss]),
                    'H_sqrt_inv': H_sqrt_inv, 'pseudo_heritability': 1.0 / (1 + opt_delta)}
        if xs is not None and return_f_stat:                             # :914-923
            h0_X = X_t[:, :self.X.shape[1]]                              # H_sqrt_inv @ self.X: the leading columns of X_t
            (h0_betas, _r, h0_rank, h0_s) = linalg.lstsq(h0_X, Y_t)
            h0_rss = float(np.sum((Y_t - h0_X @ h0_betas) ** 2))
            f_stat = (h0_rss / mahalanobis_rss - 1) * p / xs.shape[1]
            res_dict['var_perc'] = 1.0 - mahalanobis_rss / h0_rss
            res_dict['f_stat'] = float(f_stat)
            if return_pvalue:
                if xs.shape[1] == 1:
                    res_dict['p_val'] = float(self.ctx.f_sf([f_stat], p)[0])
                else:
                    from scipy import stats
                    res_dict['p_val'] = float(stats.f.sf(f_stat, xs.shape[1], p))
        return res_dict

    def expedited_REML_t_test(self, snps, ngrids=50, llim=-4, ulim=10, esp=1e-6, verbose=True, eig_L=None):
        """:931-968 -- exact EMMA for a (short) list of SNPs: one N x N eigh per SNP, on the device."""
        assert len(self.random_effects) == 2, "Expedited REMLE only works when we have exactly two random effects."
        if eig_L is None:
            eig_L = self._get_eigen_L_(self.random_effects[1][1])
        keys = ('f_stat', 'vg', 've', 'max_ll', 'var_perc', 'rss', 'p_val')
        out = {k: np.empty(len(snps)) for k in keys}
        betas = []
        # one rotation of y, X and ALL candidate SNPs into the eigenbasis (a single device GEMM) instead of an N x N
        # eigendecomposition -- or even an O(N^2) product -- per SNP; each SNP then costs O(N q^2) per delta
        U = np.asarray(eig_L['vectors'], dtype=np.float64)
        snp_mat = np.ascontiguousarray(np.asarray([np.asarray(sn, dtype=np.float64).reshape(-1) for sn in snps]))
        yt0 = U @ self.Y.reshape(-1)
        Xt0 = U @ self.X
        XSt = self.ctx.dgemm(U, snp_mat, tb=True) if len(snp_mat) else np.zeros((self.n, 0))    # N x k
        for i, snp in enumerate(snps):
            res = self.get_estimates(eig_L=eig_L, xs=np.asarray(snp, dtype=np.float64).reshape(-1, 1),
                                     ngrids=ngrids, llim=llim, ulim=ulim, esp=esp, return_pvalue=True,
                                     return_f_stat=True, return_H=False,
                                     _rot=(yt0, np.hstack([Xt0, XSt[:, i:i + 1]])))
            for k in keys:
                out[k][i] = res[k]
            betas.append([float(b) for b in res['beta'].reshape(-1)])
        return {'ps': out['p_val'], 'f_stats': out['f_stat'], 'vgs': out['vg'], 'ves': out['ve'],
                'var_perc': out['var_perc'], 'max_lls': out['max_ll'], 'betas': betas, 'rss': out['rss']}

    # ------------------------------------------------------------------ EMMAX
    def emmax_f_test(self, snps, snp_priors=None, Z=None, with_betas=False, method='REML',
                     eig_L=None, eig_R=None, emma_num=100, verbose=False):
        """:1233-1267."""
        t = {}
        s0 = time.time()
        if (self.n > EIGEN_FREE_MIN_N and not eig_L and not eig_R and Z is None and emma_num == 0
                and method == 'REML' and isinstance(self.ctx, _lib.Context) and len(self.random_effects) == 2):
            # beyond rocSOLVER's syevd index range: REML and the scan model from Cholesky factorisations of K + delta I
            # (get_estimates_eigen_free) instead of the block-Jacobi eigendecomposition (6.6 min at N = 50,000)
            res = self._try_eigen_free()
        else:
            res = None
        if res is not None:
            t['eig_L'] = t['eig_R'] = 0.0
            t['reml'] = time.time() - s0
            s0 = time.time()
            reml = res.pop('reml')
            try:
                r = self._emmax_f_test_(snps, None, snp_priors=snp_priors, emma_num=0, verbose=verbose,
                                        with_betas=with_betas, _delta=res['delta'], _reml=reml)
            finally:
                reml.close()
            t['scan'] = time.time() - s0
            r.update(pseudo_heritability=res['pseudo_heritability'], ve=res['ve'], vg=res['vg'], max_ll=res['max_ll'],
                     timings=t)
            return r
        s0 = time.time()
        if not eig_L:
            eig_L = self._get_eigen_L_()
        t['eig_L'] = time.time() - s0
        t['eig_R'] = 0.0
        s0 = time.time()
        # The reference computes eig_R here (:1252) and again inside get_estimates (:787).  The likelihood sums
        # come from eig_L alone (_SpectralSumsL), so neither N^3 eigendecomposition is needed; a caller-supplied
        # eig_R is still honoured.
        # The scan model A = Mp Mp' = P(delta), w = P y can be built from K and delta on the device (one Cholesky
        # factorisation, mmg_reml_scan_model) instead of forming H_sqrt_inv and the N^3 product T'T on the host and
        # moving three N x N matrices over PCIe: 0.6 s -> 0.05 s at N = 5000.  Taken when nothing needs H itself.
        device_model = (DEVICE_SCAN_MODEL and Z is None and not with_betas and isinstance(self.ctx, _lib.Context)
                        and len(self.random_effects) == 2)
        if eig_R:
            res = self._get_estimates_with(eig_L, eig_R, method)
        else:
            res = self.get_estimates(eig_L, method=method, return_H=not device_model)
        t['reml'] = time.time() - s0
        s0 = time.time()
        try:
            r = self._emmax_f_test_(snps, res['H_sqrt_inv'], snp_priors=snp_priors, Z=Z, with_betas=with_betas,
                                    emma_num=emma_num, eig_L=eig_L, verbose=verbose,
                                    _delta=res['delta'] if device_model else None)
        except _lib.MixmogamHipError as e:
            if not (device_model and "positive definite" in str(e)):
                raise
            # an indefinite kinship: K + delta I has no Cholesky factor; build the model from H_sqrt_inv instead
            res = self.get_estimates(eig_L, method=method)
            r = self._emmax_f_test_(snps, res['H_sqrt_inv'], snp_priors=snp_priors, Z=Z, with_betas=with_betas,
                                    emma_num=emma_num, eig_L=eig_L, verbose=verbose)
        t['scan'] = time.time() - s0
        r['pseudo_heritability'] = res['pseudo_heritability']
        r['ve'] = res['ve']
        r['vg'] = res['vg']
        r['max_ll'] = res['max_ll']
        r['timings'] = t
        if verbose:
            print('EMMAX timings (s):', t)
        return r

    def _get_estimates_with(self, eig_L, eig_R, method, ngrids=50):
        """get_estimates on a PRECOMPUTED eig_R (the reference's own route, :787-799)."""
        return self.get_estimates(eig_L, method=method, eig_R=eig_R, ngrids=ngrids, use_eig_R=True)

    def _try_eigen_free_method(self, method='REML', ngrids=50):
        """get_estimates_eigen_free(method=...) or None (K + delta I not positive definite on the grid: eigen route)."""
        try:
            return self.get_estimates_eigen_free(method=method, ngrids=ngrids)
        except _lib.MixmogamHipError as e:
            if "positive definite" not in str(e):
                raise
            warnings.warn("K + delta*I is not positive definite on the likelihood grid; taking the eigendecomposition route")
            return None

    def _try_eigen_free(self, coll=None):
        """get_estimates_eigen_free, or None when K + delta I is not positive definite somewhere on the grid (an
        indefinite user-supplied kinship: the eigen route copes with that, Cholesky cannot)."""
        try:
            return self.get_estimates_eigen_free(coll=coll)
        except _lib.MixmogamHipError as e:
            if "positive definite" not in str(e):
                raise
            warnings.warn("K + delta*I is not positive definite on the REML grid; taking the eigendecomposition route")
            return None

    def get_estimates_eigen_free(self, ngrids=50, llim=-10, ulim=10, esp=1e-6, coll=None, method='REML'):
        """get_estimates(method='REML') (:771-927) without eig_L / eig_R: the likelihood sums come from Cholesky
        factorisations on the device (_SpectralSumsChol).  Returns the same scalars (max_ll, delta, ve, vg,
        pseudo_heritability) plus 'reml': the device workspace to hand to scan_model_eigen_free.  No H_sqrt_inv:
        callers that need the matrix itself (permutation test, exact EMMA) take the eigen route."""
        K = self.random_effects[1][1]
        reml = self.ctx.reml(K, self.X, self.Y.reshape(-1))
        res = self.get_estimates(None, ngrids=ngrids, llim=llim, ulim=ulim, esp=esp, method=method,
                                 _sums=_SpectralSumsChol(reml, coll))
        res['reml'] = reml
        return res

    def scan_model_eigen_free(self, res, ndigits=0):
        """Load the EMMAX scan model of res['delta'] (A = P, w = Py, built on the device; :1290-1303 in closed form)
        into the context and return the SNP-independent outputs of scan_prepare: h0_rss, h0_betas, n_p."""
        h0_rss, beta = res['reml'].scan_model(res['delta'], ndigits)
        y = self.Y.reshape(-1)
        resid = y - self.X @ beta
        res.update(beta=beta.reshape(-1, 1), mahalanobis_rss=np.array([h0_rss]), rss=float(resid @ resid))
        return {'h0_rss': h0_rss, 'h0_betas': [float(b) for b in beta], 'n_p': self.n - (self.X.shape[1] + 1)}

    def scan_prepare(self, H_sqrt_inv, Z=None, with_betas=False):
        """SNP-independent part of _emmax_f_test_ (:1290-1306) in closed form:
        A = Mp Mp' (device dgemm), w = Mp r, plus the q rows of C = R^-1 Q' H used by with_betas."""
        H = np.asarray(H_sqrt_inv, dtype=np.float64)
        y = self.Y.reshape(-1)
        h0_X = H @ self.X                                                # :1290
        Yt = H @ y                                                       # :1291
        (h0_betas, _r, h0_rank, h0_s) = linalg.lstsq(h0_X, Yt)           # :1292
        r = Yt - h0_X @ h0_betas                                         # :1293
        h0_rss = float(r @ r)
        if Z is not None:
            H = H @ np.asarray(Z, dtype=np.float64)                      # :1296-1297
        (Q, R) = linalg.qr(h0_X, mode='economic')                        # :1300
        T = H - Q @ (Q.T @ H)                                            # (I - QQ') H   [n x n_geno]
        A = self.ctx.dgemm(T, T, ta=True)                                # Mp Mp' = T'T
        A = 0.5 * (A + A.T)
        w = T.T @ r
        prep = {'h0_rss': h0_rss, 'h0_betas': [float(b) for b in h0_betas], 'r': r, 'A': A, 'w': w,
                'n_p': self.n - (self.X.shape[1] + 1),
                'HtQ': np.ascontiguousarray((H.T @ Q).T)}              # [q x n_geno]: A = H'H - sum_c u_c u_c'

        if with_betas:
            prep['C'] = linalg.solve_triangular(R, Q.T @ H)              # q x n_geno: (X0'X0)^-1 X0' H
        return prep

    def _emmax_f_test_(self, snps, H_sqrt_inv, snp_priors=None, verbose=True, return_transformed_snps=False,
                       Z=None, with_betas=False, emma_num=100, eig_L=None, ndigits=0, _delta=None, _reml=None,
                       **kwargs):
        """:1272-1380.  `snps`: list of M arrays / [M x N] array, or a device-resident _lib.Geno.
        _delta (internal): build the scan model from K and this variance ratio on the device instead of from
        H_sqrt_inv on the host (same matrix: Mp Mp' = P(delta))."""
        ctx = self.ctx
        if return_transformed_snps and H_sqrt_inv is None:
            raise NotImplementedError("return_transformed_snps needs H_sqrt_inv")
        prep = None
        if _delta is not None and Z is None:
            reml = _reml if _reml is not None else ctx.reml(self.random_effects[1][1], self.X, self.Y.reshape(-1))
            try:
                if with_betas:                                           # + C = (X'V^-1 X)^-1 X'V^-1 = R^-1 Q'H of :1300-1303
                    h0_rss_d, beta_d, c_d = reml.scan_model(_delta, ndigits, want_C=True)
                else:
                    (h0_rss_d, beta_d), c_d = reml.scan_model(_delta, ndigits), None
                prep = {'h0_rss': h0_rss_d, 'h0_betas': [float(b) for b in beta_d],
                        'n_p': self.n - (self.X.shape[1] + 1)}
                if c_d is not None:
                    prep['C'] = c_d
            except _lib.MixmogamHipError as e:
                # an indefinite kinship has no Cholesky factor of K + delta I: callers that hold H_sqrt_inv (mlmm,
                # the chunked drivers) get the model built from it instead; without it the error stands
                if H_sqrt_inv is None or "positive definite" not in str(e):
                    raise
            finally:
                if _reml is None:
                    reml.close()
        if prep is None:
            if callable(H_sqrt_inv):                                     # a caller that holds delta and eig_L, not the matrix (mlmm)
                H_sqrt_inv = H_sqrt_inv()
            prep = self.scan_prepare(H_sqrt_inv, Z=Z, with_betas=with_betas)
        own = not isinstance(snps, _lib.Geno)
        g = ctx.geno(kinship._as_snp_matrix(snps)) if own else snps
        try:
            num_snps = g.M
            if 'A' in prep:
                ctx.scan_set_model(prep['A'], prep['w'], ndigits)
            n_p = prep['n_p']
            h0_rss = prep['h0_rss']
            out = ctx.scan(g, h0_rss, n_p, stats=with_betas)
            rss_list, f_stats, p_vals = out['rss'], out['f_stats'], out['ps']
            res_d = {'ps': p_vals, 'f_stats': f_stats, 'rss': rss_list, 'var_perc': 1 - rss_list / h0_rss,
                     'h0_rss': np.array([h0_rss]), 'h0_betas': prep['h0_betas']}
            if with_betas:
                # lstsq([h0_X, H s], r) (:1323): beta_snp = (s.w)/(s'As); the covariate part is
                # -C s * beta_snp (r is orthogonal to h0_X).  Rank-deficient SNPs keep h0_betas (:1305).
                ok = rss_list != h0_rss
                b_snp = np.where(ok, out['dot'] / np.where(ok, out['den'], 1.0), 0.0)
                Cs = g.matvec(prep['C'])                                 # q x M
                q = Cs.shape[0]
                B = np.empty((num_snps, q + 1))
                B[:, :q] = -(Cs * b_snp).T
                B[:, q] = b_snp
                # 'betas': per SNP the q + 1 coefficients as a ROW OF ONE ARRAY (q + 1 numpy floats; the reference builds a list
                # of Python floats per SNP -- 1.5 M of them cost 0.2 s at M = 500,000); a rank-deficient SNP keeps the null
                # model's q values as a plain list, as in the reference (:1305)
                betas = list(B)
                for j in np.nonzero(~ok)[0]:
                    betas[j] = list(prep['h0_betas'])
                res_d['betas'] = betas
            if return_transformed_snps:                                  # :1309-1321,:1355-1356
                res_d['t_snps'] = self._transformed_snps(g, H_sqrt_inv, Z, project=not with_betas)
            if snp_priors is not None:                                   # :1311-1314,:1357-1363
                snp_priors = np.asarray(snp_priors, dtype=np.float64)
                n = self.n
                log_bfs = np.where(rss_list != h0_rss, np.log(h0_rss) - np.log(rss_list), 0.0)
                bfs = np.exp((log_bfs * n - np.log(n)) * 1 / 2)
                pos = bfs * snp_priors / (1 - snp_priors)
                res_d.update(bfs=bfs, pos=pos, ppas=pos / (1 + pos))
            if emma_num > 0 and num_snps > 0:                            # :1365-1377
                # the emma_num smallest p-values in the order of a stable argsort, without sorting all M of them
                kth = np.partition(p_vals, emma_num - 1)[emma_num - 1] if emma_num < num_snps else np.nan
                if kth == kth:                                           # (a NaN among the smallest: sort them all)
                    cand = np.nonzero(p_vals <= kth)[0]
                    order = cand[np.argsort(p_vals[cand], kind='stable')][:emma_num]
                else:
                    order = np.argsort(p_vals, kind='stable')[:emma_num]
                top = g.download_rows(order) if not own else kinship._as_snp_matrix(snps)[order]
                top_res = self.expedited_REML_t_test(list(top), eig_L=eig_L)
                for k, pi in enumerate(order):
                    res_d['ps'][pi] = top_res['ps'][k]
                    res_d['f_stats'][pi] = top_res['f_stats'][k]
                    res_d['rss'][pi] = top_res['rss'][k]
                    res_d['var_perc'][pi] = top_res['var_perc'][k]
        finally:
            if own:
                g.close()
        return res_d

    def _transformed_snps(self, g, H_sqrt_inv, Z=None, project=True):
        """t_m = s_m Mp, Mp = H'(I - QQ') (:1300-1303,1318-1321): what the reference's loop regresses the residual
        on, returned as a list of M arrays like the reference's `t_snps`.  T = S Mp is the rotation GEMM of the
        multi-phenotype path with the rows of Mp' = (I - QQ')H in place of the eigenvectors (mmg_rot_load: exact int8
        digit GEMM: four unsigned 7-bit digits per entry, 2^-27 of each row's largest entry).  With replicates (Z: n values x n_geno individuals) Mp' is
        not square and the product is a device dgemm instead."""
        H = np.asarray(H_sqrt_inv, dtype=np.float64)
        h0_X = H @ self.X
        if Z is not None:
            H = H @ np.asarray(Z, dtype=np.float64)
        (Q, _R) = linalg.qr(h0_X, mode='economic')
        # with_betas: the reference regresses on [h0_X, H s] and its M is H' itself (:1305); otherwise Mp' = (I - QQ') H
        MpT = H - Q @ (Q.T @ H) if project else H                        # [n x n_geno]
        if MpT.shape[0] != MpT.shape[1] or g.M == 0:
            S = g.download().astype(np.float64)
            T = self.ctx.dgemm(MpT, S, tb=True) if g.M else np.zeros((MpT.shape[0], 0))
        else:
            rot = self.ctx.rot(np.ascontiguousarray(MpT), g.M)
            try:
                T = rot.load(g).fetch()                                  # [n x M]: T[i][m] = Mp'[i] . s_m
            finally:
                rot.close()
        return list(np.ascontiguousarray(T.T))

    # ------------------------------------------------------------------ permutations
    def perm_prepare(self, H_sqrt_inv, num_perm=100, perm_idx=None, reml=None, delta=None):
        """SNP-independent part of _emmax_permutations_ (:1135-1156): centred Y (mutated, as the reference does),
        null fit, the N x P matrix of permuted residuals.  perm_idx: optional [num_perm x n] index matrix (column p
        of Ys is r[perm_idx[p]]); when None the permutations are drawn exactly as the reference draws them --
        successive in-place numpy.random.shuffle calls on the global RNG (:1151-1154).
        reml / delta (H_sqrt_inv None): H = L^-1 of K + delta I = L L' from the device workspace (_lib.Reml.linv_apply) --
        H X and H y without an eigendecomposition and without the N x N matrix on the host."""
        n = self.n
        self.Y = self.Y - np.mean(self.Y)                                # :1140 (mutates, as the reference)
        y = self.Y.reshape(-1)
        if H_sqrt_inv is None:
            Zt = reml.linv_apply(delta, np.column_stack([self.X, y]))
            H, h0_X, Yt = None, Zt[:, :-1], Zt[:, -1]
        else:
            H = np.asarray(H_sqrt_inv, dtype=np.float64)
            h0_X = H @ self.X
            Yt = H @ y
        (h0_betas, _r, h0_rank, h0_s) = linalg.lstsq(h0_X, Yt)           # :1143
        r = Yt - h0_X @ h0_betas                                         # :1144
        h0_rss = float(r @ r)
        r = r - h0_X @ h0_betas                                          # :1147 (second subtraction, kept)
        if perm_idx is None:
            # the reference shuffles an n x 1 matrix in place; a 1-D array draws the same Fisher-Yates sequence from the
            # same generator state (checked: identical permutations) without numpy.matrix's per-element row swaps --
            # 0.6 of 0.75 s of a 100-permutation test at N = 1000
            idx = np.arange(n)
            perm_idx = []
            for _ in range(num_perm):
                np.random.shuffle(idx)
                perm_idx.append(idx.copy())
        perm_idx = np.asarray(perm_idx)
        return {'H': H, 'Ys': np.ascontiguousarray(r[perm_idx].T), 'h0_rss': h0_rss,   # n x P: column p = r[perm_idx[p]]
                'n_p': n - (self.X.shape[1] + 1), 'h0_X': h0_X}

    def _emmax_permutations_(self, snps, K, H_sqrt_inv, num_perm=100, perm_idx=None, ndigits=0):
        """:1125-1175 (perm_idx: see perm_prepare)."""
        ctx = self.ctx
        pp = self.perm_prepare(H_sqrt_inv, num_perm=num_perm, perm_idx=perm_idx)
        own = not isinstance(snps, _lib.Geno)
        g = ctx.geno(kinship._as_snp_matrix(snps)) if own else snps
        try:
            min_rss = ctx.perm(g, pp['H'], pp['Ys'], pp['h0_rss'], ndigits)
        finally:
            if own:
                g.close()
        max_f_stats = ((pp['h0_rss'] / min_rss) - 1.0) * pp['n_p']        # :1171
        min_pvals = ctx.f_sf(max_f_stats, pp['n_p'])                     # :1172
        return {'min_ps': min_pvals, 'max_f_stats': max_f_stats}

    def emmax_permutations(self, snps, num_perm, method='REML', perm_idx=None, H_sqrt_inv=None,
                           reference_indexing=False):
        """:1180-1230 -- the PUBLIC permutation test (emmax_perm_test's worker).  Its arithmetic is not that of
        _emmax_permutations_ (:1125): Y is not centred, the null fit is subtracted once (:1200), and the SNP is
        centred AFTER the transform, Xs - mean(Xs) (:1211), i.e. t_m = C H s_m with C = I - 11'/n -- the device
        test run on Ht = C H without SNP centring (mmg_perm_plan_create_ex flag 1).

        The reference then stores `rss_list.min()` -- the minimum over PERMUTATIONS of SNP i+j -- at index i+j of a
        per-permutation array (:1213): an IndexError once num_snps > num_perm, and otherwise not what its docstring
        ("the list of max_pvals and max_fstats" per permutation) describes.  Default here: what the docstring says,
        min over SNPs per permutation.  reference_indexing=True reproduces the reference's literal output (first
        num_snps slots = per-SNP minima over the permutations, the rest h0_rss; IndexError beyond num_perm SNPs) so
        that the wrapper can be checked against the reference's own numbers.

        perm_idx: optional [num_perm x n] index matrix (column p of Ys = r[perm_idx[p]]); None draws successive
        in-place numpy.random.shuffle calls as the reference does (:1202-1205).  H_sqrt_inv: optional, the matrix the
        estimates would give (its row signs are LAPACK's choice and the shuffled vector lives in that basis)."""
        ctx = self.ctx
        n = self.n
        n_p = n - (self.X.shape[1] + 1)                                  # :1190-1193
        y = self.Y.reshape(-1)
        reml = None
        if (H_sqrt_inv is None and perm_h_from_cholesky(ctx) and n > EIGEN_FREE_MIN_N and isinstance(ctx, _lib.Context)):
            # (the gate of emmax_f_test / get_emma_reml_estimates: a real device context and N past the eigen-free threshold)
            # H := L^-1 of K + delta I = L L' (any H with H'H = (K + delta I)^-1 is a valid H_sqrt_inv; the reference's own
            # is fixed only up to LAPACK's eigenvector signs, and the shuffled vector lives in the basis of the H that is
            # used): REML on the device without eigh(K), H X / H y by mmg_reml_linv_apply, the plan from the workspace in
            # HBM.  MMG_PERM_H=eigen: the literal route below.
            est = self._try_eigen_free_method(method)
            if est is not None:
                reml, delta = est['reml'], est['delta']
                try:
                    Zt = reml.linv_apply(delta, np.column_stack([self.X, y]))
                except Exception:
                    reml.close()                                         # the workspace does not outlive a failed product
                    raise
                h0_X, Yt = Zt[:, :-1], Zt[:, -1]
        if reml is None:
            if H_sqrt_inv is None:
                K = self.random_effects[1][1]
                eig_L = self._get_eigen_L_(K)
                H_sqrt_inv = self.get_estimates(eig_L=eig_L, method=method)['H_sqrt_inv']   # :1184-1186
            H = np.asarray(H_sqrt_inv, dtype=np.float64)
            Yt = H @ y                                                   # :1195
            h0_X = H @ self.X                                            # :1196
        (h0_betas, _r, _rank, _s) = linalg.lstsq(h0_X, Yt)               # :1197
        r = Yt - h0_X @ h0_betas                                         # :1198
        h0_rss = float(r @ r)
        if perm_idx is None:                                             # :1202-1205
            idx = np.arange(n)                                           # see perm_prepare: the same draws as the n x 1 matrix
            perm_idx = []
            for _ in range(num_perm):
                np.random.shuffle(idx)
                perm_idx.append(idx.copy())
        perm_idx = np.asarray(perm_idx)
        Ys = np.ascontiguousarray(r[perm_idx].T)                         # n x P
        own = not isinstance(snps, _lib.Geno)
        snp_mat = kinship._as_snp_matrix(snps) if own else None
        num_snps = len(snp_mat) if own else snps.M
        if reml is not None:
            try:                                                         # C H: the transformed SNP minus its mean (:1211), on the device
                plan = reml.perm_plan(delta, Ys, h0_rss, centre_snps=False, centre_H=True)
            finally:
                reml.close()
        else:
            CH = H - H.mean(axis=0, keepdims=True)                       # C H: the transformed SNP minus its mean (:1211)
            plan = ctx.perm_plan(CH, Ys, h0_rss, centre_snps=False)
        try:
            if not reference_indexing:
                g = ctx.geno(snp_mat) if own else snps
                try:
                    min_rss = plan.run(g)
                finally:
                    if own:
                        g.close()
            else:
                if num_snps > num_perm:
                    raise IndexError("index %d is out of bounds for axis 0 with size %d (linear_models.py:1213 stores "
                                     "a per-SNP minimum in a per-permutation array)" % (num_perm, num_perm))
                min_rss = np.repeat(h0_rss, num_perm).astype(np.float64)  # :1207
                rows = snp_mat if own else snps.download()
                for j in range(num_snps):                                # one SNP per run: min over the permutations
                    g1 = ctx.geno(rows[j:j + 1])
                    try:
                        min_rss[j] = plan.run(g1).min()
                    finally:
                        g1.close()
        finally:
            plan.close()
        max_f_stats = ((h0_rss / min_rss) - 1.0) * n_p                   # :1221
        min_pvals = ctx.f_sf(max_f_stats, n_p)                           # :1222
        return {'min_ps': min_pvals, 'max_f_stats': max_f_stats}


# ---------------------------------------------------------------------- module-level entry points
def _cofactor_list(cofactors):
    """The reference's callers pass a list of length-N vectors and its code tests it with `if cofactors:` (:1797-1803), which
    raises on an ndarray.  Here None / empty -> no cofactors, a 2-D array -> its rows, a 1-D array -> one cofactor."""
    if cofactors is None:
        return []
    if isinstance(cofactors, np.ndarray):
        return [cofactors] if cofactors.ndim == 1 else list(cofactors)
    return list(cofactors)


class _LazyEstimates(dict):
    """The result of get_emma_reml_estimates on the eigendecomposition-free route: the scalars are there; 'H_sqrt_inv', 'Y_t',
    'X_t' (from the Cholesky square root L^-1 of (K + delta I)^-1 -- any H with H'H = (K + delta I)^-1 serves, linear_models.py
    :898 is fixed only up to LAPACK's signs) and 'eig_L' (rocSOLVER's dsyevd, 0.27 s at N = 5000) are computed when first
    asked for.  Iterating / items() / values() materialise everything."""
    _LAZY = ('H_sqrt_inv', 'Y_t', 'X_t', 'eig_L')

    def __init__(self, scalars, lmm):
        dict.__init__(self, scalars)
        self._lmm = lmm

    def _workspace(self):
        """A fresh device workspace of this model (2 N^2 doubles + band buffers in HBM): the result does not hold one -- a caller
        that keeps the results of a loop over phenotypes would pin 0.4 GB each at N = 5000, 40 GB at N = 50,000 (advisor r5)."""
        lmm = self._lmm
        return lmm.ctx.reml(lmm.random_effects[1][1], lmm.X, lmm.Y.reshape(-1))

    def _make(self, key):
        if key == 'eig_L':
            v = self._lmm._get_eigen_L_()
        else:
            reml = self._workspace()
            try:
                if key == 'H_sqrt_inv':
                    v = reml.linv(dict.__getitem__(self, 'delta'))
                else:
                    Zt = reml.linv_apply(dict.__getitem__(self, 'delta'),
                                         np.column_stack([self._lmm.X, self._lmm.Y.reshape(-1)]))
                    dict.__setitem__(self, 'X_t', Zt[:, :-1])
                    dict.__setitem__(self, 'Y_t', Zt[:, -1:])
                    return dict.__getitem__(self, key)
            finally:
                reml.close()
        dict.__setitem__(self, key, v)
        return v

    def __getitem__(self, key):
        if key in self._LAZY and not dict.__contains__(self, key):
            return self._make(key)
        return dict.__getitem__(self, key)

    def get(self, key, default=None):
        return self[key] if key in self else default

    def __contains__(self, key):
        return key in self._LAZY or dict.__contains__(self, key)

    def _all(self):
        for k in self._LAZY:
            self[k]
        return self

    def keys(self):
        return list(dict.keys(self)) + [k for k in self._LAZY if not dict.__contains__(self, k)]

    def __iter__(self):
        return iter(self.keys())

    def __len__(self):
        return len(self.keys())

    def items(self):
        return dict.items(self._all())

    def values(self):
        return dict.values(self._all())

    def close(self):
        """Kept for callers of round 5: the result no longer owns a device workspace, there is nothing to release."""


def get_emma_reml_estimates(y, K, K2=None, cofactors=None, include_intercept=True, ctx=None):
    """:1690-1706.  Round 5: above EIGEN_FREE_MIN_N individuals the variance components come from the band reduction of K
    (no eigendecomposition); beta / rss / mahalanobis_rss from two triangular products with the Cholesky factor; the matrices
    the reference also returns ('H_sqrt_inv', 'Y_t', 'X_t', 'eig_L') are computed on first access (_LazyEstimates).
    MMG_PERM_H=eigen: the eigendecomposition route with every entry filled."""
    if K2 is not None:
        raise NotImplementedError("two-kinship estimator (get_estimates_3) is outside the hot path (SURVEY 2)")
    lmm = LinearMixedModel(y, ctx=ctx)
    lmm.add_random_effect(K)
    if cofactors is not None:
        lmm.set_factors(cofactors, include_intercept=include_intercept)
    if lmm.n > EIGEN_FREE_MIN_N and isinstance(lmm.ctx, _lib.Context) and perm_h_from_cholesky(lmm.ctx):
        est = lmm._try_eigen_free_method('REML', ngrids=100)              # get_REML's grid (:653), as the eigen route below
        if est is not None:
            reml = est.pop('reml')
            try:
                Zt = reml.linv_apply(est['delta'], np.column_stack([lmm.X, lmm.Y.reshape(-1)]))
            finally:
                reml.close()                                              # the result keeps no HBM (see _LazyEstimates)
            X_t, Y_t = Zt[:, :-1], Zt[:, -1]
            (beta_est, _res, _rank, _sigma) = linalg.lstsq(X_t, Y_t)        # :902-907
            resid = lmm.Y.reshape(-1) - lmm.X @ beta_est
            est.pop('H_sqrt_inv', None)
            est.update(beta=beta_est.reshape(-1, 1), mahalanobis_rss=np.array([float(np.sum((Y_t - X_t @ beta_est) ** 2))]),
                       rss=float(resid @ resid), lmm=lmm)
            out = _LazyEstimates(est, lmm)
            dict.__setitem__(out, 'X_t', X_t)
            dict.__setitem__(out, 'Y_t', Y_t.reshape(-1, 1))
            return out
    res = lmm.get_REML()
    res['Y_t'] = res['H_sqrt_inv'] @ lmm.Y
    res['X_t'] = res['H_sqrt_inv'] @ lmm.X
    res['lmm'] = lmm
    return res


def emmax(snps, phenotypes, K, cofactors=None, Z=None, with_betas=False, emma_num=0, ctx=None, verbose=False):
    """:1790-1816 -- run EMMAX."""
    lmm = LinearMixedModel(phenotypes, ctx=ctx)
    if Z is not None:
        Z = np.asarray(Z, dtype=np.float64)
        lmm.add_random_effect(Z @ np.asarray(K) @ Z.T)                  # :1796
        for cofactor in _cofactor_list(cofactors):
            lmm.add_factor(Z @ _col(cofactor))
    else:
        lmm.add_random_effect(K)
        for cofactor in _cofactor_list(cofactors):
            lmm.add_factor(cofactor)
    s1 = time.time()
    res = lmm.emmax_f_test(snps, Z=Z, with_betas=with_betas, emma_num=emma_num, verbose=verbose)
    if verbose:
        print('Took %f seconds.' % (time.time() - s1))
    return res


# ---------------------------------------------------------------------- multi-phenotype scans
def _multi_models(ys, X, eig_L, method='REML'):
    """Per-phenotype, SNP-independent part of a multi-phenotype scan: REML (:771-927) and the null fit of
    _emmax_f_test_ (:1290-1303) written in the eigenbasis of K.  O(N^2 P) once for the rotation of Y, then
    O(N q^2) per phenotype and grid point.  Returns (models, d, omega, G) with d, omega [P x N], G [P x q x N]
    as mmg_emmax_scan_multi takes them."""
    V = np.asarray(eig_L['vectors'], dtype=np.float64)
    lam = np.asarray(eig_L['values'], dtype=np.float64)
    ys = np.asarray(ys, dtype=np.float64)
    P, n = ys.shape
    q = X.shape[1]
    Yt = V @ ys.T                                                        # N x P: every phenotype rotated at once
    Xt = V @ X
    models, d, omega, G = [], np.empty((P, n)), np.empty((P, n)), np.empty((P, q, n))
    for p in range(P):
        lmm = LinearMixedModel(ys[p], ctx=False)
        lmm.X, lmm.p = X, q
        est = lmm.get_estimates(eig_L, method=method, return_H=False, _rot=(Yt[:, p], Xt))
        w = 1.0 / np.sqrt(lam + est['delta'])                            # :898 diag of H_sqrt_inv in the eigenbasis
        h0_X = w[:, None] * Xt                                           # :1290
        Y_t = w * Yt[:, p]                                               # :1291
        (h0_betas, _r, _rank, _s) = linalg.lstsq(h0_X, Y_t)              # :1292
        r = Y_t - h0_X @ h0_betas                                        # :1293
        (Q, _R) = linalg.qr(h0_X, mode='economic')                       # :1300
        d[p] = w * w
        omega[p] = r * w
        G[p] = (Q * w[:, None]).T
        est.update(h0_rss=float(r @ r), h0_betas=[float(b) for b in h0_betas])
        models.append(est)
    return models, d, omega, G


def emmax_multi(snps, phenotypes, K, cofactors=None, ctx=None, coll=None, max_store_bytes=64 << 30, method='REML'):
    """EMMAX for P phenotypes measured on the same individuals: the result of a loop of `emmax(snps, y_p, K,
    cofactors)` calls (one LinearMixedModel, REML and scan per phenotype -- what the reference does,
    phenotypeData.py:70-78 / hdf5_data.py:262-330 once per phenotype), computed with ONE eigendecomposition and
    ONE O(N^2) pass over the genotypes: the SNPs are rotated into the eigenbasis of K on the int8 matrix cores
    (T = S U', kept in HBM) and each phenotype's scan -- its own delta_p, H_p, null model -- is then an HBM-bound
    pass over T (mmg_emmax_scan_multi, 8 phenotypes per pass).

    phenotypes: [P x N] (list of P lists).  snps: list / [M x N] array, or a device-resident _lib.Geno.
    coll (mixmogam_amd.dist): SNP blocks are sharded over the ranks and the [P x M] results all-gathered.
    Returns {'ps','f_stats','rss','var_perc'} as [P x M] arrays and per-phenotype lists 'h0_rss', 'h0_betas',
    'pseudo_heritability', 've', 'vg', 'max_ll', 'delta' (the keys of emmax(), :1351-1354,:1262-1265)."""
    ys = np.asarray(phenotypes, dtype=np.float64)
    if ys.ndim != 2:
        raise ValueError("phenotypes must be [num_phenotypes x num_individuals]")
    P, n = ys.shape
    lmm0 = LinearMixedModel(ys[0], ctx=ctx)
    lmm0.add_random_effect(K)                                            # scale_k once (:580): same K for every phenotype
    for cofactor in _cofactor_list(cofactors):
        lmm0.add_factor(cofactor)                                        # dependence on X only: same for every phenotype
    ctx = lmm0.ctx
    X = lmm0.X
    q = X.shape[1]
    if q > 8:
        raise NotImplementedError("emmax_multi: at most 7 cofactors besides the intercept on the rotated path")
    eig_L = lmm0._get_eigen_L_()
    own = not isinstance(snps, _lib.Geno)
    if own:
        snps = kinship._as_snp_matrix(snps)
    M = snps.M if not own else len(snps)
    lam = np.asarray(eig_L['values'], dtype=np.float64)
    if coll is None and method == 'REML' and isinstance(ctx, _lib.Context) and \
            int(np.sum(lam > 1e-9 * max(float(lam.max()), 1e-300))) < n // 2:
        # A kinship of numerical rank below N / 2 (a handful of SNPs or of genotype classes): SNPs that lie in its span have
        # quadratic forms far below what the 27-bit rows of the rotation carry (p off by 1.4e-6 at N = 263 on a kinship of two
        # 0/1/2 SNPs, tools/random_parity.py).  The single-phenotype scan has an fp64 tier for exactly these SNPs
        # (mmg_scan_last_exact), so such a kinship takes the loop the rotated path replaces -- with ONE eigendecomposition
        # and one resident genotype store for all phenotypes.
        g = ctx.geno(snps) if own else snps
        try:
            per = []
            for p_ in range(P):
                lmm_p = LinearMixedModel(ys[p_], ctx=ctx)
                lmm_p.add_random_effect(K)
                for cofactor in _cofactor_list(cofactors):
                    lmm_p.add_factor(cofactor)
                per.append(lmm_p.emmax_f_test(g, eig_L=eig_L, emma_num=0))
        finally:
            if own:
                g.close()
        res = {k: np.asarray([r_[k] for r_ in per]) for k in ('ps', 'f_stats', 'rss', 'var_perc')}
        res['h0_rss'] = np.asarray([float(np.asarray(r_['h0_rss']).reshape(-1)[0]) for r_ in per])
        res['h0_betas'] = [r_['h0_betas'] for r_ in per]
        for k in ('pseudo_heritability', 've', 'vg', 'max_ll'):
            res[k] = np.array([r_[k] for r_ in per])
        res['delta'] = np.array([r_['delta'] if 'delta' in r_ else 1.0 / r_['pseudo_heritability'] - 1.0 for r_ in per])
        return res
    models, d, omega, G = _multi_models(ys, X, eig_L, method=method)
    h0 = np.array([m['h0_rss'] for m in models])
    n_p = n - (q + 1)
    rank, world = (coll.rank, coll.world) if coll is not None else (0, 1)
    from . import dist as mdist
    m0, m1 = mdist.shard_range(M, rank, world)
    rows_cap = max(256, int(max_store_bytes // (8 * (-(-n // 64) * 64))) // 256 * 256)
    rot = ctx.rot(eig_L['vectors'], min(rows_cap, max(m1 - m0, 1)))
    outs = {k: np.empty((P, m1 - m0)) for k in ('rss', 'f_stats', 'ps')}
    try:
        for c0 in range(m0, m1, rows_cap):
            c1 = min(c0 + rows_cap, m1)
            if own:
                g = ctx.geno(snps[c0:c1])
            elif (c0, c1) == (0, M):
                g = snps
            else:
                g = ctx.geno(snps.download(c0, c1 - c0))
            try:
                rot.load(g)
            finally:
                if g is not snps:
                    g.close()
            part = ctx.scan_multi(rot, d, omega, G, h0, n_p)
            for k in outs:
                outs[k][:, c0 - m0:c1 - m0] = part[k]
    finally:
        rot.close()
    if coll is not None and world > 1:
        count = max(b - a for a, b in (mdist.shard_range(M, r, world) for r in range(world)))
        for k in outs:
            blk = np.full((P, count), np.nan)
            blk[:, :m1 - m0] = outs[k]
            gathered = coll.allgather(blk.reshape(-1)).reshape(world, P, count)
            outs[k] = np.concatenate([gathered[r][:, :mdist.shard_range(M, r, world)[1] - mdist.shard_range(M, r, world)[0]]
                                      for r in range(world)], axis=1)
    res = {'ps': outs['ps'], 'f_stats': outs['f_stats'], 'rss': outs['rss'],
           'var_perc': 1 - outs['rss'] / h0[:, None], 'h0_rss': h0, 'h0_betas': [m['h0_betas'] for m in models]}
    for k in ('pseudo_heritability', 've', 'vg', 'max_ll', 'delta'):
        res[k] = np.array([m[k] for m in models])
    return res


def emma(snps, phenotypes, K, cofactors=None, ctx=None):
    """:1725-1745 -- exact EMMA per SNP (one N x N device eigh per SNP; short lists only)."""
    lmm = LinearMixedModel(phenotypes, ctx=ctx)
    lmm.add_random_effect(K)
    for cofactor in _cofactor_list(cofactors):
        lmm.add_factor(cofactor)
    return lmm.expedited_REML_t_test(list(np.asarray(snps)))


def linear_model(snps, phenotypes, cofactors=None, ctx=None):
    """:3168-3183 -- standard linear model GWAS (no kinship)."""
    lm_ = LinearModel(phenotypes, ctx=ctx)
    for cofactor in _cofactor_list(cofactors):
        lm_.add_factor(cofactor)
    return lm_.fast_f_test(snps)


# ---------------------------------------------------------------------- MLMM
def _log_choose_(n, k):
    """:1885-1892."""
    if k == 0 or n == k:
        return 0
    if n < k:
        raise Exception('Out of range.')
    return np.sum(np.log(np.arange(n, n - k, -1))) - np.sum(np.log(np.arange(k, 0, -1)))


def _calc_bic_(ll, num_snps, num_par, n):
    """:1895-1901."""
    bic = -2 * ll + num_par * np.log(n)
    extended_bic = bic + 2 * _log_choose_(num_snps, num_par - 2)
    modified_bic = bic + 2 * num_par * np.log(num_snps / 2.2 - 1)
    return (bic, extended_bic, modified_bic)


def _opt_fw_bw_(vals, max_num_cofactors, good):
    """The forward/backward optimum search shared by 'mbonf' and 'min_cof_ppa' (:2002-2051)."""
    fw = np.arange(max_num_cofactors + 1)
    for i in range(max_num_cofactors + 1):
        if not good(vals[i]):
            fw[i] = -1
    fw_i = int(fw.argmax())
    if max_num_cofactors > 1 and len(vals) > max_num_cofactors + 1:
        shift = max_num_cofactors + 1
        bw = np.arange(max_num_cofactors - 1, 0, -1)
        for i in range(len(bw)):
            if not good(vals[i + shift]):
                bw[i] = -1
        bw_max = bw[int(bw.argmax())]
        bw_i = int(bw.argmax()) + shift
        if bw_max == fw[fw_i]:
            return bw_i if vals[fw_i] > vals[bw_i] else fw_i
        return bw_i if bw_max > fw[fw_i] else fw_i
    return fw_i


def _analyze_opt_criterias_(criterias, sign_threshold, max_num_cofactors, ppa_threshold=0.5):
    """:1984-2066 without the plotting: optimal step index per criterion."""
    ret = {}
    for c in criterias:
        if c == 'bonf':
            opt_list = np.arange(max_num_cofactors + 1)
            for i, pval in enumerate(criterias['bonf'][:max_num_cofactors + 1]):
                if pval > sign_threshold:
                    opt_list[i] = -1
            ret[c] = int(opt_list.argmax())
        elif c == 'mbonf':
            ret[c] = _opt_fw_bw_(criterias[c], max_num_cofactors, lambda v: not v > sign_threshold)
        elif c == 'min_cof_ppa':
            ret[c] = _opt_fw_bw_(criterias[c], max_num_cofactors, lambda v: not v < ppa_threshold)
        else:
            ret[c] = int(np.argmin(criterias[c]))                        # first minimum (:2054-2063)
    return ret


def mlmm(phenotypes, K, sd=None, num_steps=10, forward_backwards=True, sign_threshold=None, snp_priors=None,
         snp_choose_criteria='pval', emma_num=0, save_pvals=False, ctx=None, **kwargs):
    """:2543-2923 -- multi-locus mixed model: forward inclusion of the most significant SNP as a cofactor,
    then backward elimination; every step is one full EMMAX scan with q fixed-effect columns.  The
    genotypes are uploaded once and stay in HBM for all steps (the reference re-converts its Python
    list every step).  Plotting, K2 and file output are out of scope.  kwargs: snps, positions,
    chromosomes (or `sd` providing get_snps / get_positions / get_chr_list)."""
    import math
    if sd is not None:
        kwargs['snps'] = sd.get_snps()
        kwargs['positions'] = sd.get_positions()
        kwargs['chromosomes'] = sd.get_chr_list()
    all_snps = kinship._as_snp_matrix(kwargs['snps'])
    positions = list(kwargs['positions'])
    chromosomes = list(kwargs['chromosomes'])
    lmm = LinearMixedModel(phenotypes, ctx=ctx)
    lmm.add_random_effect(K)
    ctx = lmm.ctx
    num_snps = len(all_snps)
    all_priors = np.asarray(snp_priors, dtype=np.float64) if snp_priors is not None \
        else np.full(num_snps, 1.0 / num_snps)                           # :2579-2581
    if not sign_threshold:
        sign_threshold = 1.0 / (num_snps * 20.0)                         # :2583-2584
    active = list(range(num_snps))                                       # global ids still in the scan
    geno = ctx.geno(all_snps)                                            # resident for every step

    def scan_active():
        r = lmm._emmax_f_test_(geno, H_sqrt_inv, snp_priors=all_priors, emma_num=0, verbose=False, _delta=dev_delta())
        idx = np.asarray(active)
        out = {k: np.asarray(r[k])[idx] for k in ('ps', 'rss', 'var_perc', 'ppas', 'f_stats')}
        if emma_num > 0:                                                 # :1365-1377 on the active list
            order = np.argsort(out['ps'], kind='stable')[:emma_num]
            top = lmm.expedited_REML_t_test(list(all_snps[idx[order]]), eig_L=eig_L)
            for k2, pi in enumerate(order):
                for key in ('ps', 'f_stats', 'rss', 'var_perc'):
                    out[key][pi] = top[key][k2]
        return out

    def dev_delta():
        # the scan model of the current variance ratio is built on the device from K and delta (no H_sqrt_inv product)
        return reml_res['delta'] if (DEVICE_SCAN_MODEL and isinstance(ctx, _lib.Context)) else None

    def reestimate():
        # get_REML / get_ML use 100 grid points; no eig_R: every step would need a fresh N^3 eigh for its X.
        # H_sqrt_inv (N x N, 0.07 s per estimate at N = 5000: half of an mlmm step) is formed only where a scan needs it
        # (lazy_H): with the scan model built on the device from delta that is the indefinite-kinship fallback alone
        device = DEVICE_SCAN_MODEL and isinstance(ctx, _lib.Context)
        reml = lmm.get_estimates(eig_L, method='REML', ngrids=100, return_H=not device)
        ml = lmm.get_estimates(eig_L, method='ML', ngrids=100, return_H=False)
        return reml, ml

    def lazy_H():
        if reml_res.get('H_sqrt_inv') is None:
            wts = 1.0 / np.sqrt(np.asarray(eig_L['values'], dtype=np.float64) + reml_res['delta'])   # :898
            reml_res['H_sqrt_inv'] = wts[:, None] * np.asarray(eig_L['vectors'])
        return reml_res['H_sqrt_inv']

    def cofactor_stats():
        pvals, ppas, fstats = [], [], []
        for i, gid in enumerate(cofactor_ids):
            t = [all_snps[j] for k2, j in enumerate(cofactor_ids) if k2 != i]
            lmm.set_factors(t)
            r = lmm._emmax_f_test_(all_snps[gid:gid + 1], H_sqrt_inv, snp_priors=[cof_snp_priors[i]], emma_num=0,
                                   verbose=False, _delta=dev_delta())
            pvals.append(float(r['ps'][0])); ppas.append(float(r['ppas'][0])); fstats.append(float(r['f_stats'][0]))
        lmm.set_factors([all_snps[j] for j in cofactor_ids])
        return pvals, ppas, fstats

    try:
        step_info_list, cofactors, cofactor_ids, cof_snp_priors, ppa_cofactors = [], [], [], [], []
        num_par = 2
        num_pher_0 = 0
        eig_L = lmm._get_eigen_L_()
        reml_res, ml_res = reestimate()
        H_sqrt_inv = lazy_H
        ll, rss = ml_res['max_ll'], float(reml_res['rss'])
        criterias = {'ebics': [], 'mbics': [], 'bonf': [], 'mbonf': []}
        bic, extended_bic, modified_bic = _calc_bic_(ll, num_snps, num_par, lmm.n)
        criterias['ebics'].append(extended_bic); criterias['mbics'].append(modified_bic)
        max_cofactor_pval = 0
        criterias['mbonf'].append(max_cofactor_pval); criterias['bonf'].append(0)
        criterias['min_cof_ppa'] = [1]
        pherit = reml_res['pseudo_heritability']
        first_emmax_res = None

        def info(em):
            min_i, ppa_i = int(np.argmin(em['ps'])), int(np.argmax(em['ppas']))
            d = {'pseudo_heritability': pherit, 'rss': rss, 'reml_mahalanobis_rss': reml_res['mahalanobis_rss'],
                 'mahalanobis_rss': float(em['rss'][min_i]), 'll': ll, 'bic': bic, 'e_bic': extended_bic,
                 'm_bic': modified_bic, 'mbonf': max_cofactor_pval, 'cofactors': [tuple(c) for c in cofactors],
                 'cofactor_snps': [all_snps[j] for j in cofactor_ids], 'min_pval': float(em['ps'][min_i]),
                 'min_pval_chr_pos': (chromosomes[active[min_i]], positions[active[min_i]]),
                 'max_ppa': float(em['ppas'][ppa_i]), 'max_ppa_pval': float(em['ps'][ppa_i]),
                 'max_ppa_chr_pos': (chromosomes[active[ppa_i]], positions[active[ppa_i]]),
                 'ppa_cofactors': [tuple(c) for c in ppa_cofactors]}
            if save_pvals:
                d['ps'] = em['ps'].tolist()
            return d, min_i, ppa_i

        for step_i in range(1, num_steps + 1):                           # :2631
            em = scan_active()
            if step_i == 1:
                first_emmax_res = em
            step_info, min_i, ppa_i = info(em)
            criterias['bonf'].append(step_info['min_pval'])
            step_info_list.append(step_info)
            snp_i = min_i if snp_choose_criteria == 'pval' else ppa_i
            gid = active[snp_i]
            lmm.add_factor(all_snps[gid])                                # :2686
            cofactor_ids.append(gid)
            reml_res, ml_res = reestimate()
            H_sqrt_inv = lazy_H
            ll, rss = ml_res['max_ll'], float(reml_res['rss'])
            num_par += 1
            cof_snp_priors.append(all_priors[gid])
            ppa_cofactors.append([chromosomes[gid], positions[gid], step_info['max_ppa']])
            cofactors.append([chromosomes[gid], positions[gid], step_info['min_pval']])
            pvals, ppas, _f = cofactor_stats()                           # :2712-2730
            for i, pv in enumerate(pvals):
                cofactors[i][2] = -math.log10(pv)
                ppa_cofactors[i][2] = ppas[i]
            max_cofactor_pval = max(pvals)
            criterias['mbonf'].append(max_cofactor_pval)
            criterias['min_cof_ppa'].append(min(ppas))
            del active[snp_i]                                            # :2734-2741
            num_snps -= 1
            bic, extended_bic, modified_bic = _calc_bic_(ll, num_snps, num_par, lmm.n)
            criterias['ebics'].append(extended_bic); criterias['mbics'].append(modified_bic)
            pherit = reml_res['pseudo_heritability']
            if pherit < 0.001:                                           # :2760-2765
                if num_pher_0 < 1:
                    num_pher_0 += 1
                else:
                    break
        em = scan_active()                                               # :2767
        step_info, _, _ = info(em)
        step_info_list.append(step_info)
        max_num_cofactors = len(cofactors)

        if forward_backwards:                                            # :2814-2897
            while len(cofactor_ids) > 1:
                pvals, ppas, fstats = cofactor_stats()
                for i, pv in enumerate(pvals):
                    cofactors[i][2] = -math.log10(pv)
                i_rm = int(np.argmin(fstats)) if snp_choose_criteria == 'pval' else int(np.argmin(ppas))
                # (the reference leaves cof_snp_priors untouched here, :2837-2839; kept for identical ppas)
                del ppa_cofactors[i_rm], cofactor_ids[i_rm], cofactors[i_rm]
                lmm.set_factors([all_snps[j] for j in cofactor_ids])
                num_snps += 1
                reml_res, ml_res = reestimate()
                ll, rss = ml_res['max_ll'], float(reml_res['rss'])
                H_sqrt_inv = lazy_H
                num_par -= 1
                pvals, ppas, _f = cofactor_stats()
                for i, pv in enumerate(pvals):
                    cofactors[i][2] = -math.log10(pv)
                    ppa_cofactors[i][2] = ppas[i]
                max_cofactor_pval = max(pvals)
                criterias['mbonf'].append(max_cofactor_pval)
                criterias['min_cof_ppa'].append(min(ppas))
                bic, extended_bic, modified_bic = _calc_bic_(ll, num_snps, num_par, lmm.n)
                criterias['ebics'].append(extended_bic); criterias['mbics'].append(modified_bic)
                pherit = reml_res['pseudo_heritability']
                step_info_list.append({'pseudo_heritability': pherit, 'rss': rss,
                                       'reml_mahalanobis_rss': reml_res['mahalanobis_rss'], 'll': ll, 'bic': bic,
                                       'e_bic': extended_bic, 'm_bic': modified_bic, 'mbonf': max_cofactor_pval,
                                       'cofactors': [tuple(c) for c in cofactors],
                                       'cofactor_snps': [all_snps[j] for j in cofactor_ids],
                                       'mahalanobis_rss': None, 'min_pval': None, 'min_pval_chr_pos': None,
                                       'ppa_cofactors': [tuple(c) for c in ppa_cofactors]})
        opt_dict = _analyze_opt_criterias_(criterias, sign_threshold, max_num_cofactors)
    finally:
        geno.close()
    return {'step_info_list': step_info_list, 'first_emmax_res': first_emmax_res, 'opt_dict': opt_dict,
            'criterias': criterias}


def emmax_perm_test(snps, phenotypes, K, num_perm=100, perm_idx=None, ctx=None, H_sqrt_inv=None,
                    reference_indexing=False):
    """:1819-1841.  perm_idx / H_sqrt_inv / reference_indexing: see LinearMixedModel.emmax_permutations."""
    lmm = LinearMixedModel(phenotypes, ctx=ctx)
    lmm.add_random_effect(K)
    res = lmm.emmax_permutations(snps, num_perm, perm_idx=perm_idx, H_sqrt_inv=H_sqrt_inv,
                                 reference_indexing=reference_indexing)
    p_f_list = sorted(zip(res['min_ps'], res['max_f_stats']))
    res['threshold_05'] = p_f_list[len(p_f_list) // 20]                  # :1831
    return res
