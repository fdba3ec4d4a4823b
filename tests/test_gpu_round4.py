"""GPU tests of round 4: the surface closures of tests/_round4_cases.py on the device, the Cholesky-QR band reduction of K
against the Householder one, the interpolated REML search against the step-by-step one."""
import numpy as np
import pytest

import _round4_cases as r4
from conftest import load_extras, load_extras3

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from mixmogam_amd import _lib
    return _lib.get_context()


@pytest.fixture(scope="module")
def ex3():
    return load_extras3()


def test_fast_f_test_with_betas_vs_the_reference(ctx, ex3):
    r4.fast_f_test_with_betas(ctx, ex3)


def test_transformed_snps_with_betas_vs_the_reference(ctx, ex3):
    r4.transformed_snps_with_betas(ctx, ex3)


def test_emmax_multi_with_four_cofactors_vs_the_reference_loop(ctx, ex3):
    r4.emmax_multi_four_cofactors(ctx, ex3)


def test_ml_estimates_without_an_eigendecomposition_vs_the_reference(ctx):
    r4.ml_without_an_eigendecomposition(ctx, load_extras())


def test_ibd_kinship_from_pre_normalised_snps_datasets(ctx):
    r4.ibd_kinship_from_normalised_snps(ctx)


@pytest.mark.parametrize("n", [129, 300, 1000])
def test_band_reduction_cholesky_qr_panels_vs_householder_panels(ctx, monkeypatch, n):
    """reml_band.hip: the round-4 reduction (Cholesky-QR panels, basis-kernel orthogonal factor, own products) and the
    round-3 one (MMG_BAND_IMPL=hh: Householder panel steps + rocBLAS updates) are different orthogonal similarities of the
    same K: every likelihood sum agrees to 1e-10, at sizes with 1, 3 and 14 full panels plus the short tail block."""
    rng = np.random.RandomState(n)
    m = 4 * n
    f = np.clip(0.5 + 0.25 * rng.standard_normal((m, 3)), 0.05, 0.95)
    S = (rng.random_sample((m, n)) < f[:, rng.randint(0, 3, size=n)]).astype(np.float64)
    S = S[S.std(1) > 0]
    Z = (S - S.mean(1, keepdims=True)) / S.std(1, keepdims=True)
    K = Z.T @ Z / len(Z)
    X = np.column_stack([np.ones(n), rng.standard_normal(n)])
    y = rng.standard_normal(n) + Z[3]
    deltas = np.exp(np.linspace(-8, 8, 7))
    reml = ctx.reml(K, X, y)
    new = reml.sums(deltas, route="band")
    reml.close()
    monkeypatch.setenv("MMG_BAND_IMPL", "hh")
    reml = ctx.reml(K, X, y)
    old = reml.sums(deltas, route="band")
    reml.close()
    for i in range(4):
        assert np.max(np.abs(new[i] - old[i]) / np.maximum(np.abs(old[i]), 1.0)) < 1e-10, (n, i)


def test_band_reduction_falls_back_on_duplicated_individuals(ctx):
    """Two identical individuals make a panel of the reduction exactly rank deficient: Cholesky-QR cannot factor it, the
    device flag sends the whole reduction through the Householder path, and the sums still match dense float64."""
    from test_gpu_round3 import _reml_sums_f64
    rng = np.random.RandomState(12)
    n, m = 400, 900
    S = (rng.random_sample((m, n)) < 0.4).astype(np.float64)
    S[:, 7] = S[:, 3]                                          # individuals 3 and 7: the same genotypes
    S[:, 250] = S[:, 3]
    Z = (S - S.mean(1, keepdims=True)) / S.std(1, keepdims=True)
    K = Z.T @ Z / m
    X = np.ones((n, 1))
    y = rng.standard_normal(n)
    reml = ctx.reml(K, X, y)
    deltas = [0.05, 2.0]
    assert not reml.band_info()["ready"]
    got = reml.sums(deltas, route="band")
    info = reml.band_info()
    assert info["ready"] and info["householder_fallback"] and info["seconds"] > 0
    reml.close()
    small = ctx.reml(K[:100, :100], X[:100], y[:100])          # N < 256: AUTO is the Cholesky route until a band call was made
    assert not small.uses_band("auto")
    small.sums([1.0], route="band")
    assert small.uses_band("auto")
    small.close()
    for k, d in enumerate(deltas):
        want = _reml_sums_f64(K, X, y, d)
        for i in range(4):
            assert abs(got[i][k] - want[i]) <= 1e-9 * max(abs(want[i]), 1.0), (d, i, got[i][k], want[i])


def test_reml_search_on_the_interpolant_vs_step_by_step_on_the_device(ctx, monkeypatch):
    """get_estimates_eigen_free at N = 1,500: round 5's search (the grid and its refinement to a spacing of 0.1 factored in one
    sweep, sums on the grid and then on the refined nodes around the bracket from the kept factors, the search on the polynomial
    through the 20 nodes around each question) and round 4's two full calls (grid, 16 Chebyshev nodes) against one call per
    secant step -- the variance ratio to 1e-10, likelihood and variance components to 1e-9; and sums from kept factors equal
    sums that factor themselves, bit for bit."""
    from mixmogam_amd import linear_models as lm
    n, m = 1500, 4000
    g = ctx.geno(M=m, N=n).fill_structured(5, npop=3)
    acc = ctx.kinship_accumulator(n)
    acc.add_grm(g)
    K, cnt = acc.fetch()
    acc.close()
    K = K / cnt
    rng = np.random.RandomState(0)
    y = rng.standard_normal(n) + 1.5 * (K @ rng.standard_normal(n)) / np.sqrt(n) + g.download_rows([11])[0]
    g.close()
    lmm = lm.LinearMixedModel(list(y), ctx=ctx)
    lmm.add_random_effect(K)
    a = lmm.get_estimates_eigen_free()
    a.pop("reml").close()
    n_fine = 50 * 4 + 1 + 2 * lm._SpectralSumsChol.FINE_PAD
    assert a["n_device_calls"] == 2 and n_fine + 20 < a["n_factorisations"] < n_fine + 60
    # the kept factors give the sums a self-factoring call gives
    reml = ctx.reml(K, np.ones((n, 1)), y)
    try:
        d = np.exp(np.linspace(-3.0, 3.0, 31))
        plain = reml.sums(d)
        reml.band_factor(d)
        kept = reml.sums(d[::3])
        for i in range(4):
            assert np.array_equal(kept[i], plain[i][::3]), i
        mixed = reml.sums(np.r_[d[:2], 0.777])                 # one value that was not kept: the call factors for itself
        assert np.array_equal(mixed[0][:2], plain[0][:2])
        with pytest.raises(Exception):
            reml.band_factor(np.array([1.0, -1.0]))
    finally:
        reml.close()
    monkeypatch.setattr(lm._SpectralSumsChol, "FINE_GRID", False)
    a2 = lmm.get_estimates_eigen_free()
    a2.pop("reml").close()
    assert a2["n_device_calls"] == 2
    monkeypatch.setattr(lm._SpectralSumsChol, "prepare_interval", lambda self, lo, hi: None)
    b = lmm.get_estimates_eigen_free()
    b.pop("reml").close()
    assert b["n_device_calls"] > 4 and 1e-3 < b["delta"] < 1e3
    for est in (a, a2):
        assert abs(est["delta"] / b["delta"] - 1) < 1e-10
        for k in ("max_ll", "ve", "vg"):
            assert abs(est[k] - b[k]) <= 1e-9 * max(1.0, abs(b[k])), k


def test_fp4_twin_of_the_store_follows_every_write_path(ctx, monkeypatch):
    """The IBS counts of a binary store run on its E2M1 twin (no image pass inside the call).  The twin must be in step
    after every way of writing a store: the synthetic generators, int8 / float / bit-packed uploads, partial uploads, a reset
    to fewer rows and a refill -- each time the counts equal the int64 numpy product, and they equal what the round-3 path
    (scratch image per call, MMG_FP4_TWIN=0 stores) gives."""
    from mixmogam_amd import _lib
    rng = np.random.RandomState(4)
    n, m = 333, 2100

    def counts_ref(S):
        x = 2 * S.astype(np.int64) - 1
        return x.T @ x

    S = (rng.random_sample((m, n)) < rng.uniform(0.1, 0.9, size=(m, 1))).astype(np.int8)
    g = ctx.geno(S)                                            # int8 upload
    assert np.array_equal(ctx.kinship_ibs_counts(g), counts_ref(S))
    with pytest.raises(_lib.MixmogamHipError):                 # no image pass ran in that call
        ctx.kernel_ms("pack")
    S2 = S.copy()
    S2[700:1300] = (rng.random_sample((600, n)) < 0.3).astype(np.int8)
    g.upload(S2[700:1300], 700)                                # partial rewrite
    assert np.array_equal(ctx.kinship_ibs_counts(g), counts_ref(S2))
    g.reset(1500)                                              # fewer rows: the tail must read as zeros in the twin too
    g.upload(S2[:1500], 0)
    assert np.array_equal(ctx.kinship_ibs_counts(g), counts_ref(S2[:1500]))
    g.close()
    for dt in (np.float32, np.float64):
        g = ctx.geno(S.astype(dt))
        assert np.array_equal(ctx.kinship_ibs_counts(g), counts_ref(S))
        g.close()
    g = ctx.geno(M=m, N=n)
    g.upload_packed(_lib.pack_genotypes(S, 1), 1)              # bit-packed rows, expanded on the device
    assert np.array_equal(ctx.kinship_ibs_counts(g), counts_ref(S))
    g.close()
    for filler in ("hash", "structured"):
        g = ctx.geno(M=m, N=n)
        g.fill_hash(11) if filler == "hash" else g.fill_structured(11, npop=3)
        assert np.array_equal(ctx.kinship_ibs_counts(g), counts_ref(g.download())), filler
        g.close()
    # a store that stops being binary: the twin is ignored, the int8 kernel runs
    T = S.copy()
    T[5, 7] = 2
    g = ctx.geno(T)
    x = 2 * T.astype(np.int64) - 1
    assert np.array_equal(ctx.kinship_ibs_counts(g), x.T @ x)
    g.close()


def test_own_triangular_inverse_equals_the_library_one(ctx, monkeypatch):
    """reml_chol.hip:tri_inv_own (MMG_REML_TRTRI=own: block columns from the stored inverses of the diagonal blocks, tall x 64x64
    and lower-triangular x tall products on the matrix pipe) against the default rocSOLVER / rocBLAS inverse: the likelihood
    sums of the Cholesky route (their trace term is |L^-1|_F^2) to 1e-12 at a size with partial blocks."""
    rng = np.random.RandomState(2)
    n = 1100
    B = rng.standard_normal((n, 400))
    K = B @ B.T / 400 + 0.02 * np.diag(rng.random_sample(n))
    X = np.column_stack([np.ones(n), rng.standard_normal(n)])
    y = rng.standard_normal(n)
    deltas = [1e-2, 1.0, 30.0]
    reml = ctx.reml(K, X, y)
    ref = reml.sums(deltas, route="chol")
    reml.close()
    monkeypatch.setenv("MMG_REML_TRTRI", "own")
    reml = ctx.reml(K, X, y)
    got = reml.sums(deltas, route="chol")
    reml.close()
    for i in range(4):
        assert np.max(np.abs(got[i] - ref[i]) / np.maximum(np.abs(ref[i]), 1.0)) < 1e-12, i


def _grm_f64(s):
    """kinship.py:63-69 in float64 on the host: sum_m z_m z_m', z = (s - mean) / std (population std)."""
    x = s.astype(np.float64)
    z = (x - x.mean(1, keepdims=True)) / x.std(1, keepdims=True)
    return z.T @ z


def test_grm_accumulator_runs_of_calls_share_the_digit_planes(ctx):
    """mmg_kin_acc_add_grm keeps a run of calls in the int32 planes (one combine per run, include/mixmogam_hip.h): calls with
    like weights join, a call whose weights do not fit ends the run, readers of the accumulator see everything."""
    n, m = 200, 65536                                           # >= 65,536 binary SNPs: the one-pass four-plane kernel
    rng = np.random.RandomState(11)

    def chunk(lo, hi, rows=m):
        f = rng.uniform(lo, hi, rows)
        s = (rng.random_sample((rows, n)) < f[:, None]).astype(np.int8)
        s[:, 0] = 0; s[:, 1] = 1                                # no SNP without variation
        return s

    a, b, c = chunk(0.2, 0.8), chunk(0.2, 0.8), chunk(0.2, 0.8, 3000)
    rare = chunk(0.004, 0.5, 3000)                              # weights up to ~ 1 / 0.004: beyond the run's cap
    acc = ctx.kinship_accumulator(n)
    try:
        for s in (a, b, c):
            g = ctx.geno(s)
            acc.add_grm(g)
            g.close()
        # the second call joined the first one's run, and so did the short one (a chromosome's tail group: four planes
        # like the run it joins, not the five a short call takes on its own)
        assert acc.pending() == 2 * m + len(c)
        k_ab, cnt = acc.fetch()
        assert cnt == 2 * m + len(c) and acc.pending() == 0
        want = _grm_f64(a) + _grm_f64(b) + _grm_f64(c)
        assert np.abs(k_ab - want).max() <= 2e-9 * np.abs(want).max()
        # a small call (five planes, one GEMM per plane) after a fetch starts a run of its own; rare variants end it
        for s in (c, rare):
            g = ctx.geno(s)
            acc.add_grm(g)
            g.close()
        assert acc.pending() == len(rare)
        f = acc.scale_k()                                       # reads the accumulator: combines first
        assert acc.pending() == 0
        k_all, cnt = acc.fetch()
        assert cnt == 2 * m + 2 * len(c) + len(rare)
        want = want + _grm_f64(c) + _grm_f64(rare)
        scalar = (n - 1) / (np.trace(want) - want.sum() / n)
        assert abs(f / scalar - 1) < 1e-9
        assert np.abs(k_all - want * scalar).max() <= 2e-9 * np.abs(want * scalar).max()
    finally:
        acc.close()


def test_grm_run_of_five_plane_calls_after_a_longer_one_pass_call(ctx):
    """A >= 2^16-SNP call whose weight range forces the fifth plane JOINS a five-plane run: its digit images are larger than
    the ones the run's first call allocated, while the per-SNP buffers (sized by an earlier, longer one-pass call) already
    fit -- the workspace has to grow for the plan that runs, not the one the SNP count alone suggests (api.hip,
    kinship_grm_i8_into: until round 4 the images of the third call here overran their buffer)."""
    n = 200
    rng = np.random.RandomState(12)

    def chunk(lo, hi, rows):
        f = rng.uniform(lo, hi, rows)
        s = (rng.random_sample((rows, n)) < f[:, None]).astype(np.int8)
        s[:, 0] = 0; s[:, 1] = 1
        return s

    even = chunk(0.2, 0.8, 140000)                              # one pass, four planes, no images
    wide1 = chunk(0.004, 0.5, 70000)                            # wmax / wmin > 64: five planes, one GEMM per plane
    wide2 = chunk(0.004, 0.5, 100000)
    wide2[0] = wide1[0]                                         # the same largest weight: the third call joins the second's run
    acc = ctx.kinship_accumulator(n)
    try:
        want = 0.0
        for s in (even, wide1, wide2):
            g = ctx.geno(s)
            acc.add_grm(g)
            g.close()
            want = want + _grm_f64(s)
        assert acc.pending() == len(wide1) + len(wide2)         # the third call joined the second one's run
        k, cnt = acc.fetch()
        assert cnt == len(even) + len(wide1) + len(wide2)
        assert np.abs(k - want).max() <= 2e-9 * np.abs(want).max()
    finally:
        acc.close()


@pytest.mark.parametrize("m", [3000, 70000])
def test_grm_of_a_diploid_store_through_the_centred_alphabet(ctx, m):
    """0 / 1 / 2 stores are read as s - 1 in the exact GRM (api.hip, kinship_grm_i8_into: 7-bit digits, the plane counts of a
    binary store): five planes for a short call, four from 2^16 SNPs on; K against the float64 product of the UNSHIFTED
    genotypes (kinship.py:63-69), packed 2-bit upload and int8 upload alike."""
    from mixmogam_amd import _lib
    n = 203                                                     # 53 padding individuals: they must stay out of the sums
    rng = np.random.RandomState(m)
    f = rng.uniform(0.1, 0.9, m)
    s = ((rng.random_sample((m, n)) < f[:, None]).astype(np.int8) + (rng.random_sample((m, n)) < f[:, None]).astype(np.int8))
    s[:, 0] = 0; s[:, 1] = 2
    want = _grm_f64(s)
    for packed in (False, True):
        if packed:
            g = ctx.geno(M=m, N=n)
            g.upload_packed(_lib.pack_genotypes(s, 2), 2)
        else:
            g = ctx.geno(s)
        acc = ctx.kinship_accumulator(n)
        try:
            acc.add_grm(g)
            k, cnt = acc.fetch()
        finally:
            acc.close()
            g.close()
        assert cnt == m
        assert np.abs(k - want).max() <= 2e-9 * np.abs(want).max(), packed


@pytest.mark.parametrize("n, m", [(1000, 140000), (300, 130), (300, 400), (640, 1300)])
def test_grm_kernel_generations_are_bit_for_bit_identical(n, m):
    """The shipped one-pass GRM kernel (round 5: kinship_grm4j_kernel, every slice scales its own operands) against round 3's
    quadrant kernel (MMG_GRM4_LAYOUT=quad) and round 4's row strips (=strips: Q tiles in three LDS slots): every plane is an
    exact integer sum, so the accumulated matrices must be identical -- at a size with padding individuals, and at jobs of one,
    two and a few K steps (M = 130, 400, 1300), where the stage cursor of the pipeline is clamped from the first step on.  The
    switch is read once per process: tools/grm4_layouts.py runs each layout in a process of its own."""
    import os, re, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "grm4_layouts.py"), str(n), str(m)], capture_output=True,
                         text=True, timeout=300, check=True).stdout
    digests = re.findall(r"layout (\w+)\s*:.*sha1 ([0-9a-f]+)", out)
    assert [d[0] for d in digests] == ["jit", "quad", "strips"], out
    assert digests[0][1] == digests[1][1] == digests[2][1], out


def test_grm_five_planes_one_pass_plus_one_image_equals_five_images_bit_for_bit():
    """Five weight planes of a binary store: the one-pass kernel on planes 1-4 + an image GEMM for plane 0 (api.hip, mode 2)
    against five image GEMMs (MMG_GRM_HYBRID=0) -- exact integer planes, the same combine pass: identical matrices.  N = 1000
    has padding individuals; 3,000 SNPs is a call shorter than the kernel's pipeline was ever given before (24 K steps)."""
    import os, re, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "grm_five_planes.py"), "1000", "140000", "300", "3000",
                          "77", "100"], capture_output=True, text=True, timeout=300, check=True).stdout
    got = re.findall(r"mode (\w+)\s+N=(\d+) M=(\d+):.*sha1 ([0-9a-f]+)", out)
    assert [g[0] for g in got] == ["hybrid"] * 3 + ["images"] * 3, out
    for a, b in zip(got[:3], got[3:]):
        assert a[1:] == b[1:], out


@pytest.mark.parametrize("n,m", [(504, 2), (1024, 3), (2048, 3)])
def test_band_route_on_a_kinship_of_rank_three(ctx, n, m):
    """A kinship from two or three SNPs has rank <= 4: every panel of the band reduction is rank deficient (the Cholesky-QR
    panels hand over to the Householder ones), and every column past the rank is the rounding noise of the one before it,
    down to 1e-160 -- where the squares that make up a column norm underflow.  Before reml_band.hip's HH_TINY2 guard the
    reflectors of such columns stopped being orthogonal (tau |v|^2 = 2.59) and the sums involving y were off by 2e-5, p-values
    by up to 6e-3 (found by tools/random_parity.py).  Sums against the spectral form on the host, p-values against the oracle."""
    from mixmogam_amd import kinship, linear_models as lm
    from oracle import emmax_oracle as orc
    rng = np.random.RandomState(n + m)
    snps = (rng.random_sample((m, n)) < rng.uniform(0.2, 0.8, m)[:, None]).astype(np.int8)
    y = rng.standard_normal(n) + snps[0]
    K = kinship.calc_ibs_kinship(snps, ctx=ctx)
    Ks = kinship.scale_k(K)
    X = np.ones((n, 1))
    lam, U = np.linalg.eigh(Ks)
    want = lm._SpectralSumsL({"values": lam[::-1], "vectors": U.T[::-1]}, X, y).at(np.array([1e-3, 1.0, 50.0]))
    r = ctx.reml(Ks, X, y)
    try:
        got = r.sums(np.array([1e-3, 1.0, 50.0]), route="band")
        assert r.band_info()["householder_fallback"]
    finally:
        r.close()
    for k in range(4):
        assert np.max(np.abs(got[k] / want[k] - 1)) < 1e-8, k
    res = lm.emmax(snps, list(y), K, ctx=ctx)
    ref = orc.emmax(snps, y, K)
    assert abs(res["pseudo_heritability"] - ref["pseudo_heritability"]) < 1e-6
    assert np.max(np.abs(res["ps"] / ref["ps"] - 1)) < 1e-6


def test_random_shapes_against_the_oracle():
    """tools/random_parity.py: 40 random small problems (3..700 individuals, 1..3000 SNPs, binary / 0-1-2 / signed genotypes,
    0..2 cofactors) through kinship, emmax(), linear_model() and emmax_multi() against the float64 oracle -- the sweep that
    found the rank-deficient-kinship failure above."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "random_parity.py"), "40", "3"], capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0 and "failures: 0" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]


def test_scan_exact_tier_on_a_kinship_of_twelve_genotype_classes(ctx):
    """Every individual is one of 12 genotype vectors: K has rank <= 12 and every SNP lies in its span, so den = s'P s is
    orders of magnitude below sum s^2 max |P| -- what 27 bits of the LARGEST entry cannot resolve (p off by 6e-6 at F = 7 with
    all four digit planes; tools/random_parity2.py found it).  Equal entries of P also round alike, so the errors of a SNP's
    products add up instead of averaging: the tier's sample check must notice that the independent-roundings model is 100 x
    off and switch to the worst-case bound.  With the tier: p <= 1e-6 of the oracle; the SNPs it recomputed are reported."""
    from mixmogam_amd import kinship, linear_models as lm
    from oracle import emmax_oracle as orc
    rng = np.random.RandomState(5)
    n, m = 1888, 2298
    base = (rng.random_sample((m, 12)) < rng.uniform(0.05, 0.95, m)[:, None]).astype(np.int8)
    snps = base[:, rng.randint(0, 12, n)]
    snps = snps[snps.std(1) > 0]
    y = rng.standard_normal(n) + 0.8 * snps[3] + 0.5 * snps[10]
    K = kinship.calc_ibs_kinship(snps, ctx=ctx)
    ref = orc.emmax(snps, y, K)
    res = lm.emmax(snps, list(y), K, ctx=ctx)
    st = ctx.scan_last_stats()
    assert st["n_exact"] > len(snps) // 2, st
    assert np.max(np.abs(res["ps"] / ref["ps"] - 1)) < 1e-6
    wb = lm.emmax(snps, list(y), K, with_betas=True, ctx=ctx)             # the eigen route's model takes the same tier
    assert np.max(np.abs(wb["ps"] / ref["ps"] - 1)) < 1e-6
    # several phenotypes on such a kinship: the rotated path's 27-bit rows have the same limit; emmax_multi hands over to the
    # per-phenotype scans (one eigendecomposition, one resident store)
    ys = np.vstack([y, rng.standard_normal(n) + snps[7]])
    mr = lm.emmax_multi(snps, ys, K, ctx=ctx)
    mo = orc.emmax_multi(snps, ys, K)
    assert np.max(np.abs(mr["ps"] / mo["ps"] - 1)) < 1e-6
    assert np.max(np.abs(mr["pseudo_heritability"] - mo["pseudo_heritability"])) < 1e-6


def test_ibs_kinship_converted_and_scaled_on_the_device(ctx):
    """mmg_kinship_ibs_f64 (calc_ibs_kinship above 2048 individuals): unscaled, bit for bit the host expression on the exact
    counts; scaled, kinship.scale_k to rounding."""
    from mixmogam_amd import kinship
    rng = np.random.RandomState(9)
    n, m = 2500, 3000
    snps = (rng.random_sample((m, n)) < rng.uniform(0.1, 0.9, m)[:, None]).astype(np.int8)
    g = ctx.geno(snps)
    try:
        counts = ctx.kinship_ibs_counts(g)
        host = counts.astype(np.float64) / (2 * float(m)) + 0.5
        assert np.array_equal(ctx.kinship_ibs(g, scaled=False), host)
        want = kinship.scale_k(host)
        got = ctx.kinship_ibs(g, scaled=True)
        assert np.max(np.abs(got - want)) <= 1e-13 * np.max(np.abs(want))
        assert np.array_equal(kinship.calc_ibs_kinship(None, ctx=ctx, geno=g), got)
    finally:
        g.close()


def test_diploid_ibs_kinship_combined_on_the_device(ctx):
    """mmg_kinship_ibs_diploid_f64 (calc_ibs_kinship(snps_data_format='diploid_int') above 2048 individuals) against the host
    mirror's arithmetic on the two indicator products: unscaled bit for bit, scaled to rounding."""
    from mixmogam_amd import kinship
    rng = np.random.RandomState(10)
    n, m = 2300, 1500
    f = rng.uniform(0.1, 0.9, m)
    snps = ((rng.random_sample((m, n)) < f[:, None]).astype(np.int8) + (rng.random_sample((m, n)) < f[:, None]).astype(np.int8))
    g = ctx.geno(snps)
    try:
        c12 = ctx.kinship_indicator_counts(g, 1) + ctx.kinship_indicator_counts(g, 2)
        r = np.diag(c12).astype(np.float64)
        k = float(m) - 0.5 * (r[:, None] + r[None, :] - 2.0 * c12)
        np.fill_diagonal(k, 0.0)
        k = k / float(m) + np.eye(n)
        assert np.array_equal(ctx.kinship_ibs_diploid(g, scaled=False), k)
        want = kinship.scale_k(k)
        got = kinship.calc_ibs_kinship(None, snps_data_format='diploid_int', ctx=ctx, geno=g)
        assert np.max(np.abs(got - want)) <= 1e-13 * np.max(np.abs(want))
    finally:
        g.close()


def test_with_betas_without_an_eigendecomposition(ctx, monkeypatch):
    """emmax(with_betas=True) from N = 256 up: the covariates' matrix C = (X'V^-1 X)^-1 X'V^-1 comes with the device-built scan
    model (mmg_reml_scan_model_c) instead of R^-1 Q'H from H_sqrt_inv -- same p-values and coefficients as the eigen route."""
    from mixmogam_amd import kinship, linear_models as lm
    rng = np.random.RandomState(12)
    n, m = 600, 3000
    snps = (rng.random_sample((m, n)) < rng.uniform(0.1, 0.9, m)[:, None]).astype(np.int8)
    y = rng.standard_normal(n) + snps[4] - 0.6 * snps[40]
    cof = [list(rng.standard_normal(n)), list(snps[7].astype(float))]
    K = kinship.calc_ibs_kinship(snps, ctx=ctx)
    free = lm.emmax(snps, list(y), K, cofactors=cof, with_betas=True, ctx=ctx)
    assert free["timings"]["eig_L"] == 0.0
    monkeypatch.setattr(lm, "EIGEN_FREE_MIN_N", 1 << 30)
    eig = lm.emmax(snps, list(y), K, cofactors=cof, with_betas=True, ctx=ctx)
    assert eig["timings"]["eig_L"] > 0.0
    assert np.max(np.abs(free["ps"] / eig["ps"] - 1)) < 1e-6
    # SNP 7 is one of the cofactors: the design loses rank with it and the SNP keeps the null model's three coefficients (:236-239)
    assert list(free["betas"][7]) == list(free["h0_betas"]) and list(eig["betas"][7]) == list(eig["h0_betas"])
    rest = [j for j in range(m) if j != 7]
    bf, be = np.asarray([free["betas"][j] for j in rest]), np.asarray([eig["betas"][j] for j in rest])
    assert bf.shape == be.shape == (m - 1, 4)
    assert np.max(np.abs(bf - be)) < 1e-7 * max(1.0, np.max(np.abs(be)))
