"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (/root/reference) in this container.

Run once in the build container:  python tests/golden/make_golden.py
The reference is imported through refshim.py (2to3 on the fly, nothing copied); this script is
a no-op where /root/reference is absent (e.g. the GPU box).  Fixtures hold DATA only: seeded
inputs and the reference's outputs, in two modes -- 'lit' (the reference as written, fp32
where it says 'single') and 'dbl' ('single' -> 'double', the oracle the 1e-6 target refers to).
"""
import contextlib
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refshim  # noqa: E402


def structured_genotypes(rng, n, m, npop=3, fst=0.1):
    """Balding-Nichols style sub-populations so that the REML optimum is interior."""
    anc = rng.uniform(0.1, 0.9, size=m)
    a = anc * (1 - fst) / fst
    b = (1 - anc) * (1 - fst) / fst
    pop_p = rng.beta(a[:, None], b[:, None], size=(m, npop))
    pops = rng.randint(0, npop, size=n)
    snps = (rng.random_sample((m, n)) < pop_p[:, pops]).astype(np.int8)
    keep = (snps.sum(1) > 0) & (snps.sum(1) < n)
    return snps[keep]


def bernoulli_genotypes(rng, n, m):
    """simulations.py:21-24."""
    snps = np.round(rng.random_sample((m, n))).astype(np.int8)
    return snps[snps.sum(1) > 0]


def phenotype(rng, snps, h2=0.6, ncausal=10):
    m, n = snps.shape
    idx = rng.choice(m, ncausal, replace=False)
    eff = rng.exponential(1.0, size=ncausal)
    g = eff @ snps[idx].astype(np.float64)
    e = rng.randn(n)
    y = g + e * np.sqrt((1 - h2) / h2 * g.var(ddof=1) / e.var(ddof=1))
    return (y - y.mean()) / y.std()


def quiet(fn, *a, **kw):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **kw)


def run_case(mods, snps, y, cof, nperm, perm_seed):
    lm, kin = mods['linear_models'], mods['kinship']
    out = {}
    snp_list = list(snps)
    k_unscaled = np.asarray(quiet(kin.calc_ibs_kinship, snp_list, scaled=False))
    k_ibs = np.asarray(quiet(kin.calc_ibs_kinship, snp_list))
    out['ibs_unscaled'] = k_unscaled
    out['ibs_scaled'] = k_ibs
    out['ibd_scaled'] = np.asarray(quiet(kin.calc_ibd_kinship, snp_list))
    out['scale_k_of_ibs_unscaled'] = np.asarray(kin.scale_k(k_unscaled))
    res = quiet(lm.emmax, snp_list, list(y), k_ibs, cofactors=cof)
    for k in ('ps', 'f_stats', 'rss', 'var_perc'):
        out['emmax_' + k] = np.asarray(res[k], dtype=np.float64).reshape(-1)
    out['emmax_h0_rss'] = np.asarray(res['h0_rss'], dtype=np.float64).reshape(-1)
    out['emmax_h0_betas'] = np.asarray(res['h0_betas'], dtype=np.float64).reshape(-1)
    for k in ('pseudo_heritability', 've', 'vg', 'max_ll'):
        out['emmax_' + k] = np.float64(res[k])
    reml = quiet(lm.get_emma_reml_estimates, list(y), k_ibs, cofactors=cof)
    for k in ('max_ll', 'delta', 've', 'vg', 'pseudo_heritability'):
        out['reml_' + k] = np.float64(reml[k])
    out['reml_beta'] = np.asarray(reml['beta'], dtype=np.float64).reshape(-1)
    out['reml_mahalanobis_rss'] = np.asarray(reml['mahalanobis_rss'], dtype=np.float64).reshape(-1)
    out['eig_L_values'] = np.asarray(reml['eig_L']['values'], dtype=np.float64)
    lmm = reml['lmm']
    out['eig_R_values'] = np.asarray(lmm._get_eigen_R_(X=lmm.X)['values'], dtype=np.float64)
    H = np.asarray(reml['H_sqrt_inv'], dtype=np.float64)
    probe = np.random.RandomState(99).randn(H.shape[0], 3)
    out['HtH_probe'] = H.T @ (H @ probe)
    # with_betas variant of the scan
    lmm2 = lm.LinearMixedModel(list(y))
    lmm2.add_random_effect(k_ibs)
    if cof is not None:
        for c in cof:
            lmm2.add_factor(c)
    wb = quiet(lmm2.emmax_f_test, snp_list[:200], with_betas=True, emma_num=0)
    out['wb_ps'] = np.asarray(wb['ps'], dtype=np.float64)
    out['wb_betas'] = np.asarray(wb['betas'], dtype=np.float64)
    # plain linear model (LinearModel.fast_f_test through linear_model(), linear_models.py:3168, :196)
    lres = quiet(lm.linear_model, snp_list, list(y), cofactors=cof)
    for k in ('ps', 'f_stats', 'rss'):
        out['lm_' + k] = np.asarray(lres[k], dtype=np.float64).reshape(-1)
    out['lm_h0_rss'] = np.asarray(lres['h0_rss'], dtype=np.float64).reshape(-1)
    # exact-EMMA refinement of the top hits (emma_num > 0, linear_models.py:1365-1377)
    if len(y) <= 160:
        eres = quiet(lm.emmax, snp_list, list(y), k_ibs, cofactors=cof, emma_num=15)
        for k in ('ps', 'f_stats', 'rss', 'var_perc'):
            out['emma15_' + k] = np.asarray(eres[k], dtype=np.float64).reshape(-1)
        # 'diploid_int' IBS kinship (kinship.py:33-41) on 0/1/2 genotypes
        half = len(snps) // 2
        dip = (snps[:half] + snps[half:2 * half]).astype(np.int8)
        out['dip_ibs_unscaled'] = np.asarray(quiet(kin.calc_ibs_kinship, list(dip), snps_data_format='diploid_int',
                                                   scaled=False), dtype=np.float64)
    # MLMM forward/backward (linear_models.py:2543-2923); plotting/reporting helpers are not exercised
    if cof is None and len(y) <= 160:
        import sys as _sys
        _sys.modules['gwaResults'] = mods['gwaResults']
        lm.agr.calc_median = lambda ps, exp_median=0.5: float(np.median(ps) - exp_median)   # py2 int division inside
        lm.agr.calc_ks_stats = lambda ps, exp_dist=None: {'D': 0.0, 'p_val': 1.0}
        m_all = len(snp_list)
        mres = quiet(lm.mlmm, list(y), k_ibs, num_steps=3, forward_backwards=True, file_prefix=None,
                     snps=list(snp_list), positions=list(range(m_all)), chromosomes=[1] * m_all,
                     mafs=[0.3] * m_all, macs=[30] * m_all)
        keys = ('pseudo_heritability', 'll', 'bic', 'e_bic', 'm_bic', 'mbonf', 'min_pval', 'rss',
                'reml_mahalanobis_rss', 'mahalanobis_rss')
        tab = [[np.nan if si[k] is None else float(np.asarray(si[k]).reshape(-1)[0]) for k in keys]
               for si in mres['step_info_list']]
        out['mlmm_steps'] = np.asarray(tab)
        out['mlmm_cof_pos'] = np.asarray([[c[1] for c in si['cofactors']] + [-1] * (3 - len(si['cofactors']))
                                          for si in mres['step_info_list']])
        out['mlmm_cof_mlogp'] = np.asarray([[c[2] for c in si['cofactors']] + [np.nan] * (3 - len(si['cofactors']))
                                            for si in mres['step_info_list']])
        for c in ('ebics', 'mbics', 'bonf', 'mbonf', 'min_cof_ppa'):
            out['mlmm_opt_' + c] = np.int64(mres['opt_dict'][c])
        out['mlmm_first_ps'] = np.asarray(mres['first_emmax_res']['ps'], dtype=np.float64)
    # permutations (intercept only in the reference: h0_X * list-of-floats needs q == 1)
    if cof is None and nperm:
        lmm3 = lm.LinearMixedModel(list(y))
        lmm3.add_random_effect(k_ibs)
        n = len(y)
        np.random.seed(perm_seed)
        idx = np.asmatrix(np.arange(n).reshape(n, 1))
        perm_idx = []
        for _ in range(nperm):
            np.random.shuffle(idx)
            perm_idx.append(np.asarray(idx).reshape(-1).copy())
        np.random.seed(perm_seed)
        pr = quiet(lmm3._emmax_permutations_, [s.astype(np.float64) for s in snps], k_ibs,
                   reml['H_sqrt_inv'], num_perm=nperm)
        out['perm_idx'] = np.asarray(perm_idx, dtype=np.int32)
        # H_sqrt_inv is an INPUT of _emmax_permutations_ (:1125) and its row signs are LAPACK's choice
        out['perm_H'] = np.asarray(reml['H_sqrt_inv'], dtype=np.float64)
        out['perm_min_ps'] = np.asarray(pr['min_ps'], dtype=np.float64).reshape(-1)
        out['perm_max_f_stats'] = np.asarray(pr['max_f_stats'], dtype=np.float64).reshape(-1)
    return out


def run_extras(mods_by_mode):
    """Extra known answers (kept in their own file so the five case files stay bit-identical): replicated
    measurements through Z (linear_models.py:1796,1296-1297), exact EMMA per SNP (emma(), :1725-1745), get_ML
    (:672-683) and a multi-phenotype run = the reference's loop of emmax() calls over phenotypes that share the
    genotypes and the kinship (hdf5_data.py:262-330 / phenotypeData.py:70-78 shape), without and with a cofactor."""
    rng = np.random.RandomState(11)
    n, m = 150, 700
    snps = structured_genotypes(rng, n, m)
    data = {'snps_packed': np.packbits(snps.astype(np.uint8), axis=1), 'n': np.int64(n)}
    # --- multi-phenotype: different heritabilities -> different delta per phenotype
    h2s = [0.2, 0.5, 0.8, 0.35, 0.65, 0.9]
    ys = np.asarray([phenotype(rng, snps, h2=h, ncausal=5 + 2 * i) for i, h in enumerate(h2s)])
    cof = rng.randn(n) + 0.4 * snps[11]
    data['multi_ys'] = ys
    data['multi_cof'] = cof
    # --- replicates: 190 measurements of 150 accessions, ecotypes consecutive (get_incidence_matrix assumes sorted)
    reps = np.sort(np.concatenate([np.arange(n), rng.choice(n, 40, replace=False)]))
    Z = (reps[:, None] == np.arange(n)[None, :]).astype(np.int8)
    yz = Z @ ys[1] + 0.3 * rng.randn(len(reps))
    data['z_Z'] = Z
    data['z_y'] = yz
    for mode, mods in mods_by_mode.items():
        lm, kin = mods['linear_models'], mods['kinship']
        snp_list = list(snps)
        k_ibs = np.asarray(quiet(kin.calc_ibs_kinship, snp_list))
        if mode == 'dbl':
            data['ibs_scaled'] = k_ibs
        for tag, cf in (('multi', None), ('multic', [cof])):
            rows = {k: [] for k in ('ps', 'f_stats', 'rss', 'var_perc', 'h0_rss', 'pseudo_heritability', 'max_ll')}
            for y in ys:
                r = quiet(lm.emmax, snp_list, list(y), k_ibs, cofactors=cf)
                for k in ('ps', 'f_stats', 'rss', 'var_perc'):
                    rows[k].append(np.asarray(r[k], dtype=np.float64).reshape(-1))
                rows['h0_rss'].append(float(np.asarray(r['h0_rss']).reshape(-1)[0]))
                rows['pseudo_heritability'].append(float(r['pseudo_heritability']))
                rows['max_ll'].append(float(r['max_ll']))
            for k, v in rows.items():
                data['%s_%s_%s' % (mode, tag, k)] = np.asarray(v)
        rz = quiet(lm.emmax, snp_list, list(yz), k_ibs, Z=np.asmatrix(Z))
        for k in ('ps', 'f_stats', 'rss', 'var_perc'):
            data['%s_z_%s' % (mode, k)] = np.asarray(rz[k], dtype=np.float64).reshape(-1)
        data['%s_z_h0_rss' % mode] = np.asarray(rz['h0_rss'], dtype=np.float64).reshape(-1)
        data['%s_z_pseudo_heritability' % mode] = np.float64(rz['pseudo_heritability'])
        re_ = quiet(lm.emma, [sn for sn in snps[:12]], list(ys[2]), k_ibs)
        for k in ('ps', 'f_stats', 'vgs', 'ves', 'var_perc', 'max_lls', 'rss'):
            data['%s_emma_%s' % (mode, k)] = np.asarray(re_[k], dtype=np.float64).reshape(-1)
        data['%s_emma_betas' % mode] = np.asarray(re_['betas'], dtype=np.float64)
        lmm = lm.LinearMixedModel(list(ys[2]))
        lmm.add_random_effect(k_ibs)
        ml = quiet(lmm.get_ML)
        for k in ('max_ll', 'delta', 've', 'vg', 'pseudo_heritability'):
            data['%s_ml_%s' % (mode, k)] = np.float64(ml[k])
        data['%s_ml_beta' % mode] = np.asarray(ml['beta'], dtype=np.float64).reshape(-1)
        rl = quiet(lmm.get_REML)
        data['%s_reml100_delta' % mode] = np.float64(rl['delta'])
        data['%s_reml100_max_ll' % mode] = np.float64(rl['max_ll'])
    path = os.path.join(HERE, 'extras_n150.npz')
    np.savez_compressed(path, **data)
    print('extras_n150', snps.shape, '%.0f KB' % (os.path.getsize(path) / 1024.0))


def run_extras2(mods_by_mode):
    """Round-3 known answers, in their own file (the earlier fixture files stay bit-identical):
    * t_snps of _emmax_f_test_(return_transformed_snps=True) (linear_models.py:1309-1321,1355-1356), with the
      H_sqrt_inv the reference was handed (an argument, :1272; its row signs are LAPACK's choice);
    * the PUBLIC permutation test LinearMixedModel.emmax_permutations (:1180-1230, the worker of emmax_perm_test
      :1819-1841, whose own tail `p_f_list[len(p_f_list) / 20]` is a float index under Python 3) with the recorded
      shuffles and the H_sqrt_inv it computed; num_snps <= num_perm, where its per-SNP indexing (:1213) runs."""
    rng = np.random.RandomState(23)
    n, m = 150, 500
    snps = structured_genotypes(rng, n, m)
    y = phenotype(rng, snps, h2=0.55, ncausal=8)
    cof = rng.randn(n) + 0.5 * snps[5]
    nperm, ksnps = 24, 16
    data = {'snps_packed': np.packbits(snps.astype(np.uint8), axis=1), 'n': np.int64(n), 'y': y, 'cof': cof,
            'perm_num_snps': np.int64(ksnps)}
    for mode, mods in mods_by_mode.items():
        lm, kin = mods['linear_models'], mods['kinship']
        snp_list = list(snps)
        k_ibs = np.asarray(quiet(kin.calc_ibs_kinship, snp_list))
        if mode == 'dbl':
            data['ibs_scaled'] = k_ibs
        for tag, cf in (('t', None), ('tc', [cof])):
            lmm = lm.LinearMixedModel(list(y))
            lmm.add_random_effect(k_ibs)
            if cf is not None:
                for c in cf:
                    lmm.add_factor(c)
            eig_L = lmm._get_eigen_L_()
            est = quiet(lmm.get_estimates, eig_L, method='REML')
            H = np.asarray(est['H_sqrt_inv'], dtype=np.float64)
            r = quiet(lmm._emmax_f_test_, snp_list[:64], est['H_sqrt_inv'], return_transformed_snps=True, emma_num=0)
            data['%s_%s_H' % (mode, tag)] = H if mode == 'dbl' else H.astype(np.float32)
            data['%s_%s_snps' % (mode, tag)] = np.asarray(r['t_snps'], dtype=np.float64)
            data['%s_%s_ps' % (mode, tag)] = np.asarray(r['ps'], dtype=np.float64).reshape(-1)
        # public permutation test: replay the shuffles, then run the reference with the same RNG state
        lmm = lm.LinearMixedModel(list(y))
        lmm.add_random_effect(k_ibs)
        eig_L = lmm._get_eigen_L_(lmm.random_effects[1][1])
        H = np.asarray(quiet(lmm.get_estimates, eig_L=eig_L, method='REML')['H_sqrt_inv'], dtype=np.float64)
        np.random.seed(4242)
        idx = np.asmatrix(np.arange(n).reshape(n, 1))
        perm_idx = []
        for _ in range(nperm):
            np.random.shuffle(idx)
            perm_idx.append(np.asarray(idx).reshape(-1).copy())
        np.random.seed(4242)
        pr = quiet(lmm.emmax_permutations, [s.astype(np.float64) for s in snps[:ksnps]], nperm)
        data['%s_pub_perm_idx' % mode] = np.asarray(perm_idx, dtype=np.int32)
        data['%s_pub_perm_H' % mode] = H if mode == 'dbl' else H.astype(np.float32)
        data['%s_pub_min_ps' % mode] = np.asarray(pr['min_ps'], dtype=np.float64).reshape(-1)
        data['%s_pub_max_f_stats' % mode] = np.asarray(pr['max_f_stats'], dtype=np.float64).reshape(-1)
    path = os.path.join(HERE, 'extras2_n150.npz')
    np.savez_compressed(path, **data)
    print('extras2_n150', snps.shape, '%.0f KB' % (os.path.getsize(path) / 1024.0))


def run_extras3(mods_by_mode):
    """Round-4 known answers (own file: the earlier fixtures stay bit-identical):
    * LinearModel.fast_f_test(with_betas=True) (linear_models.py:196-257: the residual regressed on [X, s] per SNP), one
      cofactor, the last SNP monomorphic (rank-deficient design: keeps h0_betas, :236-239);
    * _emmax_f_test_(with_betas=True, return_transformed_snps=True): under with_betas the reference's M is H' itself
      (:1305-1306), so t_snps are the unprojected rotated SNPs; `betas` per SNP (:1323-1326);
    * the reference's loop of emmax() runs over three phenotypes with FOUR cofactors (q = 5 fixed-effect columns)."""
    rng = np.random.RandomState(41)
    n, m = 150, 400
    snps = structured_genotypes(rng, n, m)
    y = phenotype(rng, snps, h2=0.5, ncausal=6)
    cofs = [rng.randn(n) + 0.4 * snps[3 + i] for i in range(4)]
    ys = np.array([y, phenotype(rng, snps, h2=0.7, ncausal=9), phenotype(rng, snps, h2=0.3, ncausal=5)])
    data = {'snps_packed': np.packbits(snps.astype(np.uint8), axis=1), 'n': np.int64(n), 'y': y, 'ys': ys,
            'cofs': np.asarray(cofs)}
    sub = [s for s in snps[:79]] + [np.ones(n, dtype=np.int8)]
    for mode, mods in mods_by_mode.items():
        lm, kin = mods['linear_models'], mods['kinship']
        lin = lm.LinearModel(list(y))
        lin.add_factor(cofs[0])
        r = quiet(lin.fast_f_test, sub, with_betas=True)
        for k in ('ps', 'f_stats', 'rss', 'var_perc'):
            data['%s_lmwb_%s' % (mode, k)] = np.asarray(r[k], dtype=np.float64).reshape(-1)
        data['%s_lmwb_h0_rss' % mode] = np.asarray(r['h0_rss'], dtype=np.float64).reshape(-1)
        data['%s_lmwb_h0_betas' % mode] = np.asarray(list(r['h0_betas']), dtype=np.float64)
        data['%s_lmwb_betas' % mode] = np.asarray([list(b) for b in r['betas']], dtype=np.float64)
        k_ibs = np.asarray(quiet(kin.calc_ibs_kinship, list(snps)))
        if mode == 'dbl':
            data['ibs_scaled'] = k_ibs
        lmm = lm.LinearMixedModel(list(y))
        lmm.add_random_effect(k_ibs)
        lmm.add_factor(cofs[0])
        eig_L = lmm._get_eigen_L_()
        est = quiet(lmm.get_estimates, eig_L, method='REML')
        H = np.asarray(est['H_sqrt_inv'], dtype=np.float64)
        r = quiet(lmm._emmax_f_test_, list(snps[:64]), est['H_sqrt_inv'], return_transformed_snps=True, with_betas=True,
                  emma_num=0)
        data['%s_twb_H' % mode] = H if mode == 'dbl' else H.astype(np.float32)
        data['%s_twb_snps' % mode] = np.asarray(r['t_snps'], dtype=np.float64)
        data['%s_twb_ps' % mode] = np.asarray(r['ps'], dtype=np.float64).reshape(-1)
        data['%s_twb_betas' % mode] = np.asarray([list(b) for b in r['betas']], dtype=np.float64)
        ps, deltas = [], []
        for yy in ys:
            rr = quiet(lm.emmax, list(snps), list(yy), k_ibs, cofactors=[list(c) for c in cofs])
            ps.append(np.asarray(rr['ps'], dtype=np.float64).reshape(-1))
            deltas.append(1.0 / float(rr['pseudo_heritability']) - 1.0)
        data['%s_mc4_ps' % mode] = np.asarray(ps)
        data['%s_mc4_delta' % mode] = np.asarray(deltas)
    path = os.path.join(HERE, 'extras3_n150.npz')
    np.savez_compressed(path, **data)
    print('extras3_n150', snps.shape, '%.0f KB' % (os.path.getsize(path) / 1024.0))


def hdf5_genotypes(rng, n, m, diploid):
    """One chromosome of the HDF5 layout (plink2hdf5.py:111-118): int8 raw_snps [m x n], `freqs` as the parser writes
    them for 0/1/2 codes (mean / 2, plink2hdf5.py:202) or the carrier frequency for 0/1 codes, sorted positions.
    Allele frequencies span 0.02..0.98 so that `mafs > min_maf` (hdf5_data.py:91-93) removes a good share, plus two rows
    ON the threshold of min_maf = 0.1 at n = 200 (freq 0.1 exactly, and 0.9 whose 1 - f is 0.0999...98)."""
    npop, fst = 3, 0.12
    anc = rng.uniform(0.02, 0.98, size=m)
    a = anc * (1 - fst) / fst
    b = (1 - anc) * (1 - fst) / fst
    pop_p = rng.beta(a[:, None], b[:, None], size=(m, npop))
    pops = rng.randint(0, npop, size=n)
    snps = (rng.random_sample((m, n)) < pop_p[:, pops]).astype(np.int8)
    if diploid:
        snps = snps + (rng.random_sample((m, n)) < pop_p[:, pops]).astype(np.int8)
    top = 2 if diploid else 1
    for row, carriers in ((5, n // 10), (6, n - n // 10)):         # carrier frequency 0.1 and 0.9 on the binary file
        snps[row] = 0
        snps[row, rng.choice(n, carriers, replace=False)] = top
    keep = snps.std(1) > 0                                            # monomorphic rows cannot be standardised (:103)
    snps = snps[keep]
    freqs = snps.mean(1) / float(top)
    positions = np.sort(rng.choice(10 ** 6, len(snps), replace=False)).astype(np.int64)
    return snps, freqs, positions


def run_hdf5(mods_by_mode):
    """Reference-run answers for the HDF5 DRIVERS (hdf5_data.py: calculate_ibd_kinship :17-62, run_emmax :70-187,
    run_emmax_perm :191-351) -- needs a real h5py, i.e. `/opt/conda/bin/python3.9 tests/golden/make_golden.py` with
    MMG_GOLDEN_ONLY=hdf5 in the build container.  Two 3-chromosome files in plink2hdf5's layout (0/1 codes and 0/1/2
    codes), N = 200; the fixture keeps the INPUT arrays and what the reference wrote to its files, nothing else
    (the tests rebuild the container from the arrays through chunkstore)."""
    import tempfile
    import h5py
    n, nperm, chunk, min_maf = 200, 40, 150, 0.1
    data = {'n': np.int64(n), 'num_perm': np.int64(nperm), 'chunk_size': np.int64(chunk), 'min_maf': np.float64(min_maf)}
    for variant, seed in (('bin', 61), ('dip', 62)):
        rng = np.random.RandomState(seed)
        chroms = [hdf5_genotypes(rng, n, m, variant == 'dip') for m in (420, 380, 400)]
        allsnps = np.vstack([c[0] for c in chroms])
        y = phenotype(rng, allsnps, h2=0.6, ncausal=8)
        ids = np.arange(1000, 1000 + n)
        for ci, (snps, freqs, positions) in enumerate(chroms):
            name = 'chrom_%d' % (ci + 1)
            # 0/1 codes bit-packed, 0/1/2 codes as two bit planes ([s >= 1], [s == 2]) -- the tests unpack to int8
            data['%s_%s_raw_snps_ge1' % (variant, name)] = np.packbits((snps >= 1).astype(np.uint8), axis=1)
            if variant == 'dip':
                data['%s_%s_raw_snps_eq2' % (variant, name)] = np.packbits((snps == 2).astype(np.uint8), axis=1)
            data['%s_%s_freqs' % (variant, name)] = freqs
            data['%s_%s_positions' % (variant, name)] = positions
        data['%s_phenotypes' % variant] = y
        data['%s_indiv_ids' % variant] = ids
        for mode, mods in mods_by_mode.items():
            hd = mods['hdf5_data']
            tmp = tempfile.mkdtemp(prefix='mmg_h5_')
            try:
                fn = os.path.join(tmp, 'geno.hdf5')
                f = h5py.File(fn, 'w')
                gg, ig = f.create_group('genot_data'), f.create_group('indiv_data')
                for ci, (snps, freqs, positions) in enumerate(chroms):
                    cg = gg.create_group('chrom_%d' % (ci + 1))
                    cg.create_dataset('raw_snps', compression='lzf', data=snps)
                    cg.create_dataset('positions', compression='lzf', data=positions)
                    cg.create_dataset('freqs', compression='lzf', data=freqs)
                ig.create_dataset('indiv_ids', data=ids)
                ig.create_dataset('phenotypes', data=y)
                f.create_dataset('num_snps', data=np.array(len(allsnps)))
                f.close()
                tag = '%s_%s' % (variant, mode)
                iu = np.triu_indices(n)

                def keep32(v):                                          # exactly symmetric (asserted): upper triangle
                    assert np.array_equal(v, v.T)
                    return v[iu].astype(np.float32) if mode == 'lit' else v[iu]
                # --- run_emmax, kinship recalculated from the MAF-filtered SNPs (:82-111)
                out1 = os.path.join(tmp, 'res.hdf5')
                quiet(hd.run_emmax, hdf5_filename=fn, out_file=out1, min_maf=min_maf, recalculate_kinship=True,
                      chunk_size=chunk)
                with h5py.File(out1, 'r') as o:
                    for k in ('pseudo_heritability', 've', 'vg', 'max_ll', 'num_snps'):
                        data['%s_emmax_%s' % (tag, k)] = np.asarray(o[k][...], dtype=np.float64)
                    data['%s_emmax_chroms' % tag] = np.asarray(list(o['chrom_results'].keys()))
                    for c in o['chrom_results'].keys():
                        data['%s_emmax_%s_ps' % (tag, c)] = np.asarray(o['chrom_results'][c]['ps'][...], dtype=np.float64)
                        data['%s_emmax_%s_positions' % (tag, c)] = np.asarray(o['chrom_results'][c]['positions'][...])
                # --- run_emmax_perm with the shuffles recorded (sp.random.shuffle of an n x 1 matrix, linear_models.py:1153)
                np.random.seed(777)
                idx = np.asmatrix(np.arange(n).reshape(n, 1))
                perm_idx = []
                for _ in range(nperm):
                    np.random.shuffle(idx)
                    perm_idx.append(np.asarray(idx).reshape(-1).copy())
                data['%s_perm_idx' % tag] = np.asarray(perm_idx, dtype=np.int32)
                out2 = os.path.join(tmp, 'res_perm.hdf5')
                # the outcome of a permutation depends on the SIGNS of H_sqrt_inv's rows (the rotated residuals are what is
                # shuffled, linear_models.py:1151-1154) and those are LAPACK's choice: observe the matrix the driver hands
                # to _emmax_permutations_ (:330) and keep, per row, where its largest entry sits and that entry's sign,
                # plus three products H v to verify a reconstruction against
                seen = {}
                lmm_cls = mods['linear_models'].LinearMixedModel
                orig = lmm_cls._emmax_permutations_

                def spy(self, snps_, K_, H_, num_perm=100, _orig=orig, _seen=seen):
                    _seen['H'] = np.array(H_, dtype=np.float64)
                    _seen['num_rows'] = len(snps_)
                    return _orig(self, snps_, K_, H_, num_perm=num_perm)
                lmm_cls._emmax_permutations_ = spy
                np.random.seed(777)
                try:
                    quiet(hd.run_emmax_perm, hdf5_filename=fn, out_file=out2, min_maf=min_maf, chunk_size=chunk,
                          num_perm=nperm)
                finally:
                    lmm_cls._emmax_permutations_ = orig
                H = seen['H']
                jmax = np.abs(H).argmax(axis=1)
                data['%s_perm_H_argmax' % tag] = jmax.astype(np.int32)
                data['%s_perm_H_sign' % tag] = np.sign(H[np.arange(n), jmax]).astype(np.int8)
                data['%s_perm_H_probe' % tag] = H @ np.random.RandomState(99).randn(n, 3)
                data['%s_perm_num_rows' % tag] = np.int64(seen['num_rows'])
                with h5py.File(out2, 'r') as o:
                    for k in ('pseudo_heritability', 've', 'vg', 'max_ll', 'num_snps', 'perm_min_ps', 'perm_max_f_stats',
                              'five_perc_perm_min_ps', 'five_perc_perm_max_f_stats'):
                        data['%s_perm_%s' % (tag, k)] = np.asarray(o[k][...], dtype=np.float64)
                    data['%s_perm_kinship' % tag] = keep32(np.asarray(o['kinship'][...]))
                    for c in o['chrom_results'].keys():
                        data['%s_perm_%s_ps' % (tag, c)] = np.asarray(o['chrom_results'][c]['ps'][...], dtype=np.float64)
                # --- calculate_ibd_kinship: ALL SNPs (no MAF filter, :17-62), stored into the genotype file (:60)
                quiet(hd.calculate_ibd_kinship, hdf5_filename=fn, chunk_size=chunk)
                with h5py.File(fn, 'r') as o:
                    data['%s_calc_kinship' % tag] = keep32(np.asarray(o['kinship'][...]))
                # ... a second call finds it there and leaves it (:25,61-62); overwrite=True recomputes the same
                quiet(hd.calculate_ibd_kinship, hdf5_filename=fn, chunk_size=400, overwrite=True)
                with h5py.File(fn, 'r') as o:                           # (only how far a different chunking moves it is kept)
                    data['%s_calc_kinship_chunk400_maxdiff' % tag] = np.float64(np.abs(
                        keep32(np.asarray(o['kinship'][...])) - data['%s_calc_kinship' % tag]).max())
                # --- calculate_ibd_kinship on a file that carries pre-normalised `snps` datasets (:36-38: taken as they are, no
                # standardisation; the loop bound is still len(raw_snps)) -- the 0/1 file's rows standardised in float64
                if variant == 'bin':
                    fn2 = os.path.join(tmp, 'geno_norm.hdf5')
                    f2 = h5py.File(fn2, 'w')
                    gg2, ig2 = f2.create_group('genot_data'), f2.create_group('indiv_data')
                    for ci, (snps, freqs, positions) in enumerate(chroms):
                        cg2 = gg2.create_group('chrom_%d' % (ci + 1))
                        x = snps.astype(np.float64)
                        cg2.create_dataset('raw_snps', compression='lzf', data=snps)
                        cg2.create_dataset('snps', data=(x - x.mean(1, keepdims=True)) / x.std(1, keepdims=True))
                        cg2.create_dataset('positions', data=positions)
                        cg2.create_dataset('freqs', data=freqs)
                    ig2.create_dataset('indiv_ids', data=ids)
                    ig2.create_dataset('phenotypes', data=y)
                    f2.close()
                    quiet(hd.calculate_ibd_kinship, hdf5_filename=fn2, chunk_size=chunk)
                    with h5py.File(fn2, 'r') as o:
                        data['%s_calcnorm_kinship' % tag] = keep32(np.asarray(o['kinship'][...]))
                # --- run_emmax(recalculate_kinship=False) reads that stored (unfiltered) kinship (:113-115)
                out3 = os.path.join(tmp, 'res_k.hdf5')
                try:
                    quiet(hd.run_emmax, hdf5_filename=fn, out_file=out3, min_maf=min_maf, recalculate_kinship=False,
                          chunk_size=chunk)
                    with h5py.File(out3, 'r') as o:
                        data['%s_storedk_pseudo_heritability' % tag] = np.asarray(o['pseudo_heritability'][...], dtype=np.float64)
                        for c in o['chrom_results'].keys():
                            data['%s_storedk_%s_ps' % (tag, c)] = np.asarray(o['chrom_results'][c]['ps'][...], dtype=np.float64)
                except Exception as e:                                  # recorded, not hidden: the tests assert on it
                    data['%s_storedk_error' % tag] = np.asarray('%s: %s' % (type(e).__name__, e))
            finally:
                import shutil
                shutil.rmtree(tmp, ignore_errors=True)
    path = os.path.join(HERE, 'hdf5_n200.npz')
    np.savez_compressed(path, **data)
    print('hdf5_n200', '%.0f KB' % (os.path.getsize(path) / 1024.0))
    for k in sorted(data):
        if 'error' in k:
            print(' ', k, data[k])


def _row_signs(H):
    """Per row of an H_sqrt_inv: where its largest entry sits and that entry's sign (LAPACK's choice), plus three products
    H v -- enough to rebuild the matrix from the same K and delta elsewhere (conftest.reference_row_signs) at 1/1000 of its size."""
    n = len(H)
    jmax = np.abs(H).argmax(axis=1)
    return (jmax.astype(np.int32), np.sign(H[np.arange(n), jmax]).astype(np.int8),
            H @ np.random.RandomState(99).randn(n, 3))


def run_n1000(mods_by_mode):
    """Round-6 known answers at a size that reaches the device code's real tile structure (N = 1000 -> Npad 1024: four tile
    rows of the triangular scan GEMM, its tail launch, >= 15 panels of the band reduction): structured N = 1000 x M = 4000,
    two cofactors for emmax / REML / exact-EMMA refinement of the 10 top hits (emma_num=10), and 50 recorded permutations
    of the intercept-only model (_emmax_permutations_ needs q == 1).  The 8 MB matrices are NOT stored: K is an exact
    function of the genotypes (integer counts / 2M + 0.5, scale_k) -- the fixture carries products K v to verify a rebuilt K
    against -- and H_sqrt_inv of the permutation run is kept as its row signs (_row_signs)."""
    rng = np.random.RandomState(7)
    n, m, nperm = 1000, 4000, 50
    snps = structured_genotypes(rng, n, m, npop=4, fst=0.08)
    y = phenotype(rng, snps, h2=0.6, ncausal=12)
    cof = [rng.randn(n) + 0.5 * snps[7 + i] for i in range(2)]
    data = {'snps_packed': np.packbits(snps.astype(np.uint8), axis=1), 'n': np.int64(n), 'y': y,
            'cofactors': np.asarray(cof)}
    np.random.seed(1007)
    idx = np.asmatrix(np.arange(n).reshape(n, 1))
    perm_idx = []
    for _ in range(nperm):
        np.random.shuffle(idx)
        perm_idx.append(np.asarray(idx).reshape(-1).copy())
    data['perm_idx'] = np.asarray(perm_idx, dtype=np.int16)
    for mode, mods in mods_by_mode.items():
        lm, kin = mods['linear_models'], mods['kinship']
        f = (lambda v: np.asarray(v, dtype=np.float64)) if mode == 'dbl' else (lambda v: np.asarray(v, dtype=np.float32))
        snp_list = list(snps)
        k_ibs = np.asarray(quiet(kin.calc_ibs_kinship, snp_list))
        if mode == 'dbl':
            data['ibs_scaled_probe'] = k_ibs @ np.random.RandomState(98).randn(n, 3)
            data['ibs_scaled_diag'] = np.diag(k_ibs).copy()
        res = quiet(lm.emmax, snp_list, list(y), k_ibs, cofactors=cof)
        for k in ('ps', 'f_stats', 'rss', 'var_perc'):
            data['%s_emmax_%s' % (mode, k)] = f(res[k]).reshape(-1)
        data['%s_emmax_h0_rss' % mode] = np.asarray(res['h0_rss'], dtype=np.float64).reshape(-1)
        data['%s_emmax_h0_betas' % mode] = np.asarray(res['h0_betas'], dtype=np.float64).reshape(-1)
        for k in ('pseudo_heritability', 've', 'vg', 'max_ll'):
            data['%s_emmax_%s' % (mode, k)] = np.float64(res[k])
        reml = quiet(lm.get_emma_reml_estimates, list(y), k_ibs, cofactors=cof)
        for k in ('max_ll', 'delta', 've', 'vg', 'pseudo_heritability'):
            data['%s_reml_%s' % (mode, k)] = np.float64(reml[k])
        data['%s_reml_beta' % mode] = np.asarray(reml['beta'], dtype=np.float64).reshape(-1)
        data['%s_reml_mahalanobis_rss' % mode] = np.asarray(reml['mahalanobis_rss'], dtype=np.float64).reshape(-1)
        data['%s_eig_L_values' % mode] = f(reml['eig_L']['values'])
        lmm = reml['lmm']
        data['%s_eig_R_values' % mode] = f(lmm._get_eigen_R_(X=lmm.X)['values'])
        H = np.asarray(reml['H_sqrt_inv'], dtype=np.float64)
        data['%s_HtH_probe' % mode] = H.T @ (H @ np.random.RandomState(99).randn(n, 3))
        if mode == 'dbl':
            eres = quiet(lm.emmax, snp_list, list(y), k_ibs, cofactors=cof, emma_num=10)
            for k in ('ps', 'f_stats', 'rss', 'var_perc'):
                data['%s_emma10_%s' % (mode, k)] = f(eres[k]).reshape(-1)
        if mode == 'dbl':
            # MLMM forward / backward at this size (linear_models.py:2543-2923): every step a full scan with one more cofactor
            import sys as _sys
            _sys.modules['gwaResults'] = mods['gwaResults']
            lm.agr.calc_median = lambda ps, exp_median=0.5: float(np.median(ps) - exp_median)   # py2 int division inside
            lm.agr.calc_ks_stats = lambda ps, exp_dist=None: {'D': 0.0, 'p_val': 1.0}
            mres = quiet(lm.mlmm, list(y), k_ibs, num_steps=3, forward_backwards=True, file_prefix=None, snps=list(snp_list),
                         positions=list(range(m)), chromosomes=[1] * m, mafs=[0.3] * m, macs=[30] * m)
            keys = ('pseudo_heritability', 'll', 'bic', 'e_bic', 'm_bic', 'mbonf', 'min_pval', 'rss', 'reml_mahalanobis_rss',
                    'mahalanobis_rss')
            data['dbl_mlmm_steps'] = np.asarray([[np.nan if si[k] is None else float(np.asarray(si[k]).reshape(-1)[0]) for k in keys]
                                                 for si in mres['step_info_list']])
            data['dbl_mlmm_cof_pos'] = np.asarray([[c[1] for c in si['cofactors']] + [-1] * (3 - len(si['cofactors']))
                                                   for si in mres['step_info_list']])
            data['dbl_mlmm_cof_mlogp'] = np.asarray([[c[2] for c in si['cofactors']] + [np.nan] * (3 - len(si['cofactors']))
                                                     for si in mres['step_info_list']])
            for c in ('ebics', 'mbics', 'bonf', 'mbonf', 'min_cof_ppa'):
                data['dbl_mlmm_opt_' + c] = np.int64(mres['opt_dict'][c])
        # permutations: the intercept-only model
        lmm3 = lm.LinearMixedModel(list(y))
        lmm3.add_random_effect(k_ibs)
        reml0 = quiet(lm.get_emma_reml_estimates, list(y), k_ibs)
        data['%s_reml0_delta' % mode] = np.float64(reml0['delta'])
        H0 = np.asarray(reml0['H_sqrt_inv'], dtype=np.float64)
        (data['%s_perm_H_argmax' % mode], data['%s_perm_H_sign' % mode], data['%s_perm_H_probe' % mode]) = _row_signs(H0)
        np.random.seed(1007)
        pr = quiet(lmm3._emmax_permutations_, [s_.astype(np.float64) for s_ in snps], k_ibs, reml0['H_sqrt_inv'],
                   num_perm=nperm)
        data['%s_perm_min_ps' % mode] = np.asarray(pr['min_ps'], dtype=np.float64).reshape(-1)
        data['%s_perm_max_f_stats' % mode] = np.asarray(pr['max_f_stats'], dtype=np.float64).reshape(-1)
    path = os.path.join(HERE, 'struct_n1000_s7.npz')
    np.savez_compressed(path, **data)
    print('struct_n1000_s7', snps.shape, '%.0f KB' % (os.path.getsize(path) / 1024.0))


def run_config1(mods_by_mode):
    """Round-6 known answers in the shape of BASELINE config 1 (examples.py:72-96): the FT10 phenotype (phenotype_id 5) of
    at_data/199_phenotypes.csv -- its 198 accessions and values, the rows tests/golden/at_phenotypes_ft10_ft16.csv excerpts --
    against 3,000 synthetic SNPs on 5 chromosomes (the real genotype file is not mounted) for 210 genotyped accessions in a
    shuffled order, 190 of them phenotyped, so that the reference's own coordinate_w_phenotype_data has accessions to drop on
    both sides, an order to impose and non-binary SNPs to remove; then calc_ibs_kinship(sd.get_snps()) and
    emmax(sd.get_snps(), phend.get_values(5), K), verbatim.  The sub-population of an accession follows its FT10 value
    (flowering time is structured in the real data too), so the REML optimum is interior."""
    import csv
    rows = [r for r in csv.reader(open(os.path.join(refshim.REF, 'at_data', '199_phenotypes.csv'))) if r and r[0] == '5']
    mine = [r for r in csv.reader(open(os.path.join(HERE, 'at_phenotypes_ft10_ft16.csv'))) if r and r[0] == '5']
    assert [(r[2], float(r[3])) for r in rows] == [(r[2], float(r[3])) for r in mine], 'the committed excerpt is not FT10'
    ets = [r[2] for r in rows]
    vals = np.array([float(r[3]) for r in rows])
    rng = np.random.RandomState(31)
    order = np.argsort(np.argsort(vals + 25.0 * rng.randn(len(vals))))           # noisy rank of the FT10 value (sd 17.5)
    pop_of = {e: int(3 * o // len(ets)) for e, o in zip(ets, order)}
    genotyped = [ets[i] for i in rng.choice(len(ets), 190, replace=False)] + [str(990000 + i) for i in range(20)]
    genotyped = [genotyped[i] for i in rng.permutation(len(genotyped))]
    pops = np.array([pop_of.get(a, rng.randint(3)) for a in genotyped])
    m, fst = 3000, 0.15
    anc = rng.uniform(0.1, 0.9, size=m)
    pop_p = rng.beta((anc * (1 - fst) / fst)[:, None], ((1 - anc) * (1 - fst) / fst)[:, None], size=(m, 3))
    snps = (rng.random_sample((m, len(genotyped))) < pop_p[:, pops]).astype(np.int8)
    unphen = np.array([a not in pop_of for a in genotyped])
    for row in (11, 700, 1500, 2999):                                             # carried only by accessions that get dropped
        snps[row] = 0
        snps[row, np.nonzero(unphen)[0][:3]] = 1
    chromosomes = np.repeat(np.arange(1, 6), m // 5)
    positions = np.concatenate([np.sort(rng.choice(10 ** 7, m // 5, replace=False)) for _ in range(5)]).astype(np.int64)
    data = {'snps_packed': np.packbits(snps.astype(np.uint8), axis=1), 'n_genotyped': np.int64(len(genotyped)),
            'accessions': np.asarray(genotyped), 'chromosomes': chromosomes.astype(np.int8), 'positions': positions,
            'phenotype_id': np.int64(5)}
    for mode, mods in mods_by_mode.items():
        sd_mod, pd_mod, kin, lm = mods['snpsdata'], mods['phenotypeData'], mods['kinship'], mods['linear_models']
        f = (lambda v: np.asarray(v, dtype=np.float64)) if mode == 'dbl' else (lambda v: np.asarray(v, dtype=np.float32))
        phend = quiet(pd_mod.parse_phenotype_file, os.path.join(refshim.REF, 'at_data', '199_phenotypes.csv'))
        # construct_snps_data_set's own body (snpsdata.py:3307-3323) minus its dtype assert, with the rows as the LIST the
        # parsers hand it (removeAccessionIndices :1229 assigns shorter rows in place, which a 2-D array refuses)
        snpsds = [sd_mod.SNPsData(snps=[snps[i] for i in np.nonzero(chromosomes == c)[0]],
                                  positions=list(positions[chromosomes == c]), chromosome=c, accessions=list(genotyped))
                  for c in range(1, 6)]
        sd = sd_mod.SNPsDataSet(snpsds, [1, 2, 3, 4, 5], data_format='binary')
        quiet(sd.coordinate_w_phenotype_data, phend, 5)
        s = sd.get_snps()
        values = phend.get_values(5)
        k_ibs = quiet(kin.calc_ibs_kinship, s)
        res = quiet(lm.emmax, s, values, k_ibs)
        if mode == 'dbl':
            data['coord_accessions'] = np.asarray(sd.accessions)
            data['coord_ecotypes'] = np.asarray(phend.phen_dict[5]['ecotypes'])
            data['coord_values'] = np.asarray(values, dtype=np.float64)
            data['coord_positions'] = np.asarray(sd.get_positions(), dtype=np.int64)
            data['coord_chromosomes'] = np.asarray(sd.get_chr_list(), dtype=np.int8)
            data['coord_snps_packed'] = np.packbits(np.asarray(s, dtype=np.uint8), axis=1)
            kk = np.asarray(k_ibs)
            assert np.array_equal(kk, kk.T)
            data['ibs_scaled_triu'] = kk[np.triu_indices(len(kk))]
        for k in ('ps', 'f_stats', 'rss', 'var_perc'):
            data['%s_emmax_%s' % (mode, k)] = f(res[k]).reshape(-1)
        data['%s_emmax_h0_rss' % mode] = np.asarray(res['h0_rss'], dtype=np.float64).reshape(-1)
        data['%s_emmax_h0_betas' % mode] = np.asarray(res['h0_betas'], dtype=np.float64).reshape(-1)
        for k in ('pseudo_heritability', 've', 'vg', 'max_ll'):
            data['%s_emmax_%s' % (mode, k)] = np.float64(res[k])
    path = os.path.join(HERE, 'ft10_config1.npz')
    np.savez_compressed(path, **data)
    print('ft10_config1', snps.shape, '->', len(s), 'x', len(s[0]), 'h2 %.3f' % data['dbl_emmax_pseudo_heritability'],
          '%.0f KB' % (os.path.getsize(path) / 1024.0))


CASES = [
    # name, kind, N, M, seed, n_cofactors, nperm
    ('struct_n150_s0', 'struct', 150, 600, 0, 0, 20),
    ('struct_n150_s1', 'struct', 150, 600, 1, 1, 0),
    ('struct_n300_s2', 'struct', 300, 3000, 2, 0, 25),
    ('struct_n300_s3', 'struct', 300, 3000, 3, 2, 0),
    ('bern_n200_s4', 'bern', 200, 1000, 4, 0, 10),
]


def main():
    if not refshim.available():
        print('reference not mounted; nothing to do')
        return
    if os.environ.get('MMG_GOLDEN_ONLY', '') == 'hdf5':           # under /opt/conda/bin/python3.9 (real h5py)
        run_hdf5({'lit': refshim.load('literal', with_hdf5=True), 'dbl': refshim.load('double', with_hdf5=True)})
        return
    mods = {'lit': refshim.load('literal'), 'dbl': refshim.load('double')}
    if os.environ.get('MMG_GOLDEN_ONLY', '') in ('', 'extras'):
        run_extras(mods)
    if os.environ.get('MMG_GOLDEN_ONLY', '') in ('', 'extras2'):
        run_extras2(mods)
    if os.environ.get('MMG_GOLDEN_ONLY', '') in ('', 'extras3'):
        run_extras3(mods)
    if os.environ.get('MMG_GOLDEN_ONLY', '') in ('', 'n1000'):
        run_n1000(mods)
    if os.environ.get('MMG_GOLDEN_ONLY', '') in ('', 'config1'):
        run_config1(mods)
    if os.environ.get('MMG_GOLDEN_ONLY', '') in ('extras', 'extras2', 'extras3', 'n1000', 'config1'):
        return
    for name, kind, n, m, seed, ncof, nperm in CASES:
        rng = np.random.RandomState(seed)
        snps = structured_genotypes(rng, n, m) if kind == 'struct' else bernoulli_genotypes(rng, n, m)
        y = phenotype(rng, snps)
        cof = [rng.randn(n) + 0.5 * snps[7 + i] for i in range(ncof)] if ncof else None
        data = {'snps_packed': np.packbits(snps.astype(np.uint8), axis=1), 'n': np.int64(n),
                'y': y, 'cofactors': np.asarray(cof) if cof is not None else np.zeros((0, n))}
        for mode, mm in mods.items():
            for k, v in run_case(mm, snps, y, cof, nperm, 1000 + seed).items():
                if mode == 'lit' and k.startswith(('ibs_', 'scale_k_', 'dip_')):
                    continue          # float64 in both modes and bit-identical (checked below)
                if k == 'ibs_unscaled':   # store the exact integer counts C = (K - 0.5) * 2M
                    c = (v - 0.5) * 2 * len(snps)
                    assert np.abs(c - np.rint(c)).max() < 1e-6
                    k, v = 'ibs_counts', np.rint(c).astype(np.int32)
                if mode == 'lit' and k in ('ibd_scaled', 'perm_H'):
                    v = v.astype(np.float32)
                data['%s_%s' % (mode, k)] = v
        # p-value known-answer grid (scipy.stats.f.sf as the reference calls it)
        path = os.path.join(HERE, name + '.npz')
        np.savez_compressed(path, **data)
        print(name, snps.shape, '%.0f KB' % (os.path.getsize(path) / 1024.0))
    from scipy import stats
    F = np.logspace(-8, np.log10(3e3), 400)
    kat = {'F': F}
    for nu in (197, 998, 4998, 49998):
        kat['sf_%d' % nu] = stats.f.sf(F, 1, nu)
    np.savez_compressed(os.path.join(HERE, 'f_sf_kat.npz'), **kat)


if __name__ == '__main__':
    main()
