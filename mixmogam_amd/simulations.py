"""Synthetic genotypes / phenotypes -- restates /root/reference/simulations.py with explicit seeds.

simulate_genotypes follows simulations.py:17-33 (round(U(0,1)) -> int8 [M x N], all-zero rows
dropped); simulate_phenotype follows the single-trait part of simulate_traits (:64-85):
`num_causals` random causal SNPs, Exp(1) effects, noise scaled for heritability h2, z-scored.
The reference uses the unseeded global RNGs; here every draw comes from an explicit
numpy RandomState so that runs are reproducible.
"""
import numpy as np


def simulate_genotypes(num_indivs=1000, num_snps=1000, seed=20240):
    rng = np.random.RandomState(seed)
    snps = np.round(rng.random_sample((num_snps, num_indivs))).astype(np.int8)     # :21-23
    snps = snps[np.sum(snps, 1) > 0]                                               # :24
    n_per = len(snps) // 5
    chromosomes, positions = [], []
    for chrom in range(1, 6):                                                      # :27-30
        cnt = n_per if chrom < 5 else len(snps) - 4 * n_per
        chromosomes.extend([chrom] * cnt)
        positions.extend(range(1, cnt + 1))
    return {'snps': snps, 'positions': positions, 'chromosomes': chromosomes, 'indiv_ids': list(range(num_indivs))}


def simulate_phenotype(snps, h2=0.8, num_causals=100, seed=20241):
    """One trait of simulations.py:64-85 (h2 = 0.8, exponential effects, standardised)."""
    rng = np.random.RandomState(seed)
    snps = np.asarray(snps)
    m, n = snps.shape
    nc = min(num_causals, m)
    causal = rng.choice(m, nc, replace=False)
    chosen = snps[causal].astype(np.float64)
    flip = rng.randint(0, 2, size=(nc, 1))                                          # :70-71
    chosen = np.abs(flip - chosen)
    effects = rng.exponential(1.0, size=(nc, 1))                                    # :72
    trait = np.sum(chosen * effects, 0)
    gv = np.var(trait, ddof=1)
    error = rng.normal(0, 1, size=n)
    ev = np.var(error, ddof=1)
    y = trait + error * np.sqrt(((1.0 - h2) / h2) * (gv / ev))                      # :81
    return (y - np.mean(y)) / np.std(y)                                             # :83
