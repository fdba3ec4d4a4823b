#!/usr/bin/env python3
"""The call sequence of the reference's examples.py:65-102 (`mixed_model_gwas`) against this build.

The reference's genotype zip (at_data/at_genotypes.zip) is not part of the mounted reference
(.MISSING_LARGE_BLOBS), so the genotypes here are synthetic (simulations.simulate_genotypes restated)
for the accession ids of the phenotype file; the phenotype is the real FT10 trait (phenotype_id 5 of
at_data/199_phenotypes.csv, committed as tests/golden/at_phenotypes_ft10_ft16.csv).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def load_genotypes_for(accessions, num_snps=214000, seed=20240):
    from mixmogam_amd import simulations, snpsdata
    sim = simulations.simulate_genotypes(num_indivs=len(accessions), num_snps=num_snps, seed=seed)
    return snpsdata.construct_snps_data_set(sim['snps'], sim['positions'], sim['chromosomes'], accessions)


def mixed_model_gwas(phenotype_id=5, pvalue_file='mm_results.pvals', num_snps=214000,
                     phenotype_file=os.path.join(ROOT, 'tests', 'golden', 'at_phenotypes_ft10_ft16.csv')):
    from mixmogam_amd import linear_models as lm
    from mixmogam_amd import kinship
    from mixmogam_amd import gwaResults as gr
    from mixmogam_amd import phenotypeData as pd
    phend = pd.parse_phenotype_file(phenotype_file)                       # examples.py:25
    sd = load_genotypes_for(sorted(set(phend.get_ecotypes(phenotype_id))), num_snps=num_snps)   # :15
    sd.coordinate_w_phenotype_data(phend, phenotype_id)                   # :84
    K = kinship.calc_ibs_kinship(sd.get_snps())                           # :87
    mm_results = lm.emmax(sd.get_snps(), phend.get_values(phenotype_id), K)   # :90
    res = gr.Result(scores=mm_results['ps'], snps_data=sd)                # :93
    if pvalue_file:
        res.write_to_file(pvalue_file)                                    # :96
    return mm_results, sd, phend, K


if __name__ == '__main__':
    r, sd, phend, K = mixed_model_gwas()
    print('SNPs: %d  individuals: %d  min p: %.3e  pseudo-heritability: %.4f' %
          (len(r['ps']), len(K), float(np.min(r['ps'])), r['pseudo_heritability']))
