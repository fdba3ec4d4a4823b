"""Genotype container -- the thin subset of /root/reference/snpsdata.py the EMMAX examples touch:
construct_snps_data_set (:3307), SNPsDataSet.get_snps (:2579: list of M int8 arrays of length N),
coordinate_w_phenotype_data (:2221-2297), get_positions / get_chr_list."""
import numpy as np


class SNPsDataSet(object):
    def __init__(self, snps, positions, chromosomes, accessions, data_format='binary'):
        self.snps = np.ascontiguousarray(snps, dtype=np.int8)        # [M x N], SNP-major
        assert self.snps.dtype == np.int8, "Type doesn't match the data format."   # :3320
        self.positions = list(positions)
        self.chromosomes = list(chromosomes)
        self.accessions = [str(a) for a in accessions]
        self.data_format = data_format

    def get_snps(self):
        return list(self.snps)                                        # :2579-2597

    getSnps = get_snps

    def get_positions(self):
        return self.positions

    def get_chr_list(self):
        return self.chromosomes

    def num_snps(self):
        return len(self.snps)

    def coordinate_w_phenotype_data(self, phend, pid, coord_phen=True):
        """:2221-2297 -- keep the accessions present in both (in this data set's order), drop SNPs that are
        no longer polymorphic, and filter the phenotype to the same accessions in the same order."""
        ets = [str(e) for e in phend.get_ecotypes(pid)]
        where = {}
        for j, e in enumerate(ets):
            where.setdefault(e, []).append(j)
        sd_keep = [i for i, a in enumerate(self.accessions) if a in where]
        # every matching phenotype entry, accession by accession in genotype order (:2236-2240): replicated
        # measurements survive, so get_incidence_matrix / emmax(Z=...) see them
        pd_keep = [j for i in sd_keep for j in where[self.accessions[i]]]
        self.snps = np.ascontiguousarray(self.snps[:, sd_keep])
        self.accessions = [self.accessions[i] for i in sd_keep]
        if coord_phen:
            phend.filter_ecotypes(pd_keep, pids=[pid])
        poly = (self.snps.min(axis=1) != self.snps.max(axis=1))        # :2281-2288
        self.snps = np.ascontiguousarray(self.snps[poly])
        self.positions = [p for p, k in zip(self.positions, poly) if k]
        self.chromosomes = [c for c, k in zip(self.chromosomes, poly) if k]
        return {'pd_indices_to_keep': pd_keep, 'n_filtered_snps': int((~poly).sum())}


def construct_snps_data_set(snps, positions, chromosomes, indiv_ids, data_format='binary'):
    """:3307-3323."""
    snps = np.asarray(snps)
    assert snps.dtype == np.dtype('int8'), "Type doesn't match the data format."
    return SNPsDataSet(snps, positions, chromosomes, indiv_ids, data_format=data_format)
