"""Phenotype container -- the subset of /root/reference/phenotypeData.py the hot path's callers touch
(examples.py:25,84-90): parse_phenotype_file (:1032-1105), phenotype_data.get_values (:536),
get_ecotypes, filter_ecotypes (:451), get_incidence_matrix (:555-567)."""
import numpy as np


class phenotype_data(object):
    def __init__(self, phen_dict=None, phen_ids=None):
        self.phen_dict = phen_dict or {}
        self.phen_ids = list(phen_ids) if phen_ids is not None else list(self.phen_dict.keys())

    def get_values(self, pid):
        return self.phen_dict[pid]['values']

    def get_ecotypes(self, pid):
        return self.phen_dict[pid]['ecotypes']

    def get_name(self, pid):
        return self.phen_dict[pid]['name']

    def filter_ecotypes(self, indices_to_keep, pids=None):
        """:451-461 -- keep the given positions of every (or the given) phenotype."""
        for pid in (pids or self.phen_ids):
            d = self.phen_dict[pid]
            d['ecotypes'] = [d['ecotypes'][i] for i in indices_to_keep]
            d['values'] = [d['values'][i] for i in indices_to_keep]

    def get_incidence_matrix(self, pid):
        """:555-567 -- Z [n_values x n_unique_ecotypes] for replicated measurements."""
        ets = [str(e) for e in self.phen_dict[pid]['ecotypes']]
        uniq = [e for i, e in enumerate(ets) if i == 0 or ets[i - 1] != e]   # runs: "assumed to be sorted" (:561)
        return (np.asarray(ets)[:, None] == np.asarray(uniq)[None, :]).astype(np.int8)


def parse_phenotype_file(file_name=None, file_object=None, delim=',', file_format='guess', with_db_ids=True):
    """:1032-1105 -- 'new' long format (phenotype_id,phenotype_name,ecotype_id,value,replicate_id) or the
    'old' wide format (one column per phenotype, NA for missing)."""
    f = file_object if file_object else open(file_name)
    try:
        header = next(f)
        if len(header.split(delim)) < 2:
            for n_delim in (',', '\t'):
                if len(header.split(n_delim)) > 2:
                    delim = n_delim
                    break
            else:
                raise Exception('Problems with delimiters', delim)
        cols = [c.strip() for c in header.split(delim)]
        if file_format == 'guess':
            file_format = 'new' if ('phenotype_id' in cols or 'replicate_id' in cols) else 'old'
        phen_dict = {}
        if file_format == 'old':
            pids = [int(c.split('_')[0]) for c in cols[1:]] if with_db_ids else list(range(1, len(cols)))
            for i, pid in enumerate(pids):
                phen_dict[pid] = {'ecotypes': [], 'values': [], 'name': cols[i + 1]}
            for line in f:
                l = [c.strip() for c in line.split(delim)]
                for i, v in enumerate(l[1:]):
                    if v != 'NA':
                        phen_dict[pids[i]]['ecotypes'].append(l[0])
                        phen_dict[pids[i]]['values'].append(float(v))
        else:
            for line in f:
                l = line.split(delim)
                if len(l) < 4:
                    continue
                pid = int(l[0])
                d = phen_dict.setdefault(pid, {'name': l[1], 'ecotypes': [], 'values': []})
                d['ecotypes'].append(l[2])
                d['values'].append(float(l[3]))
    finally:
        if not file_object:
            f.close()
    return phenotype_data(phen_dict=phen_dict, phen_ids=list(phen_dict.keys()))
