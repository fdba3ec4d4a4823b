"""Result holder -- only what examples.py:93-96 uses of /root/reference/gwaResults.py:
Result(scores=, snps_data=) (:64) and write_to_file (:1864-1907).  Plotting is out of scope."""
import numpy as np


class Result(object):
    def __init__(self, scores=None, snps_data=None, positions=None, chromosomes=None):
        self.scores = np.asarray(scores, dtype=np.float64)
        self.positions = list(snps_data.get_positions() if snps_data is not None else positions)
        self.chromosomes = list(snps_data.get_chr_list() if snps_data is not None else chromosomes)
        assert len(self.scores) == len(self.positions) == len(self.chromosomes)

    def write_to_file(self, filename, only_pickled=False):
        with open(filename, 'w') as f:
            f.write('chromosomes,positions,scores\n')
            for c, p, s in zip(self.chromosomes, self.positions, self.scores):
                f.write('%s,%s,%r\n' % (c, p, float(s)))

    def min_score(self):
        return float(self.scores.min())
