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


def synthetic_chunk(chunk_id, rows, num_indivs, seed=20240):
    """Rows [chunk_id * rows, ...) of a never-resident synthetic genotype matrix: i.i.d.
    Bernoulli(0.5) int8, one RandomState(seed + chunk_id) per chunk so that any chunk can be regenerated anywhere
    (by a rank, by a CPU check on a sample).  Bits come from RandomState.bytes + unpackbits -- the distribution of
    simulations.py:21 `round(U(0,1))` at a seventh of the host time per byte."""
    rng = np.random.RandomState(seed + int(chunk_id))
    nbits = int(rows) * int(num_indivs)
    raw = np.frombuffer(rng.bytes((nbits + 7) // 8), dtype=np.uint8)
    return np.unpackbits(raw)[:nbits].view(np.int8).reshape(int(rows), int(num_indivs))


def packed_chunk(chunk_id, rows, num_indivs, seed=20240):
    """synthetic_chunk's twin for 1-bit storage: uint8 [rows x ceil(N/8)], bit i of a row (LSB first, the order of
    _lib.pack_genotypes and PLINK .bed) = individual i's Bernoulli(0.5) genotype; pad bits are zero."""
    rb = (int(num_indivs) + 7) // 8
    nbytes = int(rows) * rb
    gen = np.random.Generator(np.random.SFC64(int(seed) * 1000003 + int(chunk_id)))
    raw = gen.integers(0, 2 ** 64, size=(nbytes + 7) // 8, dtype=np.uint64, endpoint=False).view(np.uint8)[:nbytes]
    raw = raw.reshape(int(rows), rb)
    if num_indivs % 8:
        raw[:, -1] &= np.uint8((1 << (num_indivs % 8)) - 1)
    return raw


def packed2_chunk(chunk_id, rows, num_indivs, seed=20240):
    """packed_chunk's twin for 0/1/2 codes (the plink2hdf5.py:171-179 coding) in 2-bit storage: uint8 [rows x ceil(N/4)],
    individual i in bits 2(i mod 4) .. 2(i mod 4)+1 of byte i // 4 (_lib.pack_genotypes(bits=2), a PLINK .bed row's order).
    Every code is the sum of two Bernoulli(0.5) alleles u + v (Hardy-Weinberg at allele frequency 0.5: 1/4, 1/2, 1/4), both
    bits of the code from the two bits of the generator's word it replaces: low = u xor v, high = u and v; never 3."""
    rb = (int(num_indivs) + 3) // 4
    nbytes = int(rows) * rb
    gen = np.random.Generator(np.random.SFC64(int(seed) * 1000003 + int(chunk_id)))
    w = gen.integers(0, 2 ** 64, size=(nbytes + 7) // 8, dtype=np.uint64, endpoint=False)
    lo_mask = np.uint64(0x5555555555555555)
    u = w & lo_mask
    w >>= np.uint64(1)
    w &= lo_mask                                              # v
    hi = u & w
    hi <<= np.uint64(1)
    u ^= w
    u |= hi
    raw = u.view(np.uint8)[:nbytes].reshape(int(rows), rb)
    if num_indivs % 4:
        raw[:, -1] &= np.uint8((1 << (2 * (num_indivs % 4))) - 1)
    return raw


def _fill_rows(args):
    """Worker of write_synthetic_container: rows [r0, r0 + rows) of one chromosome's raw_snps <- synthetic_chunk."""
    npy_path, r0, rows, num_indivs, chunk_id, seed = args
    mm = np.lib.format.open_memmap(npy_path, mode="r+")
    blk = synthetic_chunk(chunk_id, rows, num_indivs, seed)
    mm[r0:r0 + rows] = blk
    mm.flush()
    del mm
    return r0, blk.mean(axis=1, dtype=np.float64)


def write_synthetic_container(path, num_indivs, num_snps, chunk_rows=100000, seed=20240, num_chroms=5,
                              pheno_seed=20241, h2=0.8, num_causals=100, workers=1):
    """A genotype container (plink2hdf5.py layout, mixmogam_amd.chunkstore) of `num_snps` synthetic SNPs split over
    `num_chroms` chromosomes, written chunk by chunk (one `synthetic_chunk` per worker at a time is in host memory),
    plus a phenotype built from `num_causals` SNPs of the first chunk (simulations.py:64-85).  workers > 1: the
    chunks are generated and written by that many processes (the generator is one core's 1.4 GB/s).  Returns path."""
    import os
    from . import chunkstore
    st = chunkstore.Store(path, "w")
    gg = st.create_group("genot_data")
    ig = st.create_group("indiv_data")
    ig.create_dataset("indiv_ids", data=np.asarray(["i%d" % i for i in range(num_indivs)], dtype="S"))
    per = -(-num_snps // num_chroms)
    chunk_id, written = 0, 0
    jobs, chrom_rows = [], []
    for c in range(num_chroms):
        m_c = min(per, num_snps - written)
        if m_c <= 0:
            break
        cg = gg.create_group("chrom_%d" % (c + 1))
        raw = cg.create_dataset("raw_snps", shape=(m_c, num_indivs), dtype=np.int8)
        raw.flush()
        npy = os.path.join(path, "genot_data", "chrom_%d" % (c + 1), "raw_snps.npy")
        for r0 in range(0, m_c, chunk_rows):
            jobs.append((c, (npy, r0, min(chunk_rows, m_c - r0), num_indivs, chunk_id, seed)))
            chunk_id += 1
        cg.create_dataset("positions", data=np.arange(1, m_c + 1, dtype=np.int64))
        chrom_rows.append((cg, m_c))
        written += m_c
    freqs = [np.empty(m_c) for _cg, m_c in chrom_rows]
    if workers > 1:
        from concurrent.futures import ProcessPoolExecutor
        with ProcessPoolExecutor(max_workers=workers) as ex:
            for (c, _a), (r0, f) in zip(jobs, ex.map(_fill_rows, [a for _c, a in jobs])):
                freqs[c][r0:r0 + len(f)] = f
    else:
        for c, a in jobs:
            r0, f = _fill_rows(a)
            freqs[c][r0:r0 + len(f)] = f
    for (cg, _m), f in zip(chrom_rows, freqs):
        cg.create_dataset("freqs", data=f)
    first = synthetic_chunk(0, min(chunk_rows, per, num_snps), num_indivs, seed)
    ig.create_dataset("phenotypes", data=simulate_phenotype(first, h2=h2, num_causals=num_causals, seed=pheno_seed))
    st.create_dataset("num_snps", data=np.array(written))
    st.close()
    return path


class LazySyntheticGenotypes(object):
    """A `raw_snps` dataset that is never stored anywhere: rows are regenerated on every read from the counter-seeded
    generator (`synthetic_chunk(chunk_id0 + k, gen_rows, N, seed)` for generation chunk k), several generation chunks
    in parallel threads (the generator releases the GIL).  Supports what the chunked drivers do with a dataset:
    `len()`, `.shape`, `ds[i:j]`.  SURVEY 8d's config 5: "generated chunk-wise ..., never fully resident"."""

    dtype = np.dtype(np.int8)

    def __init__(self, num_indivs, num_snps, gen_rows=6250, seed=20240, chunk_id0=0, threads=8, packed=False):
        """packed: 1-bit rows (`raw_snps_packed`, LSB first: uint8 [M x ceil(N/8)]) that never exist expanded on the
        host -- the generator's 64-bit words ARE the rows (numpy's SFC64, one stream per generation chunk: 2 GB/s of
        packed bytes = 16 G genotypes/s per thread, where MT19937 `bytes` + unpackbits delivers 0.5 GB/s of int8).  A
        different, equally distributed matrix than the int8 form (`packed_chunk` regenerates any part of it).
        packed=2: 2-bit rows of 0/1/2 codes (`packed2_chunk`)."""
        self.num_indivs, self.packed = int(num_indivs), int(packed)        # 0: int8 rows, 1: 1-bit rows, 2: 2-bit 0/1/2 rows
        self.shape = (int(num_snps), (self.num_indivs * self.packed + 7) // 8 if packed else self.num_indivs)
        if packed:
            self.dtype = np.dtype(np.uint8)
        self.gen_rows, self.seed, self.chunk_id0, self.threads = int(gen_rows), int(seed), int(chunk_id0), int(threads)
        self.num_gen_chunks = -(-self.shape[0] // self.gen_rows)

    def __len__(self):
        return self.shape[0]

    def _gen(self, k):
        rows = min(self.gen_rows, self.shape[0] - k * self.gen_rows)
        if self.packed == 2:
            return packed2_chunk(self.chunk_id0 + k, rows, self.num_indivs, self.seed)
        if self.packed:
            return packed_chunk(self.chunk_id0 + k, rows, self.num_indivs, self.seed)
        return synthetic_chunk(self.chunk_id0 + k, rows, self.num_indivs, self.seed)

    def __getitem__(self, key):
        if isinstance(key, (int, np.integer)):
            return self[int(key):int(key) + 1][0]
        if key is Ellipsis:
            key = slice(None)
        lo, hi, step = key.indices(self.shape[0])
        if step != 1:
            raise IndexError("LazySyntheticGenotypes supports contiguous row ranges only")
        if hi <= lo:
            return np.zeros((0, self.shape[1]), dtype=self.dtype)
        k0, k1 = lo // self.gen_rows, (hi - 1) // self.gen_rows
        if k1 > k0 and self.threads > 1:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=min(self.threads, k1 - k0 + 1)) as ex:
                parts = list(ex.map(self._gen, range(k0, k1 + 1)))
        else:
            parts = [self._gen(k) for k in range(k0, k1 + 1)]
        parts[0] = parts[0][lo - k0 * self.gen_rows:]
        if k1 == k0:
            return np.ascontiguousarray(parts[0][:hi - lo])
        parts[-1] = parts[-1][:hi - k1 * self.gen_rows]
        return np.concatenate(parts)


def lazy_synthetic_source(num_indivs, num_snps, num_chroms=5, gen_rows=6250, seed=20240, pheno_seed=20241, h2=0.8,
                          num_causals=100, threads=8, packed=False):
    """A genot_data tree + phenotype over LazySyntheticGenotypes: what hdf5_data.run_emmax takes in place of a file
    name when the matrix (500 GB at config 5) must never exist anywhere.  `freqs` is the generator's nominal 0.5
    (every SNP passes any MAF filter below 0.5 - 5 sigma/sqrt(N))."""
    per = -(-num_snps // num_chroms)
    per = -(-per // gen_rows) * gen_rows                      # chromosomes start on generation-chunk boundaries
    tree, done, k0 = {}, 0, 0
    for c in range(num_chroms):
        m_c = min(per, num_snps - done)
        if m_c <= 0:
            break
        ds = LazySyntheticGenotypes(num_indivs, m_c, gen_rows, seed, chunk_id0=k0, threads=threads, packed=packed)
        tree["chrom_%d" % (c + 1)] = {"freqs": np.full(m_c, 0.5), "positions": np.arange(1, m_c + 1, dtype=np.int64)}
        if packed:                                            # hdf5_data._raw_dataset's packed layout
            tree["chrom_%d" % (c + 1)].update(raw_snps_packed=ds, packed_bits=np.array(int(packed)),
                                              num_indivs=np.array(num_indivs))
        else:
            tree["chrom_%d" % (c + 1)]["raw_snps"] = ds
        done += m_c
        k0 += -(-m_c // gen_rows)
    if packed:
        from ._lib import unpack_genotypes
        first = unpack_genotypes((packed2_chunk if int(packed) == 2 else packed_chunk)(0, min(gen_rows, num_snps), num_indivs, seed),
                                 num_indivs, int(packed))
    else:
        first = synthetic_chunk(0, min(gen_rows, num_snps), num_indivs, seed)
    y = simulate_phenotype(first, h2=h2, num_causals=min(num_causals, len(first)), seed=pheno_seed)
    return tree, y
