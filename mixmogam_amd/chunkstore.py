"""h5py-free container for the reference's HDF5 layouts, memory-mappable chunk by chunk.

The reference keeps genotypes, phenotypes, kinships and results in HDF5 files (h5py):
    genotype file   /genot_data/<chrom>/{raw_snps int8 [M_c x N], positions, freqs, snp_ids, nts, ...},
                    /indiv_data/{indiv_ids, sex, phenotypes}, /num_snps          plink2hdf5.py:27-28,111-118,226
    result file     /pseudo_heritability, /ve, /vg, /max_ll, /num_snps, /chrom_results/<chrom>/{ps, positions},
                    /kinship, /perm_min_ps, /perm_max_f_stats, /five_perc_*      hdf5_data.py:146-184,241-347
    kinship file    /kinship, /accessions, /n_snps                               kinship.py:145-167
h5py is not part of this image (and an HDF5 chunk cache is the wrong shape for a 500 GB genotype matrix that is
read exactly once, front to back, into pinned staging buffers), so the same trees are kept as a DIRECTORY:
    group   -> sub-directory
    dataset -> <name>.npy (numpy format 1.0/2.0 header + raw C-order data)
A dataset opens as a numpy memmap: `ds[i:j]` touches only those rows, `ds[...]` reads it all, `len(ds)` works --
the three things hdf5_data.py does with an h5py dataset -- and a writer can stream rows into a dataset created
with a shape (`create_dataset(name, shape=, dtype=)` returns a writable memmap).  `Store` mirrors the h5py calls the
reference makes (`File(name)`, `f['a']['b']`, `keys()`, `create_group`, `create_dataset(name, data=)`, `in`,
`del f[name]`, `flush`, `close`), so hdf5_data.py reads like the reference's.  `open_container` returns an
`h5py.File` instead when the path is a real HDF5 file and h5py is importable: both kinds of object drive the same
driver code.
"""
import json
import os
import shutil

import numpy as np

FORMAT = "mmg-chunkstore-1"


def _is_dataset_file(path):
    return path.endswith(".npy") and os.path.isfile(path)


class Group(object):
    """A directory of datasets (.npy) and sub-groups (directories)."""

    def __init__(self, path, mode):
        self._path, self._mode = path, mode
        self._open = []

    # ---- reading
    def keys(self):
        out = []
        for name in os.listdir(self._path):
            full = os.path.join(self._path, name)
            if os.path.isdir(full):
                out.append(name)
            elif _is_dataset_file(full):
                out.append(name[:-4])
        return sorted(out)                                   # h5py iterates in name order too

    def __iter__(self):
        return iter(self.keys())

    def __len__(self):
        return len(self.keys())

    def __contains__(self, name):
        full = os.path.join(self._path, name)
        return os.path.isdir(full) or os.path.isfile(full + ".npy")

    def __getitem__(self, name):
        full = os.path.join(self._path, name)
        if os.path.isdir(full):
            return Group(full, self._mode)
        if os.path.isfile(full + ".npy"):
            arr = np.load(full + ".npy", mmap_mode="r+" if self._mode != "r" else "r", allow_pickle=False)
            return arr                                       # numpy memmap (0-d arrays load as plain ndarrays)
        raise KeyError("%s has no member %r" % (self._path, name))

    # ---- writing
    def _writable(self):
        if self._mode == "r":
            raise IOError("container opened read-only: %s" % self._path)

    def create_group(self, name):
        self._writable()
        full = os.path.join(self._path, name)
        if os.path.exists(full):
            raise ValueError("group %r exists" % name)
        os.makedirs(full)
        return Group(full, self._mode)

    def require_group(self, name):
        return self[name] if name in self else self.create_group(name)

    def create_dataset(self, name, data=None, shape=None, dtype=None, compression=None, **_ignored):
        """data given: written at once (h5py's form, hdf5_data.py:146-150; `compression` is accepted and ignored --
        a memory-mapped stream wants raw rows).  shape/dtype given: a writable memmap to fill row block by row block."""
        self._writable()
        full = os.path.join(self._path, name) + ".npy"
        if os.path.exists(full) or os.path.isdir(full[:-4]):
            raise ValueError("dataset %r exists" % name)
        if data is not None:
            arr = np.asarray(data)
            if arr.dtype == object or arr.dtype.kind == "U":                 # lists of ids (kinship.py:165)
                arr = np.asarray([str(x) for x in arr.reshape(-1)], dtype="S").reshape(arr.shape)
            np.save(full, arr, allow_pickle=False)
            return self[name]
        if shape is None or dtype is None:
            raise ValueError("create_dataset needs data, or shape and dtype")
        mm = np.lib.format.open_memmap(full, mode="w+", dtype=np.dtype(dtype), shape=tuple(shape))
        self._open.append(mm)
        return mm

    def __delitem__(self, name):
        self._writable()
        full = os.path.join(self._path, name)
        if os.path.isdir(full):
            shutil.rmtree(full)
        elif os.path.isfile(full + ".npy"):
            os.remove(full + ".npy")
        else:
            raise KeyError(name)

    def flush(self):
        for mm in self._open:
            mm.flush()


class Store(Group):
    """Root group = the `h5py.File` of the reference.  mode: 'r', 'r+' / 'a' (create if missing), 'w' (truncate)."""

    def __init__(self, path, mode="a"):
        if mode == "w" and os.path.isdir(path):
            shutil.rmtree(path)
        if not os.path.isdir(path):
            if mode == "r":
                raise IOError("no such container: %s" % path)
            os.makedirs(path)
        meta = os.path.join(path, "mmgstore.json")
        if not os.path.isfile(meta):
            if mode == "r" and os.listdir(path):
                raise IOError("%s is not a %s container" % (path, FORMAT))
            if mode != "r":
                with open(meta, "w") as f:
                    json.dump({"format": FORMAT}, f)
        Group.__init__(self, path, "r" if mode == "r" else "a")
        self.filename = path

    def close(self):
        self.flush()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def is_store(path):
    return os.path.isdir(path) and os.path.isfile(os.path.join(path, "mmgstore.json"))


def open_container(filename, mode="a"):
    """`h5py.File(filename)` of the reference: a directory container (this module) or, for a real HDF5 file when
    h5py is installed, the h5py object itself.  A path that does not exist yet becomes a directory container
    unless it is named *.hdf5 / *.h5 and h5py is importable."""
    if os.path.isdir(filename):
        return Store(filename, mode)
    wants_hdf5 = os.path.isfile(filename) or filename.endswith((".hdf5", ".h5"))
    if wants_hdf5:
        try:
            import h5py
            return h5py.File(filename, mode)
        except ImportError:
            if os.path.isfile(filename):
                raise ImportError("%s is an HDF5 file and h5py is not installed; convert it once with "
                                  "mixmogam_amd.chunkstore.copy_tree where h5py is available, or pass a directory "
                                  "container" % filename)
    return Store(filename, mode)


def copy_tree(src, dst):
    """Copy every group / dataset of an h5py-like tree `src` into `dst` (either kind of container)."""
    for name in src.keys():
        item = src[name]
        if hasattr(item, "keys"):
            copy_tree(item, dst.create_group(name))
        else:
            dst.create_dataset(name, data=np.asarray(item[...]))


def write_genotype_container(path, chrom_snps, indiv_ids, phenotypes=None, positions=None, mode="w", packed_bits=0,
                             freqs=None):
    """Write a genotype file in the layout of plink2hdf5.py:27-28,111-118,226 from in-memory arrays:
    chrom_snps {chrom: int8 [M_c x N]}; freqs {chrom: [M_c]} (the allele frequency hdf5_data.py:91-93 filters on), default
    the parser's rule: mean / 2 for a chromosome with 0/1/2 codes (plink2hdf5.py:202), the carrier frequency for 0/1 codes.  packed_bits = 1 / 2: the rows are stored as `raw_snps_packed` (uint8, 1 or 2
    bits per genotype, low bits first) with `packed_bits` and `num_indivs` beside them instead of `raw_snps` -- an
    eighth / a quarter of the bytes for the streamed drivers to read and upload (hdf5_data._raw_dataset)."""
    st = Store(path, mode)
    gg = st.create_group("genot_data")
    ig = st.create_group("indiv_data")
    ig.create_dataset("indiv_ids", data=np.asarray([str(i) for i in indiv_ids], dtype="S"))
    if phenotypes is not None:
        ig.create_dataset("phenotypes", data=np.asarray(phenotypes, dtype=np.float64))
    total = 0
    for chrom, snps in chrom_snps.items():
        snps = np.ascontiguousarray(snps, dtype=np.int8)
        cg = gg.create_group(str(chrom))
        if packed_bits:
            from ._lib import pack_genotypes
            cg.create_dataset("raw_snps_packed", data=pack_genotypes(snps, packed_bits))
            cg.create_dataset("packed_bits", data=np.array(packed_bits))
            cg.create_dataset("num_indivs", data=np.array(snps.shape[1]))
        else:
            cg.create_dataset("raw_snps", data=snps)
        cg.create_dataset("positions", data=np.asarray(positions[chrom]) if positions is not None
                          else np.arange(len(snps), dtype=np.int64))
        if freqs is not None:
            cg.create_dataset("freqs", data=np.asarray(freqs[chrom], dtype=np.float64))
        else:
            cg.create_dataset("freqs", data=snps.mean(axis=1, dtype=np.float64) / (2.0 if snps.size and snps.max() > 1 else 1.0))
        total += len(snps)
    st.create_dataset("num_snps", data=np.array(total))
    st.close()
    return path
