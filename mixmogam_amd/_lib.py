"""ctypes binding of libmixmogam_hip.so (include/mixmogam_hip.h).

There is NO CPU fallback: if the library has not been built, cannot be loaded, or no HIP
device is present, every product entry point raises.  (The CPU oracle lives in oracle/ and is
test infrastructure only; nothing in this package imports it.)
"""
import ctypes as C
import os
import sys
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# MMG_LIB: another build of the same ABI (e.g. the `make EXPERIMENTS=1` library for A/B runs of superseded kernels)
LIB_PATH = os.environ.get("MMG_LIB") or os.path.join(_HERE, "lib", "libmixmogam_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "mixmogam_hip.h")

_lib = None
_lock = threading.Lock()

c_i8p = C.POINTER(C.c_int8)
c_f32p = C.POINTER(C.c_float)
c_f64p = C.POINTER(C.c_double)
c_i64p = C.POINTER(C.c_int64)
c_vp = C.c_void_p

# name -> (restype, argtypes); kept in sync with include/mixmogam_hip.h (tests check it)
PROTOTYPES = {
    "mmg_version": (C.c_int, []),
    "mmg_has_experiments": (C.c_int, []),
    "mmg_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "mmg_ctx_create": (C.c_int, [C.c_int, C.POINTER(c_vp)]),
    "mmg_ctx_destroy": (C.c_int, [c_vp]),
    "mmg_ctx_trim": (C.c_int, [c_vp]),
    "mmg_last_error": (C.c_char_p, [c_vp]),
    "mmg_device_info": (C.c_int, [c_vp, C.c_char_p, C.c_int, C.POINTER(C.c_int), c_i64p]),
    "mmg_device_pci_bus_id": (C.c_int, [c_vp, C.c_char_p, C.c_int]),
    "mmg_last_kernel_ms": (C.c_int, [c_vp, C.c_int, c_f64p]),
    "mmg_host_pin": (C.c_int, [c_vp, c_vp, C.c_int64]),
    "mmg_host_unpin": (C.c_int, [c_vp, c_vp]),
    "mmg_host_alloc": (C.c_int, [c_vp, C.c_int64, C.POINTER(c_vp)]),
    "mmg_host_free": (C.c_int, [c_vp, c_vp]),
    "mmg_geno_create": (C.c_int, [c_vp, C.c_int64, C.c_int32, C.POINTER(c_vp)]),
    "mmg_geno_destroy": (C.c_int, [c_vp, c_vp]),
    "mmg_geno_reset": (C.c_int, [c_vp, c_vp, C.c_int64]),
    "mmg_geno_upload": (C.c_int, [c_vp, c_vp, c_vp, C.c_int64, C.c_int64]),
    "mmg_geno_upload_f32": (C.c_int, [c_vp, c_vp, c_vp, C.c_int64, C.c_int64]),
    "mmg_geno_upload_f64": (C.c_int, [c_vp, c_vp, c_vp, C.c_int64, C.c_int64]),
    "mmg_geno_upload_packed": (C.c_int, [c_vp, c_vp, c_vp, C.c_int64, C.c_int64, C.c_int32, C.c_int64, c_vp]),
    "mmg_geno_download": (C.c_int, [c_vp, c_vp, c_vp, C.c_int64, C.c_int64]),
    "mmg_geno_download_rows": (C.c_int, [c_vp, c_vp, c_vp, C.c_int64, c_vp]),
    "mmg_geno_fill_hash": (C.c_int, [c_vp, c_vp, C.c_uint64, C.c_int64, C.c_uint32]),
    "mmg_geno_fill_structured": (C.c_int, [c_vp, c_vp, C.c_uint64, C.c_int64, C.c_int32, C.c_uint32]),
    "mmg_geno_snp_stats": (C.c_int, [c_vp, c_vp, c_vp, c_vp]),
    "mmg_geno_matvec": (C.c_int, [c_vp, c_vp, c_vp, C.c_int32, c_vp]),
    "mmg_kinship_ibs_i8": (C.c_int, [c_vp, c_vp, c_vp]),
    "mmg_kinship_indicator_i8": (C.c_int, [c_vp, c_vp, C.c_int32, c_vp]),
    "mmg_kinship_affine_f32": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp]),
    "mmg_kin_acc_create": (C.c_int, [c_vp, C.c_int32, C.POINTER(c_vp)]),
    "mmg_kin_acc_add": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp]),
    "mmg_kin_acc_add_grm": (C.c_int, [c_vp, c_vp, c_vp]),
    "mmg_kin_acc_pending": (C.c_int, [c_vp, c_vp, c_i64p]),
    "mmg_kin_acc_set_ibs": (C.c_int, [c_vp, c_vp, c_vp, c_vp, C.c_int64, C.c_int32]),
    "mmg_kin_acc_snps": (C.c_int, [c_vp, c_vp, c_i64p]),
    "mmg_kin_acc_fetch": (C.c_int, [c_vp, c_vp, c_vp, c_i64p]),
    "mmg_kin_acc_scale_k": (C.c_int, [c_vp, c_vp, c_f64p]),
    "mmg_kin_acc_destroy": (C.c_int, [c_vp, c_vp]),
    "mmg_kinship_i8": (C.c_int, [c_vp, c_vp, C.c_int64, C.c_int32, c_vp, c_vp, c_vp]),
    "mmg_scan_last_stats": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "mmg_scan_last_exact": (C.c_int, [c_vp, c_i64p]),
    "mmg_scan_deliver_begin": (C.c_int, [c_vp, c_vp, C.c_int64, c_vp, c_vp, c_vp]),
    "mmg_scan_deliver_wait": (C.c_int, [c_vp]),
    "mmg_eigh_f64": (C.c_int, [c_vp, c_vp, C.c_int32, c_vp, c_vp]),
    "mmg_dgemm_f64": (C.c_int, [c_vp, C.c_int, C.c_int, C.c_int32, C.c_int32, C.c_int32, c_vp, c_vp, c_vp]),
    "mmg_scan_set_model": (C.c_int, [c_vp, C.c_int32, c_vp, c_vp, C.c_int]),
    "mmg_emmax_scan": (C.c_int, [c_vp, c_vp, C.c_double, C.c_int32, c_vp, c_vp, c_vp]),
    "mmg_emmax_scan_device": (C.c_int, [c_vp, c_vp, C.c_double, C.c_int32]),
    "mmg_scan_fetch": (C.c_int, [c_vp, C.c_int64, c_vp, c_vp, c_vp]),
    "mmg_scan_fetch_stats": (C.c_int, [c_vp, C.c_int64, c_vp, c_vp, c_vp]),
    "mmg_emmax_scan_i8": (C.c_int, [c_vp, c_vp, C.c_int64, C.c_int32, c_vp, c_vp, C.c_double, C.c_int32,
                                    c_vp, c_vp, c_vp]),
    "mmg_kinship_f32": (C.c_int, [c_vp, c_vp, C.c_int64, C.c_int32, c_vp, c_vp, c_vp]),
    "mmg_emmax_scan_f32": (C.c_int, [c_vp, c_vp, C.c_int64, C.c_int32, c_vp, c_vp, C.c_double, C.c_int32,
                                     c_vp, c_vp, c_vp]),
    "mmg_emmax_perm_i8": (C.c_int, [c_vp, c_vp, C.c_int64, C.c_int32, c_vp, c_vp, C.c_int32, C.c_double, c_vp]),
    "mmg_emmax_perm": (C.c_int, [c_vp, c_vp, C.c_int32, c_vp, c_vp, C.c_int32, C.c_double, C.c_int, c_vp]),
    "mmg_reml_create": (C.c_int, [c_vp, C.c_int32, C.c_int32, c_vp, c_vp, c_vp, C.POINTER(c_vp)]),
    "mmg_reml_create_from_acc": (C.c_int, [c_vp, c_vp, C.c_int32, c_vp, c_vp, C.POINTER(c_vp)]),
    "mmg_reml_destroy": (C.c_int, [c_vp, c_vp]),
    "mmg_reml_sums": (C.c_int, [c_vp, c_vp, C.c_int32, c_vp, c_vp, c_vp, c_vp, c_vp, c_f64p]),
    "mmg_reml_sums_ex": (C.c_int, [c_vp, c_vp, C.c_int32, c_vp, c_vp, c_vp, c_vp, c_vp, c_f64p, C.c_int32]),
    "mmg_reml_band_factor": (C.c_int, [c_vp, c_vp, C.c_int32, c_vp]),
    "mmg_reml_sums_ml": (C.c_int, [c_vp, c_vp, C.c_int32, c_vp, c_vp, c_vp, c_vp, c_vp, C.c_int32]),
    "mmg_reml_band_info": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp]),
    "mmg_reml_scan_model": (C.c_int, [c_vp, c_vp, C.c_double, C.c_int, c_f64p, c_vp, c_f64p]),
    "mmg_reml_scan_model_c": (C.c_int, [c_vp, c_vp, C.c_double, C.c_int, c_f64p, c_vp, c_f64p, c_vp]),
    "mmg_reml_linv_apply": (C.c_int, [c_vp, c_vp, C.c_double, C.c_int32, c_vp, C.c_int32, c_vp]),
    "mmg_reml_linv_fetch": (C.c_int, [c_vp, c_vp, C.c_double, c_vp]),
    "mmg_perm_plan_create_from_reml": (C.c_int, [c_vp, c_vp, C.c_double, c_vp, C.c_int32, C.c_double, C.c_int, C.POINTER(c_vp)]),
    "mmg_rot_create": (C.c_int, [c_vp, C.c_int32, c_vp, C.c_int64, C.POINTER(c_vp)]),
    "mmg_rot_destroy": (C.c_int, [c_vp, c_vp]),
    "mmg_rot_load": (C.c_int, [c_vp, c_vp, c_vp]),
    "mmg_rot_fetch": (C.c_int, [c_vp, c_vp, C.c_int64, C.c_int64, c_vp]),
    "mmg_emmax_scan_multi": (C.c_int, [c_vp, c_vp, C.c_int32, C.c_int32, c_vp, c_vp, c_vp, c_vp, C.c_int32,
                                       c_vp, c_vp, c_vp]),
    "mmg_kinship_ibs_i8_sharded": (C.c_int, [c_vp, c_vp, c_vp, c_vp]),
    "mmg_kinship_ibs_f64": (C.c_int, [c_vp, c_vp, c_vp, C.c_int64, C.c_int32, c_vp]),
    "mmg_kinship_ibs_diploid_f64": (C.c_int, [c_vp, c_vp, C.c_int32, c_vp]),
    "mmg_kin_acc_allreduce": (C.c_int, [c_vp, c_vp, c_vp]),
    "mmg_emmax_perm_sharded": (C.c_int, [c_vp, c_vp, c_vp, C.c_int32, c_vp, c_vp, C.c_int32, C.c_double, C.c_int,
                                         c_vp]),
    "mmg_emmax_perm_after_scan": (C.c_int, [c_vp, c_vp, c_vp, C.c_int32, c_vp, c_vp, C.c_int32, C.c_double, c_vp,
                                            C.c_int32, c_vp]),
    "mmg_perm_plan_create": (C.c_int, [c_vp, C.c_int32, c_vp, c_vp, C.c_int32, C.c_double, C.POINTER(c_vp)]),
    "mmg_perm_plan_create_ex": (C.c_int, [c_vp, C.c_int32, c_vp, c_vp, C.c_int32, C.c_double, C.c_int, C.POINTER(c_vp)]),
    "mmg_perm_plan_run": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, C.c_int32, c_vp]),
    "mmg_perm_plan_destroy": (C.c_int, [c_vp, c_vp]),
    "mmg_comm_info": (C.c_int, [c_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "mmg_comm_allgather_f64": (C.c_int, [c_vp, c_vp, c_vp, C.c_int64, c_vp]),
    "mmg_f_sf": (C.c_int, [c_vp, c_vp, C.c_int64, C.c_int32, c_vp]),
    "mmg_comm_unique_id": (C.c_int, [c_vp]),
    "mmg_comm_create": (C.c_int, [c_vp, c_vp, C.c_int, C.c_int, C.POINTER(c_vp)]),
    "mmg_comm_destroy": (C.c_int, [c_vp, c_vp]),
    "mmg_comm_allgather_scan": (C.c_int, [c_vp, c_vp, C.c_int64, c_vp, c_vp, c_vp]),
    "mmg_comm_allreduce_f64": (C.c_int, [c_vp, c_vp, c_vp, C.c_int64, C.c_int]),
    "mmg_comm_allreduce_i64": (C.c_int, [c_vp, c_vp, c_vp, C.c_int64, C.c_int]),
    "mmg_comm_barrier": (C.c_int, [c_vp, c_vp]),
}


class MixmogamHipError(RuntimeError):
    pass


def load():
    """dlopen the library and bind every prototype.  Raises if it is missing."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.isfile(LIB_PATH):
            raise MixmogamHipError(
                "libmixmogam_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; "
                "g.build()'` or `make -C mixmogam_amd/csrc`. There is no CPU fallback." % LIB_PATH)
        lib = C.CDLL(LIB_PATH)   # RTLD_LOCAL: keep the system ROCm libraries it links out of other modules' way
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if hasattr(lib, "mmg_guard_check"):  # diagnostic build (make GUARD=1, MMG_LIB=...): live buffers are checked at exit
            import atexit
            lib.mmg_guard_check.restype = C.c_long
            atexit.register(lambda: sys.stderr.write("[mmg guard] damaged guard bands: %d\n" % lib.mmg_guard_check()))
        _lib = lib
        return lib


def device_count():
    """Number of HIP devices the library sees (mmg_device_count); 0 when there is none or the runtime fails."""
    n = C.c_int(0)
    rc = load().mmg_device_count(C.byref(n))
    return n.value if rc == 0 else 0


def _ptr(a):
    return None if a is None else a.ctypes.data_as(c_vp)


def _arr(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def as_store_array(snps):
    """[rows x N] genotypes in a dtype the store ingests without silent damage: int8 as is; wider integer / bool
    dtypes are range-checked on the host and narrowed; float32 / float64 pass through (the device conversion
    kernel rejects non-integral, out-of-range or NaN values); anything else goes through float64.  The reference
    takes arbitrary numeric SNPs (sp.matrix(snps, dtype='single'), linear_models.py:1317); this store is int8, so
    dosages or normalised SNPs raise instead of being rounded or wrapped."""
    a = np.asarray(snps)
    if a.ndim != 2:
        raise ValueError("snps must be [num_snps x num_individuals]")
    if a.dtype == np.int8 or a.dtype in (np.float32, np.float64):
        # int8 -128 is found by the device (mmg_geno_upload fails and rolls the rows back): a host-side min() over the
        # block costs more than moving it (measured: 19 instead of 50 GB/s end to end)
        return np.ascontiguousarray(a)
    if a.dtype == np.bool_:
        return np.ascontiguousarray(a, dtype=np.int8)
    if np.issubdtype(a.dtype, np.integer):
        if a.size and (int(a.min()) < -127 or int(a.max()) > 127):
            raise ValueError("genotype values must lie in [-127, 127] (int8 store); got [%d, %d]"
                             % (int(a.min()), int(a.max())))
        return np.ascontiguousarray(a, dtype=np.int8)
    return np.ascontiguousarray(a, dtype=np.float64)


def pack_genotypes(snps, bits=1):
    """[rows x N] genotypes with values in [0, 2^bits) -> uint8 [rows x ceil(N*bits/8)], genotype i in the low bits
    first (numpy.packbits(bitorder='little') for bits=1; the bit order of a PLINK .bed row for bits=2)."""
    a = np.asarray(snps)
    if a.ndim != 2 or bits not in (1, 2):
        raise ValueError("snps must be [rows x N], bits 1 or 2")
    if a.size and (int(a.min()) < 0 or int(a.max()) >= (1 << bits)):
        raise ValueError("genotype values must lie in [0, %d] for %d-bit packing" % ((1 << bits) - 1, bits))
    a = a.astype(np.uint8)
    if bits == 1:
        return np.packbits(a, axis=1, bitorder='little')
    n = a.shape[1]
    pad = (-n) % 4
    if pad:
        a = np.hstack([a, np.zeros((a.shape[0], pad), dtype=np.uint8)])
    q = a.reshape(a.shape[0], -1, 4)
    return (q[:, :, 0] | (q[:, :, 1] << 2) | (q[:, :, 2] << 4) | (q[:, :, 3] << 6)).astype(np.uint8)


def unpack_genotypes(packed, n, bits=1):
    """Inverse of pack_genotypes on the host (tests, CPU-side consumers of packed containers)."""
    p = np.asarray(packed, dtype=np.uint8)
    if bits == 1:
        return np.unpackbits(p, axis=1, bitorder='little')[:, :n].astype(np.int8)
    out = np.empty((p.shape[0], p.shape[1] * 4), dtype=np.int8)
    for k in range(4):
        out[:, k::4] = (p >> (2 * k)) & 3
    return out[:, :n]


class Geno(object):
    """Device-resident genotype store ([M x N] int8, SNP-major, padded in HBM)."""

    def __init__(self, ctx, M, N):
        self.ctx, self.M, self.N = ctx, int(M), int(N)
        h = c_vp()
        ctx._check(ctx.lib.mmg_geno_create(ctx.h, self.M, self.N, C.byref(h)))
        self.h = h

    def reset(self, M):
        """Reuse the allocation for another block of M <= capacity SNPs (no hipFree / hipMalloc)."""
        self.ctx._check(self.ctx.lib.mmg_geno_reset(self.ctx.h, self.h, int(M)))
        self.M = int(M)
        return self

    def upload(self, snps, m0=0):
        a = as_store_array(snps)
        if a.shape[1] != self.N:
            raise ValueError("expected [rows x %d] genotypes, got %r" % (self.N, a.shape))
        if a.dtype == np.float32:
            fn = self.ctx.lib.mmg_geno_upload_f32
        elif a.dtype == np.float64:
            fn = self.ctx.lib.mmg_geno_upload_f64
        else:
            fn = self.ctx.lib.mmg_geno_upload
        self.ctx._check(fn(self.ctx.h, self.h, _ptr(a), int(m0), a.shape[0]))
        return self

    def upload_packed(self, packed, bits=1, m0=0, lut=None):
        """Rows packed 1 or 2 bits per genotype, least significant bit first (pack_genotypes; a PLINK .bed row for
        bits=2), expanded on the device.  packed: uint8 [rows x >= ceil(N*bits/8)]; lut: the int8 value per code."""
        a = np.ascontiguousarray(packed, dtype=np.uint8)
        if a.ndim != 2 or a.shape[1] < (self.N * bits + 7) // 8:
            raise ValueError("expected [rows x >= %d] packed bytes, got %r" % ((self.N * bits + 7) // 8, a.shape))
        lt = None if lut is None else np.ascontiguousarray(lut, dtype=np.int8)
        if lt is not None and lt.size != (1 << bits):
            raise ValueError("lut needs %d entries" % (1 << bits))
        self.ctx._check(self.ctx.lib.mmg_geno_upload_packed(self.ctx.h, self.h, _ptr(a), int(m0), a.shape[0], int(bits),
                                                            a.shape[1], _ptr(lt)))
        return self

    def download(self, m0=0, rows=None):
        rows = self.M - m0 if rows is None else rows
        out = np.empty((rows, self.N), dtype=np.int8)
        self.ctx._check(self.ctx.lib.mmg_geno_download(self.ctx.h, self.h, _ptr(out), int(m0), int(rows)))
        return out

    def download_rows(self, idx):
        """Rows idx (any order, repeats allowed) gathered on the device: only len(idx) x N bytes cross PCIe."""
        idx = _arr(np.asarray(idx).reshape(-1), np.int64)
        out = np.empty((len(idx), self.N), dtype=np.int8)
        self.ctx._check(self.ctx.lib.mmg_geno_download_rows(self.ctx.h, self.h, _ptr(idx), len(idx), _ptr(out)))
        return out

    def fill_hash(self, seed, m_global0=0, thr16=32768):
        self.ctx._check(self.ctx.lib.mmg_geno_fill_hash(self.ctx.h, self.h, int(seed), int(m_global0), int(thr16)))
        return self

    def fill_structured(self, seed, m_global0=0, npop=3, spread_q16=9830):
        self.ctx._check(self.ctx.lib.mmg_geno_fill_structured(self.ctx.h, self.h, int(seed), int(m_global0), int(npop),
                                                              int(spread_q16)))
        return self

    def snp_stats(self):
        mean = np.empty(self.M)
        sd = np.empty(self.M)
        self.ctx._check(self.ctx.lib.mmg_geno_snp_stats(self.ctx.h, self.h, _ptr(mean), _ptr(sd)))
        return mean, sd

    def matvec(self, V):
        V = _arr(np.atleast_2d(V), np.float64)
        assert V.shape[1] == self.N
        out = np.empty((V.shape[0], self.M))
        self.ctx._check(self.ctx.lib.mmg_geno_matvec(self.ctx.h, self.h, _ptr(V), V.shape[0], _ptr(out)))
        return out

    def close(self):
        if self.h is not None:
            self.ctx.lib.mmg_geno_destroy(self.ctx.h, self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class KinshipAccumulator(object):
    """Device-resident sum of x x' over successive genotype chunks (mmg_kin_acc_*)."""

    def __init__(self, ctx, N):
        self.ctx, self.N = ctx, int(N)
        h = c_vp()
        ctx._check(ctx.lib.mmg_kin_acc_create(ctx.h, self.N, C.byref(h)))
        self.h = h

    def add(self, g, scale=None, shift=None):
        sc = None if scale is None else _arr(scale, np.float32)
        sh = None if shift is None else _arr(shift, np.float32)
        self.ctx._check(self.ctx.lib.mmg_kin_acc_add(self.ctx.h, self.h, g.h, _ptr(sc), _ptr(sh)))

    def add_grm(self, g):
        """acc += sum_m z_m z_m', z = (s - mean)/std per SNP: the exact int8 route (mmg_kin_acc_add_grm)."""
        self.ctx._check(self.ctx.lib.mmg_kin_acc_add_grm(self.ctx.h, self.h, g.h))

    def set_ibs(self, g, scaled=True, comm=None, m_total=None):
        """The accumulator's matrix := the IBS kinship of g (of all ranks' blocks with comm / m_total), kept in HBM
        (mmg_kin_acc_set_ibs)."""
        self.ctx._check(self.ctx.lib.mmg_kin_acc_set_ibs(self.ctx.h, comm, self.h, g.h, int(g.M if m_total is None else m_total),
                                                         1 if scaled else 0))
        return self

    def pending(self):
        """SNPs of the last add_grm calls whose sums still sit in the int32 digit planes (combined into the fp64 sum by
        the next fetch / scale_k / allreduce, or when a call cannot join the run): mmg_kin_acc_pending."""
        n = C.c_int64(0)
        self.ctx._check(self.ctx.lib.mmg_kin_acc_pending(self.ctx.h, self.h, C.byref(n)))
        return n.value

    def allreduce(self, comm):
        """Sum the device-resident accumulator (and its SNP count) over the ranks of `comm` in HBM."""
        self.ctx._check(self.ctx.lib.mmg_kin_acc_allreduce(self.ctx.h, comm, self.h))

    def scale_k(self):
        """kinship.scale_k (kinship.py:94-100) applied to the device-resident sum in place; returns the factor.  The
        rule does not change under a prior division by the SNP count, so fetch() afterwards yields the scaled kinship."""
        f = C.c_double(0.0)
        self.ctx._check(self.ctx.lib.mmg_kin_acc_scale_k(self.ctx.h, self.h, C.byref(f)))
        return f.value

    def fetch(self, via=None):
        """(matrix, SNP count).  via: another Context of the same device whose stream carries the download (a helper thread
        fetching while this accumulator's own context computes)."""
        out = np.empty((self.N, self.N))
        n = C.c_int64(0)
        c = via if via is not None else self.ctx
        c._check(c.lib.mmg_kin_acc_fetch(c.h, self.h, _ptr(out), C.byref(n)))
        return out, n.value

    def snps(self):
        n = C.c_int64(0)
        self.ctx._check(self.ctx.lib.mmg_kin_acc_snps(self.ctx.h, self.h, C.byref(n)))
        return n.value

    def close(self):
        if self.h is not None:
            self.ctx.lib.mmg_kin_acc_destroy(self.ctx.h, self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Rot(object):
    """Eigen-rotated genotype store T[i][m] = u_i . s_m in HBM (mmg_rot_*): the SNP-dependent input of every
    EMMAX scan that shares the kinship's eigenvectors, whatever its delta / phenotype / covariates."""

    def __init__(self, ctx, evecs_rows, M_cap):
        V = _arr(evecs_rows, np.float64)
        assert V.ndim == 2 and V.shape[0] == V.shape[1]
        self.ctx, self.N, self.M_cap, self.M = ctx, V.shape[0], int(M_cap), 0
        h = c_vp()
        ctx._check(ctx.lib.mmg_rot_create(ctx.h, self.N, _ptr(V), self.M_cap, C.byref(h)))
        self.h = h

    def load(self, g):
        self.ctx._check(self.ctx.lib.mmg_rot_load(self.ctx.h, self.h, g.h))
        self.M = g.M
        return self

    def fetch(self, m0=0, rows=None):
        rows = self.M - m0 if rows is None else rows
        out = np.empty((self.N, rows))
        self.ctx._check(self.ctx.lib.mmg_rot_fetch(self.ctx.h, self.h, int(m0), int(rows), _ptr(out)))
        return out

    def close(self):
        if self.h is not None:
            self.ctx.lib.mmg_rot_destroy(self.ctx.h, self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PermPlan(object):
    """SNP-independent half of the EMMAX permutation test on the device (mmg_perm_plan_*): built once per (H, Ys),
    run over any number of genotype stores."""

    def __init__(self, ctx, H, Ys, h0_rss, centre_snps=True, reml=None, delta=None, centre_H=False):
        """H: the N x N matrix, or None with reml= / delta=: H = L^-1 of K + delta I = L L' as the REML workspace holds it in
        HBM (mmg_perm_plan_create_from_reml).  centre_H: H <- C H on the device (the public test, flags bit 1)."""
        Ys = _arr(Ys, np.float64)
        flags = (0 if centre_snps else 1) | (2 if centre_H else 0)
        h = c_vp()
        if reml is not None:
            self.ctx, self.N, self.P = ctx, reml.N, Ys.shape[1]
            assert H is None and Ys.shape[0] == self.N
            ctx._check(ctx.lib.mmg_perm_plan_create_from_reml(ctx.h, reml.h, float(delta), _ptr(Ys), self.P, float(h0_rss),
                                                              flags, C.byref(h)))
        else:
            H = _arr(H, np.float64)
            self.ctx, self.N, self.P = ctx, H.shape[0], Ys.shape[1]
            assert H.shape == (self.N, self.N) and Ys.shape[0] == self.N
            ctx._check(ctx.lib.mmg_perm_plan_create_ex(ctx.h, self.N, _ptr(H), _ptr(Ys), self.P, float(h0_rss), flags,
                                                       C.byref(h)))
        self.h = h

    def run(self, g, comm=None, after_scan_HtQ=None):
        out = np.empty(self.P)
        U, q = None, 0
        if after_scan_HtQ is not None:
            U = _arr(np.atleast_2d(after_scan_HtQ), np.float64)
            assert U.shape[1] == self.N
            q = U.shape[0]
        self.ctx._check(self.ctx.lib.mmg_perm_plan_run(self.ctx.h, comm, self.h, g.h, _ptr(U), q, _ptr(out)))
        return out

    def close(self):
        if self.h is not None:
            self.ctx.lib.mmg_perm_plan_destroy(self.ctx.h, self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceKinship(object):
    """A kinship matrix that lives in HBM: the accumulator a streamed kinship pass filled (and scaled).  Stands in for the
    N x N array where the next consumer is on the device too (Context.reml, kinship.scale_k); numpy sees it through
    __array__ / host(), which download it once.  `host_future`: a concurrent.futures future of the downloaded matrix, when
    a helper thread is fetching it in the background (hdf5_data.run_emmax)."""

    def __init__(self, acc, scaled=False, host_future=None):
        self.acc, self.N, self.shape, self.ndim = acc, acc.N, (acc.N, acc.N), 2
        self._host, self.host_future, self.scaled = None, host_future, scaled

    def host(self):
        if self._host is None:
            self._host = self.host_future.result() if self.host_future is not None else self.acc.fetch()[0]
        return self._host

    def __array__(self, dtype=None, copy=None):
        a = self.host()
        return a if dtype is None else a.astype(dtype, copy=False)

    def __len__(self):
        return self.N

    def close(self):
        if self.acc is not None:
            self.acc.close()
            self.acc = None


class Reml(object):
    """Eigendecomposition-free REML workspace (mmg_reml_*): K, X, y resident; the four likelihood sums per delta from
    one Cholesky factorisation each; the scan model P(delta), P y built on the device."""

    def __init__(self, ctx, K, X, y):
        X = _arr(X, np.float64)
        y = _arr(np.asarray(y).reshape(-1), np.float64)
        h = c_vp()
        if isinstance(K, DeviceKinship):                       # the kinship as a streamed pass left it in HBM: no host visit
            self.ctx, self.N, self.q = ctx, K.N, X.shape[1]
            assert X.shape[0] == self.N and len(y) == self.N
            ctx._check(ctx.lib.mmg_reml_create_from_acc(ctx.h, K.acc.h, self.q, _ptr(X), _ptr(y), C.byref(h)))
        else:
            K = _arr(K, np.float64)
            self.ctx, self.N, self.q = ctx, K.shape[0], X.shape[1]
            assert K.shape == (self.N, self.N) and X.shape[0] == self.N and len(y) == self.N
            ctx._check(ctx.lib.mmg_reml_create(ctx.h, self.N, self.q, _ptr(K), _ptr(X), _ptr(y), C.byref(h)))
        self.h = h

    ROUTES = {"auto": 0, "chol": 1, "band": 2}

    def band_info(self):
        """{'ready': K has been reduced, 'householder_fallback': a Cholesky-QR panel was rank deficient and the reduction
        was redone with Householder panels, 'seconds': what the reduction took} (mmg_reml_band_info)."""
        ready, fb, sec = C.c_int32(0), C.c_int32(0), C.c_double(0.0)
        self.ctx._check(self.ctx.lib.mmg_reml_band_info(self.ctx.h, self.h, C.byref(ready), C.byref(fb), C.byref(sec)))
        return {"ready": bool(ready.value), "householder_fallback": bool(fb.value), "seconds": sec.value}

    def uses_band(self, route="auto"):
        """Whether sums(route) goes through the band reduction: mmg_reml_sums_ex's AUTO rule, including its 'a workspace
        that has been reduced stays on the band route' term (advisor r3)."""
        if route != "auto":
            return route == "band"
        env = os.environ.get("MMG_REML_ROUTE", "")
        return env != "chol" and (env == "band" or self.N >= 256 or self.band_info()["ready"])

    def sums(self, deltas, route="auto"):
        """s1..s4 for every delta, sum_sq_etas.  route: 'chol' = one Cholesky factorisation per delta, 'band' = K reduced
        once to a band matrix and every delta from that (csrc/reml_band.hip), 'auto' = band from N = 256 up."""
        d = _arr(np.asarray(deltas).reshape(-1), np.float64)
        out = [np.empty(len(d)) for _ in range(4)]
        sse = C.c_double(0.0)
        self.ctx._check(self.ctx.lib.mmg_reml_sums_ex(self.ctx.h, self.h, len(d), _ptr(d), *[_ptr(o) for o in out],
                                                      C.byref(sse), self.ROUTES[route]))
        return out[0], out[1], out[2], out[3], sse.value

    def band_factor(self, deltas):
        """mmg_reml_band_factor: factor B + delta I for all of `deltas` (<= 256) in one sweep and keep the factors; sums() calls
        whose variance ratios are all among them then cost the substitutions and the trace recurrence only."""
        d = _arr(np.asarray(deltas).reshape(-1), np.float64)
        self.ctx._check(self.ctx.lib.mmg_reml_band_factor(self.ctx.h, self.h, len(d), _ptr(d)))

    def sums_ml(self, deltas, route="auto"):
        """s1, s3, log|K + delta I|, tr (K + delta I)^-1 for every delta: what the ML likelihood needs (mmg_reml_sums_ml)."""
        d = _arr(np.asarray(deltas).reshape(-1), np.float64)
        out = [np.empty(len(d)) for _ in range(4)]
        self.ctx._check(self.ctx.lib.mmg_reml_sums_ml(self.ctx.h, self.h, len(d), _ptr(d), *[_ptr(o) for o in out],
                                                      self.ROUTES[route]))
        return tuple(out)

    def scan_model(self, delta, ndigits=0, want_C=False):
        """Load the EMMAX scan model of `delta` into the context; returns (h0_rss, beta) -- and, want_C, the q x N matrix
        (X'V^-1 X)^-1 X'V^-1 of the with_betas form (mmg_reml_scan_model_c)."""
        h0, mah = C.c_double(0.0), C.c_double(0.0)
        beta = np.empty(self.q)
        Cm = np.empty((self.q, self.N)) if want_C else None
        self.ctx._check(self.ctx.lib.mmg_reml_scan_model_c(self.ctx.h, self.h, float(delta), int(ndigits), C.byref(h0),
                                                           _ptr(beta), C.byref(mah), _ptr(Cm)))
        return (h0.value, beta, Cm) if want_C else (h0.value, beta)

    def linv_apply(self, delta, V, trans=False):
        """L^-1 V (or L^-T V) for K + delta I = L L': H V for the square root H = L^-1 of (K + delta I)^-1 that stands in for
        the reference's H_sqrt_inv (mmg_reml_linv_apply).  V: [N] or [N x k]; returns the same shape."""
        V = np.asarray(V, dtype=np.float64)
        one = V.ndim == 1
        Vc = np.asfortranarray(V.reshape(self.N, -1))
        out = np.empty(Vc.shape, order='F')
        self.ctx._check(self.ctx.lib.mmg_reml_linv_apply(self.ctx.h, self.h, float(delta), 1 if trans else 0, _ptr(Vc), Vc.shape[1],
                                                         _ptr(out)))
        return out[:, 0].copy() if one else out

    def linv(self, delta):
        """L^-1 itself, [N x N] lower triangular: a matrix H with H'H = (K + delta I)^-1 (mmg_reml_linv_fetch)."""
        H = np.empty((self.N, self.N))
        self.ctx._check(self.ctx.lib.mmg_reml_linv_fetch(self.ctx.h, self.h, float(delta), _ptr(H)))
        return H

    def perm_plan(self, delta, Ys, h0_rss, centre_snps=True, centre_H=False):
        return PermPlan(self.ctx, None, Ys, h0_rss, centre_snps=centre_snps, reml=self, delta=delta, centre_H=centre_H)

    def close(self):
        if self.h is not None:
            self.ctx.lib.mmg_reml_destroy(self.ctx.h, self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def scan_multi_batch(q=1):
    """Phenotypes per HBM pass of mmg_emmax_scan_multi (api.hip): 16 on the fp64 matrix pipe for q <= 2 fixed-effect
    columns, else 8 (also with MMG_MULTI_KERNEL=valu / MMG_MULTI_PB=8, A/B runs)."""
    if os.environ.get("MMG_MULTI_KERNEL") == "valu" or os.environ.get("MMG_MULTI_PB") == "8" or q > 2:
        return 8
    return 16


class Context(object):
    """One HIP device context (stream, scan model, result buffers)."""

    KERNEL_SLOTS = {"kinship": 0, "scan_quad": 1, "scan_finalize": 2, "perm": 3, "eigh": 4, "pack": 5,
                    "rotate": 7, "scan_multi": 8, "grm": 9}

    def __init__(self, device=0):
        self.lib = load()
        n = C.c_int(0)
        rc = self.lib.mmg_device_count(C.byref(n))
        if rc != 0 or n.value <= 0:
            raise MixmogamHipError("no HIP device available (mmg_device_count -> %d devices): %s; "
                                   "this package has no CPU path" %
                                   (n.value, (self.lib.mmg_last_error(None) or b"").decode()))
        h = c_vp()
        rc = self.lib.mmg_ctx_create(int(device), C.byref(h))
        if rc != 0:
            raise MixmogamHipError("mmg_ctx_create(%d) failed: %s" %
                                   (device, (self.lib.mmg_last_error(None) or b"").decode()))
        self.h = h
        self.device = int(device)

    def _check(self, rc):
        if rc != 0:
            raise MixmogamHipError("libmixmogam_hip error %d: %s" %
                                   (rc, (self.lib.mmg_last_error(self.h) or b"").decode()))

    def device_info(self):
        name = C.create_string_buffer(64)
        ncu = C.c_int(0)
        mem = C.c_int64(0)
        self._check(self.lib.mmg_device_info(self.h, name, 64, C.byref(ncu), C.byref(mem)))
        bus = C.create_string_buffer(32)
        self._check(self.lib.mmg_device_pci_bus_id(self.h, bus, 32))
        return {"arch": name.value.decode(), "n_cu": ncu.value, "hbm_bytes": mem.value, "pci_bus_id": bus.value.decode()}

    def kernel_ms(self, which):
        ms = C.c_double(0.0)
        self._check(self.lib.mmg_last_kernel_ms(self.h, self.KERNEL_SLOTS[which], C.byref(ms)))
        return ms.value

    def pinned_empty(self, n, dtype=np.float64):
        """1-D numpy array in page-locked host memory owned by the library (result fetches into it
        run at PCIe rate).  Freed when the array is garbage collected."""
        import weakref
        nbytes = int(n) * np.dtype(dtype).itemsize
        p = c_vp()
        self._check(self.lib.mmg_host_alloc(self.h, max(nbytes, 1), C.byref(p)))
        buf = (C.c_char * max(nbytes, 1)).from_address(p.value)
        arr = np.frombuffer(buf, dtype=dtype, count=int(n))
        lib, h, addr = self.lib, self.h, p.value
        weakref.finalize(buf, lambda: lib.mmg_host_free(h, c_vp(addr)))
        return arr

    # --- genotype store
    def geno(self, snps=None, M=None, N=None):
        if snps is not None:
            a = as_store_array(snps)
            g = Geno(self, a.shape[0], a.shape[1])
            if a.shape[0]:
                g.upload(a)
            return g
        return Geno(self, M, N)

    # --- kinship
    def kinship_ibs_counts(self, g, comm=None):
        """comm: RCCL communicator handle -- the counts then cover the SNP blocks of all ranks (summed in HBM)."""
        out = np.empty((g.N, g.N), dtype=np.int64)
        self._check(self.lib.mmg_kinship_ibs_i8_sharded(self.h, comm, g.h, _ptr(out)))
        return out

    def kinship_ibs(self, g, scaled=True, comm=None, m_total=None):
        """The IBS kinship counts / (2 M) + 0.5 (scale_k'd when scaled) converted and scaled on the device
        (mmg_kinship_ibs_f64); comm / m_total: all ranks' SNP blocks."""
        out = np.empty((g.N, g.N), dtype=np.float64)
        self._check(self.lib.mmg_kinship_ibs_f64(self.h, comm, g.h, int(g.M if m_total is None else m_total), 1 if scaled else 0,
                                                 _ptr(out)))
        return out

    def kinship_ibs_diploid(self, g, scaled=True):
        """'diploid_int' IBS kinship of a 0/1/2 store, combined and scaled on the device (mmg_kinship_ibs_diploid_f64)."""
        out = np.empty((g.N, g.N), dtype=np.float64)
        self._check(self.lib.mmg_kinship_ibs_diploid_f64(self.h, g.h, 1 if scaled else 0, _ptr(out)))
        return out

    def kinship_indicator_counts(self, g, thr):
        out = np.empty((g.N, g.N), dtype=np.int64)
        self._check(self.lib.mmg_kinship_indicator_i8(self.h, g.h, int(thr), _ptr(out)))
        return out

    def kinship_affine(self, g, scale=None, shift=None):
        out = np.empty((g.N, g.N), dtype=np.float64)
        sc = None if scale is None else _arr(scale, np.float32)
        sh = None if shift is None else _arr(shift, np.float32)
        self._check(self.lib.mmg_kinship_affine_f32(self.h, g.h, _ptr(sc), _ptr(sh), _ptr(out)))
        return out

    def kinship_accumulator(self, N):
        return KinshipAccumulator(self, N)

    # --- dense fp64 helpers
    def eigh(self, A, vectors=True):
        A = _arr(A, np.float64)
        n = A.shape[0]
        vals = np.empty(n)
        vecs = np.empty((n, n)) if vectors else None
        self._check(self.lib.mmg_eigh_f64(self.h, _ptr(A), n, _ptr(vals), _ptr(vecs)))
        return vals, vecs          # rows of vecs are eigenvectors

    def dgemm(self, A, B, ta=False, tb=False):
        A = _arr(A, np.float64)
        B = _arr(B, np.float64)
        m, k = (A.shape[1], A.shape[0]) if ta else A.shape
        k2, n = (B.shape[1], B.shape[0]) if tb else B.shape
        assert k == k2, (A.shape, B.shape, ta, tb)
        out = np.empty((m, n))
        self._check(self.lib.mmg_dgemm_f64(self.h, int(ta), int(tb), m, n, k, _ptr(A), _ptr(B), _ptr(out)))
        return out

    # --- scan
    def scan_set_model(self, A, w, ndigits=0):
        A = _arr(A, np.float64)
        w = _arr(np.asarray(w).reshape(-1), np.float64)
        assert A.shape == (len(w), len(w))
        self._check(self.lib.mmg_scan_set_model(self.h, len(w), _ptr(A), _ptr(w), int(ndigits)))

    def scan(self, g, h0_rss, df2, fetch=True, stats=False, out=None):
        """out: optional (rss, F, p) arrays of length M to fetch into (e.g. page-locked with pin())."""
        self._check(self.lib.mmg_emmax_scan_device(self.h, g.h, float(h0_rss), int(df2)))
        if not fetch:
            return None
        rss, F, p = out if out is not None else (np.empty(g.M), np.empty(g.M), np.empty(g.M))
        self._check(self.lib.mmg_scan_fetch(self.h, g.M, _ptr(rss), _ptr(F), _ptr(p)))
        out = {"rss": rss, "f_stats": F, "ps": p}
        if stats:
            dot, den, sm = np.empty(g.M), np.empty(g.M), np.empty(g.M)
            self._check(self.lib.mmg_scan_fetch_stats(self.h, g.M, _ptr(dot), _ptr(den), _ptr(sm)))
            out.update(dot=dot, den=den, sum=sm)
        return out

    def rot(self, evecs_rows, M_cap):
        return Rot(self, evecs_rows, M_cap)

    def reml(self, K, X, y):
        return Reml(self, K, X, y)

    def perm_plan(self, H, Ys, h0_rss, centre_snps=True):
        return PermPlan(self, H, Ys, h0_rss, centre_snps)

    def scan_multi(self, rot, d, omega, G, h0_rss, df2, want=("rss", "f_stats", "ps"), out=None):
        """P phenotypes over the rotated store: d, omega [P x N], G [P x q x N], h0_rss [P] -> {'rss','f_stats','ps'}
        each [P x M] (mmg_emmax_scan_multi).  out: {'ps': array, ...} -- C-contiguous float64 buffers with at least
        P * M elements to receive the results (page-locked ones from pinned_empty() arrive at PCIe rate, and the
        download of one batch then hides behind the next batch's pass; reuse them across calls)."""
        d = _arr(d, np.float64)
        omega = _arr(omega, np.float64)
        G = _arr(G, np.float64)
        h0 = _arr(np.asarray(h0_rss).reshape(-1), np.float64)
        P, N = d.shape
        assert omega.shape == (P, N) and G.ndim == 3 and G.shape[0] == P and G.shape[2] == N and len(h0) == P
        q = G.shape[1]
        outs = {}
        for k in ("rss", "f_stats", "ps"):
            if k not in want:
                outs[k] = None
            elif out is not None and out.get(k) is not None:
                buf = out[k]
                assert buf.dtype == np.float64 and buf.flags.c_contiguous and buf.size >= P * rot.M
                outs[k] = buf.reshape(-1)[:P * rot.M].reshape(P, rot.M)
            else:
                outs[k] = np.empty((P, rot.M))
        self._check(self.lib.mmg_emmax_scan_multi(self.h, rot.h, P, q, _ptr(d), _ptr(omega), _ptr(G), _ptr(h0), int(df2),
                                                  _ptr(outs["rss"]), _ptr(outs["f_stats"]), _ptr(outs["ps"])))
        return {k: v for k, v in outs.items() if v is not None}

    def scan_last_stats(self):
        """{'adaptive', 'n_refined', 'eps_max', 'fell_back'} of the last scan (adaptive digit schedule); 'n_exact': SNPs
        recomputed from the fp64 matrix (mmg_scan_last_exact; -1: over budget)."""
        a, n, e, r, f = C.c_int32(0), C.c_int64(0), C.c_double(0.0), C.c_double(0.0), C.c_int32(0)
        self._check(self.lib.mmg_scan_last_stats(self.h, C.byref(a), C.byref(n), C.byref(e), C.byref(r), C.byref(f)))
        x = C.c_int64(0)
        self._check(self.lib.mmg_scan_last_exact(self.h, C.byref(x)))
        return {"adaptive": bool(a.value), "n_refined": n.value, "eps_max": e.value, "sigma_ratio_max": r.value,
                "fell_back": bool(f.value), "n_exact": x.value}

    def scan_deliver_begin(self, outs, count=None, comm=None):
        """Background delivery of the last scan's (rss, F, p) into the three host arrays `outs`
        (page-locked for overlap): this rank's `count` values, or the RCCL all-gather over `comm`."""
        self._check(self.lib.mmg_scan_deliver_begin(self.h, comm, int(count if count is not None else len(outs[0])),
                                                    *[_ptr(o) for o in outs]))

    def scan_deliver_wait(self):
        self._check(self.lib.mmg_scan_deliver_wait(self.h))

    def f_sf(self, F, df2):
        F = _arr(np.asarray(F).reshape(-1), np.float64)
        p = np.empty_like(F)
        self._check(self.lib.mmg_f_sf(self.h, _ptr(F), len(F), int(df2), _ptr(p)))
        return p

    def perm(self, g, H, Ys, h0_rss, ndigits=0, comm=None, after_scan_HtQ=None):
        """comm: RCCL communicator handle -- the minima then cover the SNP blocks of all ranks (reduced in HBM).
        after_scan_HtQ: [q x N] rows H'Q_c of the scan that just ran over g with the same H -- t.t is then rebuilt
        from that scan's quadratic forms (mmg_emmax_perm_after_scan) instead of a second O(N^2) pass."""
        H = _arr(H, np.float64)
        Ys = _arr(Ys, np.float64)
        P = Ys.shape[1]
        out = np.empty(P)
        if after_scan_HtQ is not None:
            U = _arr(np.atleast_2d(after_scan_HtQ), np.float64)
            assert U.shape[1] == g.N
            self._check(self.lib.mmg_emmax_perm_after_scan(self.h, comm, g.h, g.N, _ptr(H), _ptr(Ys), P, float(h0_rss),
                                                           _ptr(U), U.shape[0], _ptr(out)))
            return out
        self._check(self.lib.mmg_emmax_perm_sharded(self.h, comm, g.h, g.N, _ptr(H), _ptr(Ys), P, float(h0_rss),
                                                    int(ndigits), _ptr(out)))
        return out

    def trim(self):
        """Free what the context keeps between calls for speed (mmg_ctx_trim: the band-route factor stores, up to 2 x 2 GB)."""
        self._check(self.lib.mmg_ctx_trim(self.h))

    def close(self):
        if getattr(self, "h", None) is not None:
            self.lib.mmg_ctx_destroy(self.h)
            self.h = None


_default_ctx = None


def get_context():
    """Process-wide default context on device LOCAL_RANK (one process per GPU)."""
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(int(os.environ.get("MMG_DEVICE", os.environ.get("LOCAL_RANK", "0"))))
    return _default_ctx
