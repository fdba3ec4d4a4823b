"""Load the read-only reference (/root/reference, Python 2) into this Python 3 process.

TEST INFRASTRUCTURE ONLY.  Used by make_golden.py (fixture generation) and by the optional
`-m "not gpu"` test that re-validates the oracle against the live reference when
/root/reference is mounted.  Nothing here is shipped, nothing is copied into the repo: the
reference sources are translated on the fly into a throw-away temp dir
(SURVEY.md §8c recipe): lib2to3, `== None` -> `is None`, numpy aliases that old scipy
re-exported, stub h5py / cPickle.  mode='double' additionally rewrites the 'single' dtype
literals to 'double' (the double-promoted oracle the 1e-6 parity target is defined against).
"""
import importlib
import os
import re
import shutil
import subprocess
import sys
import tempfile
import types

REF = os.environ.get("MIXMOGAM_REFERENCE", "/root/reference")
_MODS = ["kinship", "linear_models", "simulations", "snpsdata", "analyze_gwas_results",
         "phenotypeData", "gwaResults"]
# hdf5_data.py needs a REAL h5py (load(with_hdf5=True)): in the build container that is /opt/conda/bin/python3.9
# (h5py 3.3, numpy 1.26, scipy 1.7.1), not the python3.10 the tests run under.
_HDF5_MODS = ["hdf5_data"]


def available():
    return os.path.isfile(os.path.join(REF, "linear_models.py"))


def _install_aliases():
    import numpy
    import scipy
    # under scipy < 1.9 the sub-packages are not lazy attributes: import them before the alias loop, or the loop would
    # bind scipy.linalg to numpy.linalg (no lstsq(overwrite_a=), no qr(mode=) ...)
    import scipy.linalg, scipy.stats, scipy.optimize, scipy.special  # noqa: E401,F401
    for name in dir(numpy):
        if name.startswith("_"):
            continue
        if not hasattr(scipy, name):
            try:
                setattr(scipy, name, getattr(numpy, name))
            except Exception:
                pass
    scipy.mat = numpy.asmatrix
    scipy.round_ = numpy.round
    scipy.alterdot = lambda: None
    if "h5py" not in sys.modules:
        try:
            import h5py  # noqa
        except Exception:
            sys.modules["h5py"] = types.ModuleType("h5py")
    import pickle
    sys.modules.setdefault("cPickle", pickle)
    import itertools
    if not hasattr(itertools, "izip"):      # `import itertools as it; it.izip(...)` escapes lib2to3
        itertools.izip = zip
    try:
        import matplotlib
        matplotlib.use("Agg")
    except Exception:
        pass


def load(mode="double", with_hdf5=False):
    """Return dict name->module of the translated reference.  mode in {'literal','double'}.
    with_hdf5: also hdf5_data.py (needs a real h5py); its `h5py.File(name)` calls get the mode 'a' that h5py 2 (the
    reference's) defaulted to -- h5py 3 defaults to read-only."""
    assert mode in ("literal", "double")
    if not available():
        raise RuntimeError("reference not mounted at %s" % REF)
    _MODS = globals()["_MODS"] + (_HDF5_MODS if with_hdf5 else [])
    tmp = tempfile.mkdtemp(prefix="mmg_ref_%s_" % mode)
    for m in _MODS:
        shutil.copy(os.path.join(REF, m + ".py"), os.path.join(tmp, m + ".py"))
    subprocess.run([sys.executable, "-m", "lib2to3", "-w", "-n"] +
                   [os.path.join(tmp, m + ".py") for m in _MODS],
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    for m in _MODS:
        p = os.path.join(tmp, m + ".py")
        src = open(p).read()
        src = src.replace(" == None", " is None").replace(" != None", " is not None")
        if m == "kinship":
            # sp.negative(bool) is rejected by numpy >= 1.13 (kinship.py:67)
            src = src.replace("sp.negative(sp.isnan(norm_snps_array))",
                              "~sp.isnan(norm_snps_array)")
        if m == "hdf5_data":
            src = re.sub(r"h5py\.File\((\w+)\)", r"h5py.File(\1, 'a')", src)
        if mode == "double" and m in ("kinship", "linear_models", "hdf5_data"):
            src = src.replace("'single'", "'double'")
        open(p, "w").write(src)
    _install_aliases()
    # isolate module namespace per mode
    saved = {m: sys.modules.pop(m, None) for m in _MODS}
    sys.path.insert(0, tmp)
    try:
        mods = {m: importlib.import_module(m) for m in _MODS}
    finally:
        sys.path.remove(tmp)
        for m in _MODS:
            sys.modules.pop(m, None)
            if saved[m] is not None:
                sys.modules[m] = saved[m]
        shutil.rmtree(tmp, ignore_errors=True)
    return mods
