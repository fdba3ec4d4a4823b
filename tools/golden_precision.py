"""p-values of emmax() on the golden cases against the double-precision reference, adaptive schedule and all planes."""
import glob, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_case
from mixmogam_amd import linear_models as lm


def rel(a, b):
    a = np.asarray(a, float).ravel(); b = np.asarray(b, float).ravel()
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))


names = sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(ROOT, "tests", "golden", "*.npz")) if "extras" not in f and "fsf" not in f and "kat" not in f)
for mode in ("adaptive", "all planes"):
    if mode == "all planes":
        os.environ["MMG_SCAN_ADAPTIVE"] = "0"
    worst = 0.0
    for nm in names:
        try:
            c = load_case(nm)
        except KeyError:
            continue
        if "dbl_emmax_ps" not in c:
            continue
        res = lm.emmax(list(c["snps"]), list(c["y"]), c["dbl_ibs_scaled"], cofactors=c["cof"])
        r = rel(res["ps"], c["dbl_emmax_ps"]); worst = max(worst, r)
        print(mode, nm, "%.2e" % r)
    print(mode, "worst %.2e" % worst)
