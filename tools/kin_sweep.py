"""Job-order sweep of the FP4 IBS kinship GEMM (kinship_f4_tr_kernel) at N x M: every (MMG_KIN_PATCH, MMG_KIN_KSPLIT) pair in a
fresh process (the knobs are read once), GEMM milliseconds (hipEvents) and a digest of the counts.   python tools/kin_sweep.py N M"""
import os, subprocess, sys
N, M = sys.argv[1], sys.argv[2]
one = r'''
import hashlib, os, sys
sys.path.insert(0, %r)
from mixmogam_amd import _lib
ctx = _lib.get_context()
g = ctx.geno(M=int(sys.argv[2]), N=int(sys.argv[1])).fill_hash(1)
ms = []
for rep in range(5):
    c = ctx.kinship_ibs_counts(g)
    ms.append(ctx.kernel_ms("kinship"))
print("%%-8s ksplit %%-4s  min %%.3f ms  median %%.3f ms  digest %%s" %% (os.environ.get("MMG_KIN_PATCH", "4x8"), os.environ.get("MMG_KIN_KSPLIT", "auto"),
      min(ms[1:]), sorted(ms[1:])[2], hashlib.sha256(c.tobytes()).hexdigest()[:12]), flush=True)
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for patch in ("4x8", "8x8", "2x16", "6x6", "3x11", "20x20", "1x32"):
    for ks in ("auto", "6", "11", "17", "28"):
        env = dict(os.environ, MMG_KIN_PATCH=patch)
        if ks != "auto":
            env["MMG_KIN_KSPLIT"] = ks
        subprocess.run([sys.executable, "-c", one, N, M], env=env, check=False)
