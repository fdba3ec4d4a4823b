"""Timing ablations of kinship_f4_tr_kernel (WRONG results; `make EXPERIMENTS=1` library only): what a K step costs without its
LDS-DMA, without its fragment reads, without either, and without the epilogue's atomics.   python tools/kin_abl.py N M"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
one = r'''
import os, sys
sys.path.insert(0, %r)
from mixmogam_amd import _lib
ctx = _lib.get_context()
g = ctx.geno(M=int(sys.argv[2]), N=int(sys.argv[1])).fill_hash(1)
ms = []
for rep in range(5):
    ctx.kinship_ibs_counts(g)
    ms.append(ctx.kernel_ms("kinship"))
names = {"0": "production", "1": "no LDS-DMA in the loop", "2": "no fragment reads in the loop", "3": "neither (MFMA + barrier + waits)", "4": "epilogue: one atomic per lane",
         "12": "N3 = 12 DMA pieces right after the barrier", "16": "N3 = 16", "101": "L2 prefetch one stage ahead", "102": "L2 prefetch two stages ahead", "113": "N3 = 12 + prefetch 1"}
print("MMG_F4_ABL=%%s %%-36s min %%.3f ms  median %%.3f ms" %% (os.environ.get("MMG_F4_ABL", "0"), names[os.environ.get("MMG_F4_ABL", "0")], min(ms[1:]), sorted(ms[1:])[2]), flush=True)
''' % ROOT
for abl in (sys.argv[3].split(",") if len(sys.argv) > 3 else "0 1 2 3 4".split()):
    env = dict(os.environ, MMG_F4_ABL=abl, MMG_LIB=os.path.join(ROOT, "mixmogam_amd", "lib", "libmixmogam_hip_exp.so"))
    subprocess.run([sys.executable, "-c", one, sys.argv[1], sys.argv[2]], env=env, check=False)
