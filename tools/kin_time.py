"""Kernel time of the kinship GEMMs (tools/kin_time.py N M): A/B runs of the kinship kernels.  Prints a digest of the
counts so that variants (MMG_KIN_KERNEL=w4|w8, MMG_KIN_N3=8|12|16) can be checked for identical bits."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mixmogam_amd import _lib
N, M = int(sys.argv[1]), int(sys.argv[2])
ctx = _lib.get_context()
g = ctx.geno(M=M, N=N).fill_hash(1)


def pack_ms():
    try:
        return ctx.kernel_ms("pack")
    except _lib.MixmogamHipError:
        return 0.0


ms = []
for rep in range(4):
    c = ctx.kinship_ibs_counts(g)
    ms.append(ctx.kernel_ms("kinship"))
print("env %s: IBS GEMM ms %s, image pass %.3f ms, digest %s" % (
    {k: v for k, v in os.environ.items() if k.startswith("MMG_KIN")}, ["%.3f" % x for x in ms], pack_ms(),
    hashlib.sha256(c.tobytes()).hexdigest()[:16]), flush=True)
acc = ctx.kinship_accumulator(N)
for rep in range(3):
    acc.add_grm(g)
    print("  exact GRM: digit-plane GEMMs %.3f ms, image pass %.3f ms" % (ctx.kernel_ms("grm"), pack_ms()), flush=True)
k, _ = acc.fetch()
print("  GRM digest", hashlib.sha256(k.tobytes()).hexdigest()[:16])
