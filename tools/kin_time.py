"""Kernel time of the IBS kinship GEMM (tools/kin_time.py N M): A/B runs of the kinship kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mixmogam_amd import _lib
N, M = int(sys.argv[1]), int(sys.argv[2])
ctx = _lib.get_context()
g = ctx.geno(M=M, N=N).fill_hash(1)
for rep in range(3):
    ctx.kinship_ibs_counts(g)
    print("kinship GEMM %.3f ms, transposition %.3f ms" % (ctx.kernel_ms("kinship"), ctx.kernel_ms("pack")), flush=True)
if hasattr(ctx, "kinship_indicator_counts"):
    for rep in range(3):
        ctx.kinship_indicator_counts(g, 1)
        print("indicator (0/1 operands) GEMM %.3f ms" % ctx.kernel_ms("kinship"), flush=True)
