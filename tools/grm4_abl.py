#!/usr/bin/env python3
"""kinship_grm4_kernel under its timing ablations (MMG_GRM4_ABL, WRONG results): what the DMA, the LDS reads and the digit
scaling each cost.  One process per setting, on the `make EXPERIMENTS=1` library (the shipped one ignores MMG_GRM4_ABL since
round 5).   python tools/grm4_abl.py [N] [M]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    from mixmogam_amd import _lib
    n, m = int(sys.argv[2]), int(sys.argv[3])
    ctx = _lib.get_context()
    g = ctx.geno(M=m, N=n).fill_hash(20240)
    acc = ctx.kinship_accumulator(n)
    ms = []
    for _ in range(4):
        acc.add_grm(g)
        ms.append(ctx.kernel_ms("grm"))
    print("  MMG_GRM4_ABL=%s: %.2f ms (best of %s)" % (os.environ.get("MMG_GRM4_ABL", "0"), min(ms[1:]), ["%.2f" % x for x in ms[1:]]), flush=True)
    sys.exit(0)
n = sys.argv[1] if len(sys.argv) > 1 else "5000"
m = sys.argv[2] if len(sys.argv) > 2 else "1000000"
names = {"0": "as shipped", "1": "no DMA in the loop", "2": "no LDS reads, no scaling", "3": "no barrier in the loop",
         "6": "half of the scaling VALU work, operand statistics unchanged (second A block = first one's scaled fragments)",
         "7": "row-strip layout WITHOUT its digit reads from LDS",
         "4": "half of the scaling VALU work (one A fragment instead of two)", "5": "no scaling VALU work (LDS reads kept)"}
for abl in (sys.argv[3].split(",") if len(sys.argv) > 3 else ("0", "6", "4", "5", "2", "1", "3")):
    print("%s:" % names[abl], flush=True)
    subprocess.run([sys.executable, os.path.abspath(__file__), "--child", n, m], env=dict(os.environ, MMG_GRM4_ABL=abl, MMG_LIB=os.path.join(ROOT, "mixmogam_amd", "lib", "libmixmogam_hip_exp.so")), check=False)
