#!/usr/bin/env python3
"""Where a K step of kinship_grm4_kernel spends its cycles: the stamp variants of the `make EXPERIMENTS=1` library
(MMG_GRM4_ABL=8..11, gemm_i8_grm4.h), one process each, differenced into a per-slice table.  The variants compute the right
result: the digest of the kinship they leave is compared with the shipped kernel's.
    python tools/grm4_stamps.py [N] [M] [quad]"""
import hashlib, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXP = os.path.join(ROOT, "mixmogam_amd", "lib", "libmixmogam_hip_exp.so")
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    from mixmogam_amd import _lib
    n, m = int(sys.argv[2]), int(sys.argv[3])
    ctx = _lib.get_context()
    g = ctx.geno(M=m, N=n).fill_hash(20240)
    ms = []
    for _ in range(3):
        acc = ctx.kinship_accumulator(n)
        acc.add_grm(g)
        ms.append(ctx.kernel_ms("grm"))
        K, cnt = acc.fetch()
        acc.close()
    print("RESULT ms=%.3f digest=%s" % (min(ms[1:]), hashlib.sha1(K.tobytes()).hexdigest()[:16]), flush=True)
    sys.exit(0)
n = sys.argv[1] if len(sys.argv) > 1 else "5000"
m = sys.argv[2] if len(sys.argv) > 2 else "1000000"
rows = {}
jit = not (len(sys.argv) > 3 and sys.argv[3] == "quad")   # the shipped kernel (stamp variants 12..15); quad: rounds 3-4 (8..11)
variants = ("0", "12", "13", "14", "15") if jit else ("0", "8", "9", "10", "11")
for abl in variants:
    env = dict(os.environ, MMG_GRM4_ABL=abl, MMG_LIB=EXP)
    if not jit:
        env["MMG_GRM4_LAYOUT"] = "quad"
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", n, m], env=env, capture_output=True, text=True)
    res = re.search(r"RESULT ms=([\d.]+) digest=(\w+)", out.stdout)
    if not res:
        print(out.stdout[-2000:], out.stderr[-2000:])
        sys.exit(1)
    st = re.findall(r"\[grm4 stamps\] position (\d)  waves (\d+)  K steps / wave (\d+)  cycles per K step: start -> position ([\d.]+)  "
                    r"position -> end ([\d.]+)  whole loop ([\d.]+)   \(start -> position by wave: ([\d. ]+)\)", out.stderr)
    rows[abl] = (float(res.group(1)), res.group(2), st[-1] if st else None)
    print("MMG_GRM4_ABL=%-2s  GEMM kernels %.2f ms  kinship digest %s  %s" % (abl, rows[abl][0], rows[abl][1], st[-1] if st else ""), flush=True)
same = len({r[1] for r in rows.values()}) == 1
print("digests equal: %s" % same)
cum = [0.0] + [float(rows[a][2][3]) for a in variants[1:]]
step = [float(rows[a][2][3]) + float(rows[a][2][4]) for a in variants[1:]]
names = ["1st slice of the step: MFMAs of k-slice 0 (+ the loop latch in front of it)", "2nd slice: MFMAs of k-slice 1",
         "s_waitcnt vmcnt(0) lgkmcnt(0) + s_barrier", "3rd slice: MFMAs of k-slice 2 + this wave's 4 P pieces of stage t + 2",
         "4th slice: MFMAs of k-slice 3 + 4 Q pieces (wave 0: + the digits)"]
if jit:
    names = ["1st slice: MFMAs of k-slice 0 + 4 Q pieces of stage t + 1 (wave 0: + the digits)", "2nd slice: MFMAs of k-slice 1",
             "3rd slice: MFMAs of k-slice 2, then s_waitcnt vmcnt(0) lgkmcnt(0) + s_barrier", "",
             "4th slice: MFMAs of k-slice 3 + 4 P pieces of stage t + 2"]
whole = sum(step) / len(step)
print("\ncycles per K step (64 MFMA = 2048 cycles of matrix-pipe issue), mean over all waves; stamp overhead included:")
if jit:                                                   # positions 0, 1, 2; the variant of position 3 stamps the end of the step only
    for k in range(3):
        print("  %-82s %7.1f" % (names[k], cum[k + 1] - cum[k]))
    print("  %-82s %7.1f" % (names[4], float(rows["14"][2][4])))
    print("  %-82s %7.1f" % ("K step with one stamp per step (the least disturbed)", float(rows["15"][2][4])))
else:
    for k in range(4):
        print("  %-75s %7.1f" % (names[k], cum[k + 1] - cum[k]))
    print("  %-75s %7.1f" % (names[4], whole - cum[4]))
print("  %-75s %7.1f   (per variant: %s)" % ("K step", whole, " ".join("%.0f" % s for s in step)))
sys.exit(0 if same else 1)
