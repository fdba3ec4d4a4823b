"""The first-generation 8-wave kernels of the permutation / rotation / kinship GEMMs stay in the library behind
MMG_{PERM,ROT,KIN}_KERNEL=w8 for A/B timing; everything they compute is integer-exact, so the 4-wave job-stream
kernels (gemm_i8_w4s.h) that replaced them must give the SAME BITS -- as must the 64-bit form of the digit-pair
epilogue (MMG_W4_SLOW_EPI).  The switches are read once per process: each variant runs in its own interpreter."""
import hashlib
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu

SCRIPT = textwrap.dedent("""
    import hashlib, sys
    import numpy as np
    sys.path.insert(0, %(root)r)
    from mixmogam_amd import _lib
    ctx = _lib.get_context()
    rng = np.random.RandomState(5)
    h = hashlib.sha256()
    for n, m, hi in [(300, 2500, 2), (257, 700, 3), (1100, 3000, 2)]:
        snps = rng.randint(0, hi, size=(m, n)).astype(np.int8)
        g = ctx.geno(snps)
        h.update(ctx.kinship_ibs_counts(g).tobytes())
        Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
        rot = ctx.rot(np.ascontiguousarray(Q.T), m).load(g)
        h.update(rot.fetch().tobytes())
        rot.close()
        H = rng.standard_normal((n, n)) / np.sqrt(n)
        Ys = rng.standard_normal((n, 130))
        h.update(np.asarray(ctx.perm(g, H, Ys, float(n))).tobytes())
        g.close()
    # exact GRM of a binary store with >= 2^16 SNPs: four weight planes, all of them in one pass (gemm_i8_grm4.h)
    snps = (rng.random_sample((70000, 390)) < rng.uniform(0.1, 0.9, size=(70000, 1))).astype(np.int8)
    g = ctx.geno(snps[snps.std(1) > 0])
    acc = ctx.kinship_accumulator(390)
    acc.add_grm(g)
    print("GRMDIGEST", hashlib.sha256(acc.fetch()[0].tobytes()).hexdigest())
    acc.close(); g.close()
    print("DIGEST", h.hexdigest())
""")


def _digest(tmp_path, extra_env, key="DIGEST"):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "gen.py"
    script.write_text(SCRIPT % {"root": root})
    env = dict(os.environ)
    env.update(extra_env)
    out = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith(key + " ")]
    assert lines, out.stdout + out.stderr
    return lines[-1].split()[1]


def test_four_wave_kernels_give_the_bits_of_the_eight_wave_generation(tmp_path):
    new = _digest(tmp_path, {})
    old = _digest(tmp_path, {"MMG_PERM_KERNEL": "w8", "MMG_ROT_KERNEL": "w8", "MMG_KIN_KERNEL": "w8"})
    slow = _digest(tmp_path, {"MMG_W4_SLOW_EPI": "1"})
    gv1 = _digest(tmp_path, {"MMG_PERM_GV": "1", "MMG_ROT_GV": "8"})
    # round 3: the default IBS kinship reads the SNP-major store through transposed LDS reads (gemm_i8_w4tr.h);
    # w4 / w8 are the two generations on the individual-major image
    kin_w4 = _digest(tmp_path, {"MMG_KIN_KERNEL": "w4"})
    # binary stores take the FP4-operand kinship GEMM by default (exact integers in fp32 accumulators);
    # MMG_KIN_FP4=0 is the int8 transposed-read kernel (the 0/1/2 case of the script runs it in every variant)
    kin_i8 = _digest(tmp_path, {"MMG_KIN_FP4": "0"})
    assert new == old == slow == gv1 == kin_w4 == kin_i8


def test_fused_grm_planes_give_the_bits_of_one_gemm_per_plane(tmp_path):
    """Exact GRM of a binary store with four weight planes: all planes in one pass over the plain genotype tiles
    (gemm_i8_grm4.h, digit scaling as a byte mask in registers) against one transposed-read GEMM per plane over
    digit-scaled images (MMG_GRM_FUSED=0) -- the same integer planes, the same fp64 rank-one terms, the same bits.
    (The individual-major generations sum the rank-one terms in another order and are not part of this comparison.)"""
    assert _digest(tmp_path, {}, "GRMDIGEST") == _digest(tmp_path, {"MMG_GRM_FUSED": "0"}, "GRMDIGEST")
