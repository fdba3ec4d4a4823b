"""GPU tests of round 5: the randomised sweeps 2-5 (tools only in round 4, because one of them had ended in a GPU memory fault
whose cause was unknown) and the stresses that found that cause -- the streaming driver under 1-9-SNP chunks, and two threads
in the library at once (profiles/r5_two_thread_fault_bisect.txt: a late LDS read of the kinship GEMMs overwrote an epilogue
index when another kernel shared the CU; 2 s to a fault on the library of round 4).  Each tool is a process of its own, one
at a time; all of them check against the oracle or against an undisturbed run of the same call."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_tool(name, *args, env=None, timeout=900):
    e = dict(os.environ)
    e.update(env or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", name)] + [str(a) for a in args], capture_output=True,
                         text=True, timeout=timeout, env=e)
    text = out.stdout[-4000:] + out.stderr[-3000:]
    assert "Memory access fault" not in text, text
    assert out.returncode == 0 and "failures: 0" in out.stdout, text
    return out.stdout


@pytest.mark.parametrize("tool, cases", [("random_parity2.py", 10), ("random_parity3.py", 8), ("random_parity4.py", 12),
                                         ("random_parity5.py", 10)])
def test_randomised_sweeps_against_the_oracle(tool, cases):
    """tools/random_parity2..5.py: larger N on the band route with duplicated / related individuals, the public permutation
    test, with_betas, exact EMMA, the chunked driver (2); REML and ML by both routes over kinships of different make (3);
    containers (int8, 1-bit, 2-bit rows) through run_emmax / run_emmax_perm with MAF filters, replicates (4); mlmm (5)."""
    run_tool(tool, cases, 5)


def test_streaming_driver_under_tiny_chunks_and_pool_churn():
    """tools/stress_stream.py: run_emmax / run_emmax_perm with chunks of 1-9 SNPs through the prefetching loop, N in {odd, 199,
    257, 1001}, pools re-allocated mid-run, int8 / packed chromosomes mixed in one tree, files and memory, device handles in
    cyclic garbage -- every result against the one-chunk, no-prefetch run of the same data."""
    out = run_tool("stress_stream.py", 150, 5)
    assert "150 calls" in out


@pytest.mark.parametrize("n, parts, binary", [(257, "grm,stats,scan", "0"), (256, "grm_keep", "1"), (1001, "ibs,grm", "1")])
def test_two_threads_in_the_library_at_once(n, parts, binary):
    """tools/stress_two_threads.py: an uploader thread on its own context beside GRM / IBS kinship / scan calls on the default
    one -- the library of round 4 faulted within 2 s of this (write to a wild address from kinship_i8_tr_kernel /
    kinship_grm4_kernel); results must not change from round to round and nothing may fault."""
    # (5 s per variant: the library of round 4 faulted within 2-3 s; 8 s until round 5 -- the suite's time)
    out = run_tool("stress_two_threads.py", 5, n, 1, env={"MMG_STRESS_PARTS": parts, "MMG_STRESS_BINARY": binary})
    assert "compute rounds" in out
