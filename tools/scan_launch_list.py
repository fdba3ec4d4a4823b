#!/usr/bin/env python3
"""Per-launch list of the scan GEMM from a rocprofv3 --kernel-trace csv of `bench.py` (launch order, durations, labels), and
the first-pass mean written into profiles/traffic_c3.json so that bench.py's roofline.ms can be checked against a committed file.

usage: scan_launch_list.py <dir with *kernel_trace.csv> <out.txt> [traffic_c3.json]

Labels: a bench step launches the full-grid kernel once with the adaptive 3-plane schedule ('first pass'), then -- where the
schedule asks for the 4th plane -- a small-grid 'refinement' launch; the all-planes reference record afterwards launches the
full grid with four planes ('all planes', a third longer).  Full-grid launches are split by duration at 1.15 x the fastest."""
import csv
import glob
import json
import sys

root, out_path = sys.argv[1], sys.argv[2]
traffic_path = sys.argv[3] if len(sys.argv) > 3 else None
files = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)
rows = [r for f in files for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
want = ("scan_quad", "scan_finalize", "kinship_f4_tr", "kinship_f32", "kinship_grm4")
sel = [r for r in rows if any(w in r["Kernel_Name"] for w in want)]


def short(name):
    return name.replace("void ", "").replace("mmg::", "").split("(")[0][:44]


def grid(r):
    return int(r.get("Grid_Size") or r.get("Grid_Size_X") or 0) if "Grid_Size" in r else \
        int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1)) * int(r.get("Grid_Size_Z", 1))


quad = [r for r in sel if "scan_quad" in r["Kernel_Name"]]
big = max(grid(r) for r in quad)
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
fastest = min(dur(r) for r in quad if grid(r) == big)
first, allp = [], []
WARMUP, STEPS = int(__import__("os").environ.get("BENCH_WARMUP", 1)), int(__import__("os").environ.get("BENCH_STEPS", 3))
n_first = 0
timed = []
with open(out_path, "w") as f:
    f.write("scan / kinship kernels of `%s` under rocprofv3 --kernel-trace, in launch order (tools/scan_launch_list.py)\n"
            % " ".join(sys.argv[4:] or ["bench.py"]))
    for r in sel:
        label = ""
        if "scan_quad" in r["Kernel_Name"]:
            if grid(r) != big:
                label = "refinement (4th plane where asked for)"
            elif dur(r) <= 1.15 * fastest:
                n_first += 1
                if n_first <= WARMUP:
                    label = "first pass (3 planes, every SNP) -- warm-up step"
                elif n_first <= WARMUP + STEPS:
                    label = "first pass (3 planes, every SNP) -- TIMED step %d" % (n_first - WARMUP)
                    timed.append(dur(r))
                else:
                    label = "first pass (3 planes) -- outside the timed region (end-to-end calls, structured / ingest records)"
                first.append(dur(r))
            else:
                label = "all planes (reference record / warm-up of a cold clock)"
                allp.append(dur(r))
        f.write("%-46s grid %9d  %9.3f ms  %s\n" % (short(r["Kernel_Name"]), grid(r), dur(r), label))
    mean = sum(timed) / len(timed)
    f.write("\nTIMED steps (what bench.py's roofline.ms averages, there from hipEvents): %d launches, mean %.3f ms, min %.3f, max %.3f\n"
            % (len(timed), mean, min(timed), max(timed)))
    f.write("all first passes of the run: %d launches, mean %.3f ms, min %.3f, max %.3f\n" % (len(first), sum(first) / len(first), min(first), max(first)))
    if allp:
        f.write("all-plane passes: %d launches, mean %.3f ms\n" % (len(allp), sum(allp) / len(allp)))
if traffic_path:
    tj = json.load(open(traffic_path))
    k = tj["kernels"].setdefault("scan_quad_w4s_kernel", {})
    k["first_pass_ms_mean"] = mean
    k["first_pass_launches"] = len(timed)
    k["first_pass_source"] = out_path.split("/")[-1]
    json.dump(tj, open(traffic_path, "w"), indent=1)
print("timed first-pass mean %.3f ms over %d launches" % (mean, len(timed)))
