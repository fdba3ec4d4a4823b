#!/usr/bin/env python3
"""profiles/traffic_c3.json from the FETCH_SIZE / WRITE_SIZE passes of tools/gpu_profile.sh.
usage: make_traffic_json.py <gpurun_out/prof_TAG> <tag> [n m digits]
HBM-side bytes per launch = 2 * FETCH_SIZE (gfx950 reports half of wide coalesced reads,
MI355X_MICROARCH.md HBM section) + WRITE_SIZE; counter unit KB; mean over the dispatches of a kernel."""
import collections, csv, glob, json, os, sys
root, tag = sys.argv[1], sys.argv[2]
n, m, d = (int(x) for x in sys.argv[3:6]) if len(sys.argv) > 5 else (5000, 1000000, 4)
vals = collections.defaultdict(lambda: collections.defaultdict(list))
rows = [r for f in glob.glob(root + "/pmc_*SIZE*/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f))]
name = lambda r: r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").replace("mmg::", "")
# a kernel may be launched with different grids in one step (adaptive scan: the full pass and the small
# refinement pass): "per launch" means the launches with the largest grid
biggest = collections.defaultdict(int)
for r in rows:
    biggest[name(r)] = max(biggest[name(r)], int(r["Grid_Size"]))
adaptive = len(sys.argv) > 6 and sys.argv[6] == "adaptive"
# adaptive bench runs launch the scan GEMM first with the 3-plane schedule (warm-up + timed steps) and afterwards, for
# the all-planes reference record, with all four planes (same grid): split the full-grid dispatches in dispatch
# order -- first half adaptive (the timed region), second half all planes
per_counter_seq = collections.defaultdict(list)
for r in rows:
    if int(r["Grid_Size"]) == biggest[name(r)]:
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        per_counter_seq[(name(r), r["Counter_Name"])].append((int(r["Dispatch_Id"]), float(r["Counter_Value"]), dur))
for (k, c), seq in per_counter_seq.items():
    seq.sort()
    if adaptive and k.startswith("scan_quad") and len(seq) >= 2:
        # one run launches the scan GEMM with the 3-plane schedule (timed steps, end-to-end emmax calls) and with all four
        # planes (the all-planes reference record), same grid: the all-plane launches take a third longer
        fastest = min(d for _, _, d in seq)
        vals[k][c] = [x for _, x, d in seq if d <= 1.15 * fastest]
        allp = [x for _, x, d in seq if d > 1.15 * fastest]
        if allp:
            vals[k + "_all_planes"][c] = allp
    else:
        vals[k][c] = [x for _, x, _d in seq]
out = {"config": {"n": n, "m": m, "digits": d}, "adaptive": adaptive,
       "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes on bench.py (tools/gpu_profile.sh %s); "
               "FETCH_SIZE doubled (gfx950 reports half of wide coalesced reads, MI355X_MICROARCH.md HBM section); "
               "WRITE_SIZE as is; counter unit KB" % tag,
       "kernels": {}}
for k, c in sorted(vals.items()):
    if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        continue
    fs = sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"])
    ws = sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"])
    out["kernels"][k] = {"FETCH_SIZE_KB": fs, "WRITE_SIZE_KB": ws, "hbm_bytes_corrected": (2 * fs + ws) * 1024.0}
json.dump(out, sys.stdout, indent=1)
