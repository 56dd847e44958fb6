#!/usr/bin/env python3
"""tools/summarize_prof.py -- turn the rocprofv3 CSVs of tools/profile.sh into the small summaries
kept under profiles/ (run here after gpurun merged gpurun_out/prof back).

    python tools/summarize_prof.py [tag]              # default r02 (reads gpurun_out/prof_<tag>/)

Writes
    profiles/<tag>_kernel_stats.csv     per-kernel durations (copy of rocprofv3 --stats)
    profiles/<tag>_pmc_per_launch.json  every collected counter, mean per dispatch and kernel
    profiles/<tag>_hbm_traffic.json     HBM bytes per launch of the solve kernel (bench.py reads the
                                        newest one of the profiled workload)
    profiles/<tag>_bench_n1.json        the bench line of the profiled command
Counter handling follows MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are collected in
separate --pmc passes, are in KiB, and FETCH_SIZE counts half the bytes on gfx950 (x2).
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
P = os.path.join(REPO, "gpurun_out", "prof_" + tag)
out = os.path.join(REPO, "profiles")
os.makedirs(out, exist_ok=True)


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0]


shutil.copy(os.path.join(P, "kt", "r1_kernel_stats.csv"), os.path.join(out, f"{tag}_kernel_stats.csv"))

per = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(P, "pmc_*", "r1_counter_collection.csv"))):
    acc = defaultdict(float)      # (dispatch, kernel, counter) -> summed over instances
    for row in csv.DictReader(open(f)):
        acc[(row["Dispatch_Id"], short(row["Kernel_Name"]), row["Counter_Name"])] += float(row["Counter_Value"])
    for (_, k, c), v in acc.items():
        if k.startswith("gik::"):
            per[k][c].append(v)
summary = {k: {c: sum(v) / len(v) for c, v in sorted(cs.items())} for k, cs in sorted(per.items())}
for k, cs in summary.items():
    cs["_dispatches"] = len(next(iter(per[k].values())))
json.dump(summary, open(os.path.join(out, f"{tag}_pmc_per_launch.json"), "w"), indent=1)

solve = next(k for k in summary if "rtr_wave_kernel" in k or "rtr_block_kernel" in k or "rtr_npt_kernel" in k or "rtr_quad_kernel" in k)
fetch = summary[solve]["FETCH_SIZE"] * 1024 * 2
write = summary[solve]["WRITE_SIZE"] * 1024
prep = next((k for k in summary if "prep_quad_kernel" in k or "prep_wave_kernel" in k or "prep_block_kernel" in k), None)
prep_entry = None
if prep and "FETCH_SIZE" in summary[prep]:
    pf, pw = summary[prep]["FETCH_SIZE"] * 1024 * 2, summary[prep]["WRITE_SIZE"] * 1024
    prep_entry = {"kernel": prep, "bytes_per_launch": pf + pw, "fetch_bytes_corrected": pf, "write_bytes": pw,
                  "lds_bank_conflict_ratio": (summary[prep]["SQ_LDS_BANK_CONFLICT"] / summary[prep]["SQ_LDS_IDX_ACTIVE"]
                                              if summary[prep].get("SQ_LDS_IDX_ACTIVE") else None)}
try:
    digest = open(os.path.join(P, "source_digest.txt")).read().strip()
except OSError:
    digest = None
json.dump({"kernel": solve, "bytes_per_launch": fetch + write, "fetch_bytes_corrected": fetch,
           "write_bytes": write, "prepare": prep_entry,
           # graphik_amd.build.source_digest() of the library these counters were read from (tools/profile.sh): bench.py
           # prints "traffic_stale": true when the loaded library's differs
           "source_digest": digest,
           "how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, "
                  "tools/profile.sh %s), bench.py --steps 2 --warmup 1, mean over the dispatches; " % tag +
                  "FETCH_SIZE (KiB) x1024 x2 (gfx950 half-count correction, MI355X_MICROARCH.md HBM "
                  "section), WRITE_SIZE (KiB) x1024"},
          open(os.path.join(out, f"{tag}_hbm_traffic.json"), "w"), indent=1)

for line in open(os.path.join(P, "kt_bench.json")):
    if line.startswith("{"):
        open(os.path.join(out, f"{tag}_bench_n1.json"), "w").write(line)

print(open(os.path.join(out, f"{tag}_kernel_stats.csv")).read())
s = summary[solve]
if "SQ_INSTS_VALU" in s:
    b = json.loads(open(os.path.join(out, f"{tag}_bench_n1.json")).read())
    hv = b["hv_products"].get("executed_per_gpu", b["hv_products"]["total_per_gpu"])
    print("per tCG iteration: VALU %.1f  LDS %.1f  MFMA %.2f  SALU %.1f  wave cycles %.0f  (hv %d)" % (
        s["SQ_INSTS_VALU"] / hv, s["SQ_INSTS_LDS"] / hv, s.get("SQ_INSTS_MFMA", 0) / hv,
        s["SQ_INSTS_SALU"] / hv, s["SQ_WAVE_CYCLES"] / hv, hv))
    # SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* count quad-cycles; per product and WAVEFRONT (a problem = 1, 2 or 8 of them)
    wpp = 8 if "rtr_block" in solve else (2 if "rtr_npt_kernel<1, 1, 2>" in solve or "rtr_npt_kernel<4, 1, 2>" in solve else 1)
    if "rtr_quad_kernel" in solve:
        print("four problems per wavefront: %.0f wave cycles and %.0f VALU-active cycles per product of ONE problem "
              "(a wavefront's tCG step serves up to four)" % (4 * s["SQ_WAVE_CYCLES"] / hv, 4 * s["SQ_ACTIVE_INST_VALU"] / hv))
    else:
        print("wavefronts per problem %d: %.0f wave cycles per product and wavefront (batch average, co-resident problems included), "
              "VALU-active %.0f" % (wpp, 4 * s["SQ_WAVE_CYCLES"] / hv / wpp, 4 * s["SQ_ACTIVE_INST_VALU"] / hv / wpp))
    print("VALU active / wave cycles %.2f   LDS bank conflict / LDS active %.2f" % (
        s["SQ_ACTIVE_INST_VALU"] / s["SQ_WAVE_CYCLES"],
        s["SQ_LDS_BANK_CONFLICT"] / max(s.get("SQ_LDS_IDX_ACTIVE", 1), 1)))
print("HBM traffic per launch: %.2f MB (fetch %.2f, write %.2f)" % ((fetch + write) / 1e6, fetch / 1e6, write / 1e6))
