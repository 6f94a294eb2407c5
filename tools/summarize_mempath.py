#!/usr/bin/env python3
"""Per-kernel medians of the memory-path passes of tools/mempath_counters.sh (one launch = 500 frames of the serialized stage
pass, 1000 for the matcher / back-projection) and the derived figures DESIGN.md quotes: L2 hit rate, TCP tag accesses and
TCP -> L2 read requests per keypoint (describe) or per frame (level kernels), TA busy share.
usage: summarize_mempath.py <out.json> <counter_collection.csv> ..."""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

vals = collections.defaultdict(lambda: collections.defaultdict(list))
pair_sums = {}   # partner counter of a (X, GRBM_GUI_ACTIVE) pass -> kernel -> [sum X, sum GRBM, launches]
for path in sys.argv[2:]:
    per_file = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        per_file[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    names = set(c for cs in per_file.values() for c in cs)
    if "GRBM_GUI_ACTIVE" in names and len(names) == 2:
        partner = (names - {"GRBM_GUI_ACTIVE"}).pop()
        pair_sums[partner] = {k: [sum(cs[partner]), sum(cs["GRBM_GUI_ACTIVE"]), len(cs[partner])] for k, cs in per_file.items()
                              if partner in cs and "GRBM_GUI_ACTIVE" in cs}
        continue     # these passes only feed the busy fractions
    for k, cs in per_file.items():
        for c, v in cs.items():
            vals[k][c].extend(v)
for d in pair_sums.values():
    for k in d:
        vals[k]  # (a kernel seen only in a pair pass still gets a row)
med = {k: {c: sorted(v)[len(v) // 2] for c, v in cs.items()} for k, cs in vals.items()}
KP_PER_FRAME = 1882.0  # cfg2 at the reference's min-area 1000
out = {"_meta": {"csrc_sha": bench.csrc_sha(), "units": "counter value per launch (median over launches) of "
                 "`rocprofv3 --pmc <one block, <= 2 counters> -- python3 tools/stage_times.py --reps 2`; detector kernels: "
                 "500 frames per launch", "keypoints_per_frame": KP_PER_FRAME}, "kernels": {}}
for k in sorted(vals):
    c = med.get(k, {})
    row = dict(c)
    fpl = 1000.0 if ("k_match" in k or "k_ratio" in k or "k_backproject" in k) else 500.0
    if "TCC_HIT_sum" in c and (c["TCC_HIT_sum"] + c.get("TCC_MISS_sum", 0)) > 0:
        row["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    if "TCP_TCC_READ_REQ_sum" in c:
        row["tcp_to_l2_read_req_per_frame"] = c["TCP_TCC_READ_REQ_sum"] / fpl
        row["tcp_to_l2_read_req_per_keypoint"] = c["TCP_TCC_READ_REQ_sum"] / fpl / KP_PER_FRAME
    if "TCP_TOTAL_CACHE_ACCESSES_sum" in c:
        row["tcp_tag_accesses_per_frame"] = c["TCP_TOTAL_CACHE_ACCESSES_sum"] / fpl
        row["tcp_tag_accesses_per_keypoint"] = c["TCP_TOTAL_CACHE_ACCESSES_sum"] / fpl / KP_PER_FRAME
        if c.get("TCP_TCC_READ_REQ_sum"):
            row["l1_hit_rate_reads_approx"] = 1.0 - c["TCP_TCC_READ_REQ_sum"] / max(c["TCP_TOTAL_CACHE_ACCESSES_sum"], 1.0)
    # busy fractions over the kernel's own active cycles, every launch of the pass summed (ratios of sums): GRBM_GUI_ACTIVE counts
    # per XCD and arrives summed over the 8 XCDs, TA_TA_BUSY_sum over the 256 CUs' texture addressers; a wave64 vector
    # instruction occupies its SIMD for 4 cycles, 1024 SIMDs.  Each partner counter has its own pass with GRBM_GUI_ACTIVE.
    for partner, key, scale in (("TA_TA_BUSY_sum", "ta_busy_frac", 1.0 / 256.0), ("SQ_INSTS_VALU", "valu_issue_frac", 4.0 / 1024.0)):
        if partner in pair_sums and k in pair_sums[partner] and pair_sums[partner][k][1] > 0:
            x, g, n = pair_sums[partner][k]
            row[key] = x * scale / (g / 8.0)
            row[key + "_sums"] = {"counter": x, "GRBM_GUI_ACTIVE": g, "launches": n}
    out["kernels"][k] = row
json.dump(out, open(sys.argv[1], "w"), indent=1, sort_keys=True)
for k, row in out["kernels"].items():
    if not any(x in k for x in ("k_describe", "k_orient", "k_resize_blur", "k_gray_blur", "k_fast_cells", "k_match_knn2_fp4")):
        continue
    print(k[:60])
    for c, v in sorted(row.items()):
        if not isinstance(v, dict):
            print("    %-40s %16.4g" % (c, v))
