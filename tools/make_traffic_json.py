#!/usr/bin/env python3
"""Builds profiles/<tag>_pmc_traffic.json from the PMC summaries of tools/profile_round.sh:
   HBM-side bytes per kernel = FETCH_SIZE * kf + WRITE_SIZE * kw   (rocprofv3 reports both in KiB),
with kf, kw calibrated on a streaming copy of known size with the same access width (k_calib_copy: one dword per lane), as
MI355X_MICROARCH.md §HBM prescribes for access widths other than its 16 B/lane reference (where FETCH_SIZE reads 1/2).
usage: make_traffic_json.py <gpurun_out dir> <tag> <out.json> <res> <scene> <rounds_with_work_per_step> <batches_per_step>"""
import csv
import json
import sys

d, tag, out, res, scene, rounds, batches = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5], float(sys.argv[6]), float(sys.argv[7])


def table(path, col):
    r = {}
    rows = list(csv.reader(open(path)))
    ci = rows[0].index(col + "_sum")
    for row in rows[1:]:
        if row and row[0].startswith("k_"):          # our kernels only (library kernel names contain commas)
            r[row[0]] = (float(row[ci]), int(row[1]))
    return r


fetch = table(f"{d}/{tag}_pmc_FETCH_SIZE.csv", "FETCH_SIZE")
write = table(f"{d}/{tag}_pmc_WRITE_SIZE.csv", "WRITE_SIZE")
cal_f = table(f"{d}/{tag}_calib_FETCH_SIZE.csv", "FETCH_SIZE")["k_calib_copy"]
cal_w = table(f"{d}/{tag}_calib_WRITE_SIZE.csv", "WRITE_SIZE")["k_calib_copy"]
calib = json.load(open(f"{d}/{tag}_calib.json"))
known = calib["n_dwords"] * 4.0 * calib["repeats"]           # bytes read == bytes written
kf = known / (cal_f[0] * 1024.0)
kw = known / (cal_w[0] * 1024.0)
kernels = {}
for k in fetch:
    if not k.startswith("k_") or k == "k_calib_copy":
        continue
    fb = fetch[k][0] * 1024.0 * kf
    wb = write.get(k, (0.0, 0))[0] * 1024.0 * kw
    kernels[k] = {"fetch_bytes_per_step": fb, "write_bytes_per_step": wb, "dispatches": fetch[k][1],
                  "hbm_bytes_per_launch": (fb + wb) / max(1, fetch[k][1])}
json.dump({"workload": {"scene": scene, "res": res, "steps": 1}, "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py --steps 1 --warmup 0",
           "calibration": {"known_bytes_each_way": known, "FETCH_SIZE_KiB": cal_f[0], "WRITE_SIZE_KiB": cal_w[0], "fetch_factor": kf, "write_factor": kw,
                           "note": "factor = true bytes / (counter * 1024) for a one-dword-per-lane coalesced streaming copy"},
           "note": "per launch = per step / dispatches of that kernel in one step (kMaxWalkIters rounds per batch for the round kernels, most of them empty)",
           "kernels": kernels}, open(out, "w"), indent=1)
# ... and into the calibration record itself, next to the copy's size (profiles/<tag>_calib.json)
import os
cal_out = os.path.join(os.path.dirname(os.path.abspath(out)), f"{tag}_calib.json")
json.dump(dict(calib, known_bytes_each_way=known, FETCH_SIZE_KiB=cal_f[0], WRITE_SIZE_KiB=cal_w[0], fetch_factor=kf, write_factor=kw,
               note="k_calib_copy: streaming copy, one dword per lane; factor = true bytes / (counter * 1024)"), open(cal_out, "w"))
print(json.dumps({"fetch_factor": kf, "write_factor": kw, "total_GB_per_step": sum(v["fetch_bytes_per_step"] + v["write_bytes_per_step"] for v in kernels.values()) / 1e9}))
