#!/usr/bin/env python3
"""Prints DESIGN.md §4's kernel table from the committed summaries of a round (profiles/<tag>_*):
   excl. ms / pass = <tag>_kernel_stats_streams1.csv total ÷ passes (three 2-spp steps = 6 passes),
   lanes = SQ_THREAD_CYCLES_VALU ÷ SQ_ACTIVE_INST_VALU, wait = SQ_WAIT_ANY ÷ SQ_WAVE_CYCLES, VALU = SQ_ACTIVE_INST_VALU × 4 ÷ (GRBM_GUI_ACTIVE ÷ 8 × 1024),
   L1 hit = 1 − TCP_TCC_READ_REQ ÷ TCP_TOTAL_CACHE_ACCESSES, L2 hit = TCC_HIT ÷ TCC_REQ, latency = TCP_TCC_READ_REQ_LATENCY ÷ TCP_TCC_READ_REQ,
   VGPR / spilled / scratch from <tag>_kernel_resources.txt, GB per step from <tag>_pmc_traffic.json.
usage: tools/kernel_table.py [tag] [passes]"""
import csv
import json
import os
import re
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
passes = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")


def short(name):
    m = re.match(r"_ZN3wtk\d+(k_[a-z_0-9]+?)ENS_", name)
    return m.group(1) if m else name


def rows(path):
    with open(path) as f:
        r = list(csv.reader(f))
    return {short(x[0]): dict(zip(r[0], x)) for x in r[1:] if x}


st = rows(f"{P}/{tag}_kernel_stats_streams1.csv")
sq = rows(f"{P}/{tag}_pmc_SQ_lane_utilisation.csv")
ta = rows(f"{P}/{tag}_pmc_TA.csv")
tcp = rows(f"{P}/{tag}_pmc_TCP.csv")
tcc = rows(f"{P}/{tag}_pmc_TCC.csv")
tr = json.load(open(f"{P}/{tag}_pmc_traffic.json"))["kernels"]
res = {}
for line in open(f"{P}/{tag}_kernel_resources.txt"):
    m = re.match(r"(k_\S+)\s+vgpr\s+(\d+)\s+spill\s+(\d+)\s+sgpr_spill\s+\d+\s+lds\s+\d+\s+scratch\s+(\d+)", line)
    if m:
        res[m.group(1)] = (m.group(2), m.group(3), m.group(4))

f = lambda d, k: (float(d[k]) or float("nan")) if d and k in d else float("nan")
print("| Kernel | excl. ms / pass | lanes of 64 | wait % | VALU % | L1 / L2 hit, latency | VGPR / spilled / scratch B | GB per step |")
print("|---|---|---|---|---|---|---|---|")
tot_ms = tot_gb = 0.0
for k in sorted((k for k in st if k.startswith("k_")), key=lambda k: -float(st[k]["total_ms"])):
    ms = float(st[k]["total_ms"]) / passes
    s, a, c1, c2 = sq.get(k), ta.get(k), tcp.get(k), tcc.get(k)
    lanes = f(s, "SQ_THREAD_CYCLES_VALU_sum") / f(s, "SQ_ACTIVE_INST_VALU_sum") if s else float("nan")
    wait = 100 * f(s, "SQ_WAIT_ANY_sum") / f(s, "SQ_WAVE_CYCLES_sum") if s else float("nan")
    valu = 100 * f(s, "SQ_ACTIVE_INST_VALU_sum") * 4 / (f(a, "GRBM_GUI_ACTIVE_sum") / 8 * 1024) if s and a else float("nan")
    l1 = 1 - f(c1, "TCP_TCC_READ_REQ_sum_sum") / f(c1, "TCP_TOTAL_CACHE_ACCESSES_sum_sum") if c1 else float("nan")
    lat = f(c1, "TCP_TCC_READ_REQ_LATENCY_sum_sum") / f(c1, "TCP_TCC_READ_REQ_sum_sum") if c1 else float("nan")
    l2 = f(c2, "TCC_HIT_sum_sum") / f(c2, "TCC_REQ_sum_sum") if c2 else float("nan")
    gb = (tr[k]["fetch_bytes_per_step"] + tr[k]["write_bytes_per_step"]) / 1e9 if k in tr else float("nan")
    r = res.get(k, ("?", "?", "?"))
    tot_ms += ms
    tot_gb += gb if gb == gb else 0.0
    print(f"| `{k}` | {ms:.1f} | {lanes:.1f} | {wait:.0f} | {valu:.0f} | {l1:.2f} / {l2:.2f}, {lat:.0f} | {r[0]} / {r[1]} / {r[2]} | {gb:.1f} |")
print(f"| **sum** | **{tot_ms:.1f}** | | | | | | **{tot_gb:.1f}** |")
