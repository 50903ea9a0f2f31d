#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd (sqlite) kernel trace into the --stats style table: per kernel calls, total, avg, min, max
(durations in ms) plus register / LDS / scratch usage.   usage: rocpd_stats.py results.db [out.csv]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
t = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [x for x in t if 'kernel_dispatch' in x][0]
ks = [x for x in t if 'kernel_symbol' in x][0]
rows = db.execute(f"""select s.kernel_name, count(*), sum(d.end-d.start)/1e6, avg(d.end-d.start)/1e6, min(d.end-d.start)/1e6, max(d.end-d.start)/1e6,
                      s.arch_vgpr_count, s.accum_vgpr_count, s.sgpr_count, max(d.group_segment_size), max(d.private_segment_size), max(d.workgroup_size_x)
                      from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc""").fetchall()
total = sum(r[2] for r in rows)
lines = ["kernel,calls,total_ms,avg_ms,min_ms,max_ms,percent,arch_vgpr,accum_vgpr,sgpr,lds_bytes,scratch_bytes_per_lane,workgroup"]
for r in rows:
    name = r[0].replace('wtk::', '').split('(')[0][-60:]
    lines.append(f"{name},{r[1]},{r[2]:.3f},{r[3]:.4f},{r[4]:.4f},{r[5]:.4f},{100*r[2]/total:.2f},{r[6]},{r[7]},{r[8]},{r[9]},{r[10]},{r[11]}")
out = "\n".join(lines)
print(out)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out + "\n")

# optional per-dispatch dump (launch order): rocpd_stats.py results.db out.csv dispatches.csv
if len(sys.argv) > 3:
    rows = db.execute(f"""select s.kernel_name, d.start, d.end-d.start, d.grid_size_x, d.workgroup_size_x from {kd} d join {ks} s on d.kernel_id=s.id
                          order by d.start""").fetchall()
    t0 = rows[0][1] if rows else 0
    with open(sys.argv[3], "w") as f:
        f.write("kernel,start_ms,dur_ms,grid,workgroup\n")
        for r in rows:
            name = r[0].replace("(anonymous namespace)::", "").split("(")[0][-40:]
            f.write(f"{name},{(r[1]-t0)/1e6:.3f},{r[2]/1e6:.4f},{r[3]},{r[4]}\n")
