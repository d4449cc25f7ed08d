"""Per-kernel statistics (calls, total, average) from a rocprofv3 rocpd SQLite database (`*_results.db`),
for images whose rocprofv3 writes the database instead of CSV.  usage: python tools/rocpd_stats.py <db> [csv out]"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
q = """select s.kernel_name, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), max(d.end - d.start)
       from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
       group by s.kernel_name order by 3 desc"""
rows = list(con.execute(q))
tot = sum(r[2] for r in rows) or 1
lines = ['"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"']
for r in rows:
    lines.append(f'"{r[0]}",{r[1]},{r[2]},{r[3]:.1f},{100.0 * r[2] / tot:.2f},{r[4]},{r[5]}')
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write("\n".join(lines) + "\n")
for r in rows[:16]:
    print(f"{r[0][:88]:88s} {r[1]:6d} {r[2] / 1e6:10.2f} ms {r[3] / 1e3:10.1f} us {100.0 * r[2] / tot:6.2f} %")
