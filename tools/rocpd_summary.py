"""Turn a rocprofv3 rocpd database (<out>_results.db) into small CSV summaries.
usage: python tools/rocpd_summary.py stats <db> > kernel_stats.csv      # per (kernel, grid) durations
       python tools/rocpd_summary.py pmc <db> [kernel-substring]         # per (kernel, grid, counter) means
       python tools/rocpd_summary.py timeline <db> [last-n]              # the last n launches in order: start, duration, gap
Launches of one kernel at different grid sizes are kept apart so that the bench's full-size launches
are not averaged with the small side configurations that run in the same process."""
import math
import sqlite3
import sys


def short(name):
    return name.split("(")[0].replace("void ", "")


def main():
    mode, db = sys.argv[1], sys.argv[2]
    con = sqlite3.connect(db)
    if mode == "stats":
        # a persistent kernel has one grid for every problem size: launches are also split by decade of duration
        groups = {}
        for n, g, wg, st, en in con.execute("select name, grid_x*grid_y*grid_z, workgroup_x*workgroup_y*workgroup_z, start, end from kernels"):
            d = (en - st) / 1e3
            groups.setdefault((short(n), g, wg, int(math.floor(math.log10(max(d, 1e-3))))), []).append(d)
        tot = sum(sum(v) for v in groups.values()) or 1
        print("kernel,grid_threads,workgroup,duration_decade_us,calls,total_us,avg_us,min_us,max_us,percent")
        for (n, g, wg, dec), v in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
            print(f"\"{n}\",{g},{wg},1e{dec},{len(v)},{sum(v):.1f},{sum(v)/len(v):.2f},{min(v):.2f},{max(v):.2f},{100*sum(v)/tot:.2f}")
    elif mode == "timeline":
        last = int(sys.argv[3]) if len(sys.argv) > 3 else 100
        rows = con.execute("select name, grid_x*grid_y*grid_z, workgroup_x*workgroup_y*workgroup_z, start, end from kernels order by start").fetchall()[-last:]
        print("start_us,duration_us,gap_before_us,grid_threads,workgroup,kernel")
        t0, prev = rows[0][3], None
        for n, g, wg, st, en in rows:
            print(f"{(st-t0)/1e3:.1f},{(en-st)/1e3:.1f},{((st-prev)/1e3 if prev is not None else 0):.1f},{g},{wg},\"{short(n)}\"")
            prev = en
    else:
        sub = sys.argv[3] if len(sys.argv) > 3 else ""
        rows = con.execute("select kernel_name, grid_size, counter_name, count(*), avg(value), avg(duration) from counters_collection "
                           "where kernel_name like ? group by kernel_name, grid_size, counter_name", (f"%{sub}%",)).fetchall()
        print("kernel,grid_threads,counter,launches,mean_value,avg_us")
        for n, g, cn, c, v, d in rows:
            print(f"\"{short(n)}\",{g},{cn},{c},{v:.6g},{(d or 0)/1e3:.2f}")


main()
