"""Turn a rocprofv3 rocpd database (<out>_results.db) into small CSV summaries.
usage: python tools/rocpd_summary.py stats <db> > kernel_stats.csv      # per (kernel, grid) durations
       python tools/rocpd_summary.py pmc <db> [kernel-substring]         # per (kernel, grid, counter) means
Launches of one kernel at different grid sizes are kept apart so that the bench's full-size launches
are not averaged with the small side configurations that run in the same process."""
import sqlite3
import sys


def short(name):
    return name.split("(")[0].replace("void ", "")


def main():
    mode, db = sys.argv[1], sys.argv[2]
    con = sqlite3.connect(db)
    if mode == "stats":
        cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
        gcol = "grid_size" if "grid_size" in cols else "grid_size_x"
        rows = con.execute(f"select name, {gcol}, workgroup_size, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                           f"from kernels group by name, {gcol}, workgroup_size order by sum(end-start) desc").fetchall()
        tot = sum(r[4] for r in rows) or 1
        print("kernel,grid_threads,workgroup,calls,total_us,avg_us,min_us,max_us,percent")
        for n, g, wg, c, s, a, mn, mx in rows:
            print(f"\"{short(n)}\",{g},{wg},{c},{s/1e3:.1f},{a/1e3:.2f},{mn/1e3:.2f},{mx/1e3:.2f},{100*s/tot:.2f}")
    else:
        sub = sys.argv[3] if len(sys.argv) > 3 else ""
        rows = con.execute("select kernel_name, grid_size, counter_name, count(*), avg(value), avg(duration) from counters_collection "
                           "where kernel_name like ? group by kernel_name, grid_size, counter_name", (f"%{sub}%",)).fetchall()
        print("kernel,grid_threads,counter,launches,mean_value,avg_us")
        for n, g, cn, c, v, d in rows:
            print(f"\"{short(n)}\",{g},{cn},{c},{v:.6g},{(d or 0)/1e3:.2f}")


main()
