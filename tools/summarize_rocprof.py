"""Summarise rocprofv3 rocpd databases (ROCm 7.2 default output) into text for profiles/.

    python tools/summarize_rocprof.py stats  <results.db>          # --kernel-trace --stats
    python tools/summarize_rocprof.py pmc    <results.db> [...]    # --pmc passes
"""
import sqlite3
import sys


def stats(db):
    cur = sqlite3.connect(db).cursor()
    print(f"# rocprofv3 --kernel-trace --stats  ({db})")
    print(f"{'kernel':<110} {'calls':>6} {'total_us':>12} {'avg_us':>12} {'%':>7}")
    for name, calls, total, avg, pct in cur.execute(
            "select name,total_calls,total_duration,average,percentage from top_kernels order by total_duration desc"):
        print(f"{name[:110]:<110} {calls:>6} {total:>12.1f} {avg:>12.3f} {pct:>7.2f}")
    print("\n# per-dispatch geometry of the product kernels")
    for r in cur.execute("select name, count(*), min(duration), avg(duration), max(duration), grid_x, workgroup_x, "
                         "lds_size, vgpr_count, sgpr_count from kernels where name like '%kta::%' or name like '%kafka_%' "
                         "group by name, grid_x, lds_size"):     # one row per kernel AND launch size
        print("  %s\n     n=%d dur_ns min/avg/max=%d/%d/%d grid_x=%d wg_x=%d lds=%d vgpr=%d sgpr=%d" % r)


def pmc(dbs):
    for db in dbs:
        cur = sqlite3.connect(db).cursor()
        print(f"# rocprofv3 --pmc  ({db})")
        print(f"{'kernel':<100} {'counter':<12} {'n':>4} {'avg':>16} {'min':>16} {'max':>16}")
        try:     # one row per kernel AND launch size: a kernel's launches of different sizes must not be averaged
            q = cur.execute("select kernel_name || ' grid=' || grid_size, counter_name, count(*), avg(value), min(value), "
                            "max(value) from counters_collection group by kernel_name, grid_size, counter_name").fetchall()
        except sqlite3.OperationalError:
            q = cur.execute("select kernel_name, counter_name, count(*), avg(value), min(value), max(value) "
                            "from counters_collection group by kernel_name, counter_name").fetchall()
        for r in q:
            name = r[0] if len(r[0]) <= 100 else r[0][:70] + " .. " + r[0][-26:]
            print(f"{name:<100} {r[1]:<22} {r[2]:>4} {r[3]:>16.1f} {r[4]:>16.1f} {r[5]:>16.1f}")


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    else:
        pmc(sys.argv[2:])
