"""Kernel statistics of a rocprofv3 run (ROCm 7.2 writes an SQLite `*_results.db`): per kernel calls, average / min / max
duration and share of the total kernel time, as the tracked CSV summaries under profiles/.

    python scripts/rocprof_db_stats.py gpurun_out/prof_x/x_results.db profiles/r02_x_kernel_stats.csv "title"
"""
import sqlite3
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    tail = name[name.rfind(" [grid "):] if " [grid " in name else ""
    if tail:
        name = name[:-len(tail)]
    cut = name.find("(")
    return (name if cut < 0 else name[:cut])[:70] + tail


def main():
    db_path, out_path = sys.argv[1], sys.argv[2]
    title = sys.argv[3] if len(sys.argv) > 3 else db_path
    cur = sqlite3.connect(db_path).cursor()
    # --by-grid: one line per (kernel, grid size) - tells a launch group's union launches (e.g. 1024 workgroups) from the
    # single-batch launches of the same kernel in the same run
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
    by_grid = "--by-grid" in sys.argv and "grid_x" in cols
    key = "name || ' [grid ' || grid_x || ']'" if by_grid else "name"
    rows = cur.execute("select %s, count(*), avg(end - start), min(end - start), max(end - start), sum(end - start) "
                       "from kernels group by 1 order by 6 desc" % key).fetchall()
    total = sum(r[5] for r in rows) or 1
    with open(out_path, "w") as f:
        f.write("# %s\n# source: rocprofv3 --kernel-trace --stats (%s), MI355X\n" % (title, db_path))
        f.write("kernel,calls,avg_us,min_us,max_us,percent\n")
        for name, calls, avg, mn, mx, tot in rows:
            f.write("%s,%d,%.2f,%.2f,%.2f,%.2f\n" % (short(name), calls, avg / 1e3, mn / 1e3, mx / 1e3, 100.0 * tot / total))
    print(open(out_path).read())


if __name__ == "__main__":
    main()
