#!/usr/bin/env python
"""Per-kernel summary (calls, total, average, share, registers, LDS) of a rocprofv3 `--kernel-trace` run stored in the
rocpd SQLite format (the default output of rocprofv3 on ROCm 7.2):   python tools/rocpd_stats.py <results.db> [top-N]"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("void ", "").replace("xc::", "").replace("unsigned short", "bf16")
    return name[:110]


def main():
    db = sqlite3.connect(sys.argv[1])
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), "
                      "max(accum_vgpr_count), max(sgpr_count), max(lds_size) from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows)
    print(f"{'kernel':110s} {'calls':>6s} {'total ms':>10s} {'avg us':>10s} {'min us':>9s} {'max us':>9s} {'%':>6s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'lds':>7s}")
    for r in rows[:top]:
        print(f"{short(r[0]):110s} {r[1]:6d} {r[2] / 1e6:10.3f} {r[3] / 1e3:10.2f} {r[4] / 1e3:9.2f} {r[5] / 1e3:9.2f} {100 * r[2] / total:6.2f} {r[6]:5d} {r[7]:5d} {r[8]:5d} {r[9]:7d}")
    print(f"{'TOTAL kernel time':110s} {sum(r[1] for r in rows):6d} {total / 1e6:10.3f}")


if __name__ == "__main__":
    main()
