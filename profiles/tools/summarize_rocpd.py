#!/usr/bin/env python
"""Turn rocprofv3 (ROCm 7.2, rocpd sqlite output) result databases into the text summaries kept
under profiles/.  Usage: summarize_rocpd.py <dir with */*_results.db> > profiles/rNN_xxx.txt"""
import glob
import os
import sqlite3
import sys


def q(db, sql):
    con = sqlite3.connect(db)
    try:
        return con.execute(sql).fetchall()
    finally:
        con.close()


def main(root):
    for db in sorted(glob.glob(os.path.join(root, "*", "*_results.db"))):
        tag = os.path.basename(os.path.dirname(db))
        print(f"== {tag} ({os.path.basename(db)})")
        try:
            rows = q(db, "select name, total_calls, total_duration, average, percentage from top_kernels")
            if rows:
                print("  kernel-trace stats (durations in us):")
                for r in rows[:6]:
                    print(f"    {r[0][:70]:70s} calls={r[1]:6d} total={r[2]:12.1f} avg={r[3]:9.3f} pct={r[4]:6.2f}")
            rows = q(db, "select kernel_name, vgpr_count, accum_vgpr_count, sgpr_count, lds_block_size, workgroup_size, grid_size, count(*) "
                         "from counters_collection group by kernel_name")
            for r in rows:
                print(f"    {r[0][:60]:60s} vgpr={r[1]} agpr={r[2]} sgpr={r[3]} lds={r[4]} wg={r[5]} grid={r[6]} dispatches={r[7]}")
            rows = q(db, "select kernel_name, counter_name, count(*), avg(value), min(value), max(value) from counters_collection "
                         "group by kernel_name, counter_name order by kernel_name, counter_name")
            if rows:
                print("  PMC per dispatch (FETCH_SIZE/WRITE_SIZE in KiB):")
                for r in rows:
                    if "rocclr" in r[0]:
                        continue
                    print(f"    {r[0][:46]:46s} {r[1]:22s} n={r[2]:5d} avg={r[3]:14.1f} min={r[4]:14.1f} max={r[5]:14.1f}")
        except sqlite3.Error as e:
            print("  (", e, ")")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else ".")
