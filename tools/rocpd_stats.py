"""Kernel statistics of a rocprofv3 run whose output is the rocpd SQLite database (ROCm 7.2 default):
   python tools/rocpd_stats.py gpurun_out/<dir>/<name>_results.db profiles/<name>_kernel_stats.csv
Writes name (template arguments kept, argument lists cut), calls, total / average duration in microseconds, share."""
import csv
import re
import sqlite3
import sys


def main():
    if len(sys.argv) != 3 or not sys.argv[2].endswith('.csv'):
        sys.exit(__doc__)      # a shell glob that matched two databases would make the second one the output
    db, out = sys.argv[1], sys.argv[2]
    con = sqlite3.connect(db)
    rows = con.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
        for name, calls, tot, avg, pct in rows:
            short = re.sub(r"\(.*$", "", name)
            short = re.sub(r"^void ", "", short)[:120]
            w.writerow([short, calls, "%.1f" % tot, "%.2f" % avg, "%.3f" % pct])
    print("wrote", out, len(rows), "kernels")


if __name__ == "__main__":
    main()
