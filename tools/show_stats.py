#!/usr/bin/env python3
"""tools/show_stats.py <kernel_stats.csv> <steps> : per-step kernel time table"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("total per step %.2f ms" % (tot / 1e6 / steps))
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:int(sys.argv[3]) if len(sys.argv) > 3 else 28]:
    print("%-100s calls %5s avg %9.1f us  /step %7.2f ms  %4.1f%%" % (r['Name'][:100], r['Calls'], float(r['AverageNs']) / 1e3,
          float(r['TotalDurationNs']) / 1e6 / steps, 100 * float(r['TotalDurationNs']) / tot))
