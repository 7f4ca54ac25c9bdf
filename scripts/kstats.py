"""Summarise a rocprofv3 *_kernel_stats.csv: ms/step per kernel.  usage: kstats.py <csv> <steps>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / steps / 1e6:.3f} ms/step")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print(f"{float(r['TotalDurationNs']) / steps / 1e6:8.3f} ms/step {int(r['Calls']) / steps:6.1f} calls/step avg {float(r['AverageNs']) / 1e3:8.1f} us  {r['Name'][:100]}")
