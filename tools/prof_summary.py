"""Print a per-step summary of a rocprofv3 kernel_stats.csv: python tools/prof_summary.py <csv> <steps>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = sum(float(r['TotalDurationNs']) for r in rows) / 1e6 / steps
print(f'total kernel time per step: {tot:.2f} ms')
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print(f"{r['Name'][:64]:64s} calls/step={float(r['Calls'])/steps:6.1f} avg_us={float(r['AverageNs'])/1e3:8.1f} "
          f"ms/step={float(r['TotalDurationNs'])/1e6/steps:7.2f} {float(r['Percentage']):5.1f}%")
