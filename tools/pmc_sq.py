"""Per-kernel averages of SQ counters from one rocprofv3 --pmc pass.  usage: pmc_sq.py <counter_collection.csv> [name filter]"""
import csv, sys
from collections import defaultdict
tot, n = defaultdict(lambda: defaultdict(float)), defaultdict(lambda: defaultdict(int))
flt = sys.argv[2] if len(sys.argv) > 2 else ''
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        k = r['Kernel_Name']
        if flt and flt not in k:
            continue
        tot[k][r['Counter_Name']] += float(r['Counter_Value'])
        n[k][r['Counter_Name']] += 1
for k in tot:
    print(k[:90])
    for c in sorted(tot[k]):
        print(f'    {c:28s} {tot[k][c] / n[k][c]:16.0f}  (x{n[k][c]})')
