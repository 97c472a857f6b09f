"""Per-kernel HBM-side traffic from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; KB units; FETCH doubled on gfx950 as
MI355X_MICROARCH.md prescribes).  usage: pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <steps>"""
import csv, sys
from collections import defaultdict


def load(path, counter):
    tot, n = defaultdict(float), defaultdict(int)
    with open(path) as fh:
        for r in csv.DictReader(fh):
            if r.get('Counter_Name') != counter:
                continue
            k = r['Kernel_Name']
            tot[k] += float(r['Counter_Value'])
            n[k] += 1
    return tot, n


def main():
    f, nf = load(sys.argv[1], 'FETCH_SIZE')
    w, nw = load(sys.argv[2], 'WRITE_SIZE')
    steps = float(sys.argv[3])
    rows = []
    for k in f:
        fetch = 2.0 * f[k] / nf[k] * 1024 / 1e6
        write = (w.get(k, 0.0) / max(nw.get(k, 1), 1)) * 1024 / 1e6
        rows.append((f[k] * 2 + w.get(k, 0.0), k, nf[k] / steps, fetch, write))
    rows.sort(reverse=True)
    print('| kernel | launches/step | fetch MB/launch | write MB/launch |\n|---|---|---|---|')
    for _, k, l, fe, wr in rows[:int(sys.argv[4]) if len(sys.argv) > 4 else 28]:
        print(f'| `{k[:72]}` | {l:.0f} | {fe:.0f} | {wr:.0f} |')


if __name__ == '__main__':
    main()
