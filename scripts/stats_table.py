#!/usr/bin/env python3
"""Summarise rocprofv3 --kernel-trace --stats CSV output (*kernel_stats.csv) as a fixed-width table."""
import csv, glob, os, sys
rows = []
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, '**', '*kernel_stats.csv'), recursive=True):
        rows += list(csv.DictReader(open(f)))
print('{:<96s} {:>6s} {:>11s} {:>11s} {:>11s} {:>10s} {:>6s}'.format('kernel', 'calls', 'avg_us', 'min_us', 'max_us', 'total_ms', '%'))
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs'])):
    print('{:<96s} {:>6d} {:>11.2f} {:>11.2f} {:>11.2f} {:>10.3f} {:>6.1f}'.format(
        r['Name'][:96], int(r['Calls']), float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3,
        float(r['TotalDurationNs']) / 1e6, float(r['Percentage'])))
