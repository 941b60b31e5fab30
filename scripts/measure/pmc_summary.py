#!/usr/bin/env python
"""Summarise rocprofv3 --pmc CSVs: mean counter value per dispatch for kernels matching a substring."""
import csv, glob, sys, collections
d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else 'hpk_stencil'
acc = collections.defaultdict(list)
for f in sorted(glob.glob(d + '/*_counter_collection.csv')):
    for row in csv.DictReader(open(f)):
        if pat in row['Kernel_Name']:
            acc[row['Counter_Name']].append(float(row['Counter_Value']))
for k, v in acc.items():
    print('%-28s n=%d mean=%.4g' % (k, len(v), sum(v) / len(v)))
