#!/usr/bin/env python
"""Summarise rocprofv3 --pmc CSV output dirs: mean counter value per dispatch of kernels matching a substring."""
import csv, glob, os, sys, collections
root = sys.argv[1]; match = sys.argv[2] if len(sys.argv) > 2 else 'stft_fwd'
agg = collections.OrderedDict()
for f in sorted(glob.glob(os.path.join(root, 'pmc_*', '*counter_collection.csv'))):
    for row in csv.DictReader(open(f)):
        if match not in row.get('Kernel_Name', ''):
            continue
        agg.setdefault(row['Counter_Name'], []).append(float(row['Counter_Value']))
for k, v in agg.items():
    print('%-32s n=%d mean=%.4g' % (k, len(v), sum(v) / len(v)))
for f in sorted(glob.glob(os.path.join(root, 'pmc_*', '*kernel_trace.csv')))[:1]:
    d = [float(r['End_Timestamp']) - float(r['Start_Timestamp']) for r in csv.DictReader(open(f)) if match in r['Kernel_Name']]
    print('kernel duration under PMC (us):', [round(x / 1e3, 1) for x in d])
