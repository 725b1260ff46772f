#!/usr/bin/env python
"""per-kernel instruction counts from rocprofv3 --pmc passes: tools/pmc_per_kernel.py <dir with pmc_*/...counter_collection.csv>
prints waves, VALU / scalar / LDS / VMEM instructions per wave, the waiting share of the wave cycles and the mean duration"""
import csv, glob, os, sys, collections, re
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(root, 'pmc_*', '*counter_collection.csv'))):
    for row in csv.DictReader(open(f)):
        m = re.search(r'(\w+_kernel)', row['Kernel_Name'])
        name = (m.group(1) if m else row['Kernel_Name'][:40]) + ('<' + row['Kernel_Name'].split('<', 1)[1][:24] if '<' in row['Kernel_Name'] and m else '')
        agg[name][row['Counter_Name']].append(float(row['Counter_Value']))
dur = collections.defaultdict(list)
for f in sorted(glob.glob(os.path.join(root, 'pmc_*', '*kernel_trace.csv')))[:1]:
    for r in csv.DictReader(open(f)):
        m = re.search(r'(\w+_kernel)', r['Kernel_Name'])
        name = (m.group(1) if m else r['Kernel_Name'][:40]) + ('<' + r['Kernel_Name'].split('<', 1)[1][:24] if '<' in r['Kernel_Name'] and m else '')
        dur[name].append((float(r['End_Timestamp']) - float(r['Start_Timestamp'])) / 1e3)
print('%-60s %8s %8s %8s %8s %8s %7s %8s' % ('kernel', 'waves', 'VALU/w', 'SALU/w', 'LDS/w', 'VMEM/w', 'wait', 'us'))
mean = lambda v: sum(v) / len(v) if v else 0.0
for k, c in sorted(agg.items(), key=lambda kv: -mean(dur.get(kv[0], [0]))):
    w = mean(c.get('SQ_WAVES', [])) or 1.0
    print('%-60s %8.0f %8.0f %8.0f %8.0f %8.0f %7.2f %8.1f' % (k[:60], w, mean(c.get('SQ_INSTS_VALU', [])) / w, mean(c.get('SQ_INSTS_SALU', [])) / w,
          mean(c.get('SQ_INSTS_LDS', [])) / w, (mean(c.get('SQ_INSTS_VMEM_RD', [])) + mean(c.get('SQ_INSTS_VMEM_WR', []))) / w,
          mean(c.get('SQ_WAIT_ANY', [])) / (mean(c.get('SQ_WAVE_CYCLES', [])) or 1.0), mean(dur.get(k, [0]))))
