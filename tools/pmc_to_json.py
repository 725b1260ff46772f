#!/usr/bin/env python
"""profiles/stft_pmc.json from rocprofv3 --pmc passes of tools/pmc_stft.sh (the HBM bytes per launch that bench.py reports as
`roofline.traffic` / `roofline_config5.traffic`; counters cannot be read inside an un-profiled run).

  usage: tools/pmc_to_json.py n1024=<gpurun_out dir of `pmc_stft.sh <dir> 1024 1024 44100`> n4096=<dir of `pmc_stft.sh <dir> 4096 32 1323000`>

HBM bytes = FETCH_SIZE x 2 + WRITE_SIZE (KB -> bytes): on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads
(MI355X_MICROARCH.md, HBM section); WRITE_SIZE as reported."""
import csv, glob, json, os, sys, collections

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORK = {'n1024': ('stft_fwd_n1024_kernel<mag>', '1024 clips x 2 s, 1024/256', 4 * 1024 * 44100 + 4 * 1024 * 513 * 173),
        'n4096': ('stft_fwd 4096/1024 <mag>', '32 clips x 30 s at 44.1 kHz, 4096/1024', 4 * 32 * 1323000 + 4 * 32 * 2049 * 1292),
        # psnd_stft_mag_nfk: the same transforms writing (N, F, K) (tools/pmc_stft.sh ... RUNNER=tools/r04/run_nfk_only.py)
        'nfk1024': ('stft_fwd_n1024q_kernel', '1024 clips x 2 s, 1024/256, output (N, F, K)', 4 * 1024 * 44100 + 4 * 1024 * 513 * 173),
        'nfk4096': ('stft_fwd_n4096r_kernel', '32 clips x 30 s at 44.1 kHz, 4096/1024, output (N, F, K)', 4 * 32 * 1323000 + 4 * 32 * 2049 * 1292),
        # psnd_mel_fwd (tools/pmc_stft.sh ... RUNNER=tools/r05/run_mel_only.py; kernel name match 'mel_kernel')
        'mel': ('mel_kernel', '1024 clips x 2 s: (N, K, F) magnitudes -> 80 log-mel bands', 4 * 1024 * 513 * 173 + 4 * 1024 * 80 * 173)}
MATCH = {'mel': 'mel_'}                 # mel_fwd_once_kernel (round 6: the read-once forward) or mel_kernel


def summarise(d, match='stft_fwd'):
    agg = collections.OrderedDict()
    for f in sorted(glob.glob(os.path.join(d, 'pmc_*', '*counter_collection.csv'))):
        for row in csv.DictReader(open(f)):
            if match in row.get('Kernel_Name', ''):
                agg.setdefault(row['Counter_Name'], []).append(float(row['Counter_Value']))
    dur, names = [], set()
    for f in sorted(glob.glob(os.path.join(d, 'pmc_*', '*kernel_trace.csv')))[:1]:
        for r in csv.DictReader(open(f)):
            if match in r['Kernel_Name']:
                dur.append((float(r['End_Timestamp']) - float(r['Start_Timestamp'])) / 1e3)
                names.add((__import__('re').search(r'(\w+_kernel)', r['Kernel_Name']) or [r['Kernel_Name'][:80]])[0])
    return {k: sum(v) / len(v) for k, v in agg.items()}, dur, sorted(names)


def main():
    out = {}
    for a in sys.argv[1:]:
        which, d = a.split('=', 1)
        c, dur, names = summarise(d, MATCH.get(which, 'stft_fwd'))
        kern, workload, alg = WORK[which]
        rd, wr = c['FETCH_SIZE'] * 1024 * 2, c['WRITE_SIZE'] * 1024
        out[which] = {'kernel': names[0] if names else kern, 'workload': workload, 'fetch_size_kb': c['FETCH_SIZE'], 'write_size_kb': c['WRITE_SIZE'],
                      'read_bytes': rd, 'write_bytes': wr, 'hbm_bytes_per_launch': rd + wr, 'algorithmic_bytes': alg,
                      'traffic_over_algorithmic': (rd + wr) / alg,
                      'write_requests': c.get('TCC_EA0_WRREQ_sum'), 'write_requests_64B': c.get('TCC_EA0_WRREQ_64B_sum'),
                      'kernel_us_under_pmc': [round(x, 1) for x in dur],
                      'source': 'RECORDED, not measured in the bench run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/pmc_stft.sh %s); '
                                'FETCH_SIZE doubled per MI355X_MICROARCH.md' % os.path.basename(d.rstrip('/'))}
    p = os.path.join(ROOT, 'profiles', 'stft_pmc.json')
    old = json.load(open(p)) if os.path.exists(p) else {}
    old = {k: v for k, v in old.items() if k in WORK}
    old.update(out)
    json.dump(old, open(p, 'w'), indent=1)
    print(json.dumps(old, indent=1))


if __name__ == '__main__':
    main()
