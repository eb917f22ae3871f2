#!/usr/bin/env python3
"""Reduce the rocprofv3 output of profiles/collect.sh to the small files kept under profiles/<round>/:
kernel_stats.csv (the profiler's own per-kernel statistics, library-internal kernels dropped) and
pmc_hbm_bytes.json (FETCH_SIZE / WRITE_SIZE in KB, averaged per launch, by kernel)."""
import csv
import glob
import json
import os
import re
import shutil
import sys


def short_name(kernel):
    """'void (anonymous namespace)::k_bin_hash_direct<512, 8>(...)' -> 'k_bin_hash_direct'"""
    m = re.search(r'(k_[a-z0-9_]+)', kernel)
    return m.group(1) if m else kernel.split('(')[0].strip()


def main(src, dst):
    os.makedirs(dst, exist_ok=True)
    stats = glob.glob(os.path.join(src, 'trace', '**', '*kernel_stats.csv'), recursive=True)
    if stats:
        with open(stats[0]) as fh, open(os.path.join(dst, 'kernel_stats.csv'), 'w') as out:
            rows = list(csv.reader(fh))
            w = csv.writer(out)
            w.writerow(rows[0])
            for row in rows[1:]:
                if row and ('k_' in row[0] or 'memset' in row[0].lower() or 'rocprim' in row[0]):
                    w.writerow(row)
    pmc = {}
    for counter, sub in (('FETCH_SIZE', 'pmc_fetch'), ('WRITE_SIZE', 'pmc_write')):
        files = glob.glob(os.path.join(src, sub, '**', '*counter_collection.csv'), recursive=True)
        if not files:
            continue
        acc = {}
        with open(files[0]) as fh:
            for row in csv.DictReader(fh):
                if row.get('Counter_Name') != counter:
                    continue
                name = short_name(row['Kernel_Name'])
                key = (name, row.get('Dispatch_Id'))
                acc[key] = acc.get(key, 0.0) + float(row['Counter_Value'])
        per = {}
        for (name, _), v in acc.items():
            per.setdefault(name, []).append(v)
        for name, vals in per.items():
            d = pmc.setdefault(name, {})
            d[counter + '_KB_per_launch_avg'] = round(sum(vals) / len(vals), 1)
            d['launches_' + counter] = len(vals)
    with open(os.path.join(dst, 'pmc_hbm_bytes.json'), 'w') as fh:
        json.dump(pmc, fh, indent=1)
    for name in ('bench.json', 'bench_under_rocprof.json'):
        p = os.path.join(src, name)
        if os.path.exists(p):
            shutil.copy(p, os.path.join(dst, name))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
