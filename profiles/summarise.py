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
    if not m:
        return kernel.split('(')[0].strip()
    name = m.group(1)
    if name in ('k_bin_split', 'k_bin_apply') and re.search(r'<[^>]*true', kernel):       # the weighted template instances
        name += '_w'
    return name


def kernel_sources_sha(root):
    """sha256 over the kernel sources: a PMC file says which kernels it measured"""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(root, 'kevlar_amd', 'csrc')
    for name in sorted(os.listdir(csrc)):
        if name.endswith(('.hip', '.h')):
            h.update(name.encode())
            h.update(open(os.path.join(csrc, name), 'rb').read())
    return h.hexdigest()[:16]


def reduce_pmc(src):
    """{'raw_KB_per_launch': ..., 'kernels': {name: {fetch_factor, launches_per_step, hbm_bytes_per_launch, hbm_bytes_per_step}}} from
    the counter CSVs under src/pmc_fetch and src/pmc_write (one bench step each)"""
    pmc = {}
    for counter, sub in (('FETCH_SIZE', 'pmc_fetch'), ('WRITE_SIZE', 'pmc_write')):
        files = sorted(glob.glob(os.path.join(src, sub, '**', '*counter_collection.csv'), recursive=True), key=os.path.getmtime, reverse=True)
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
    # HBM bytes per bench step by kernel: FETCH_SIZE x correction + WRITE_SIZE, x launches per step.  The guide
    # (MI355X_MICROARCH.md, HBM) measured FETCH_SIZE at exactly 1/2 of the bytes of wide coalesced reads on gfx950;
    # that holds for the streaming kernels here (checked against their known input sizes, profiles/README.md), while
    # random single-sector reads (the scan's table probes) calibrate at 1.0.
    streaming = ('k_skm_split', 'k_skm_count', 'k_bin_split', 'k_bin_apply', 'k_bin_hash', 'k_bin_list', 'k_skm_emit')
    kernels = {}
    for name, d in pmc.items():
        if not name.startswith('k_'):
            continue
        factor = 2.0 if name.startswith(streaming) else 1.0
        fetch_kb, write_kb = d.get('FETCH_SIZE_KB_per_launch_avg', 0.0), d.get('WRITE_SIZE_KB_per_launch_avg', 0.0)
        launches = d.get('launches_FETCH_SIZE', d.get('launches_WRITE_SIZE', 0))      # the PMC passes run ONE step
        kernels[name] = {'fetch_factor': factor, 'launches_per_step': launches,
                         'hbm_bytes_per_launch': int((fetch_kb * factor + write_kb) * 1024),
                         'hbm_bytes_per_step': int((fetch_kb * factor + write_kb) * 1024 * launches)}
    return {'raw_KB_per_launch': pmc, 'kernels': kernels}


def main(src, dst):
    os.makedirs(dst, exist_ok=True)
    stats = sorted(glob.glob(os.path.join(src, 'trace', '**', '*kernel_stats.csv'), recursive=True), key=os.path.getmtime, reverse=True)
    if stats:
        with open(stats[0]) as fh, open(os.path.join(dst, 'kernel_stats.csv'), 'w') as out:
            rows = list(csv.reader(fh))
            w = csv.writer(out)
            w.writerow(rows[0])
            for row in rows[1:]:
                if row and ('k_' in row[0] or 'memset' in row[0].lower() or 'rocprim' in row[0]):
                    w.writerow(row)
    out = {'collected': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes of bench.py --steps 1 --warmup 0 --count-streams 1',
           'kernel_sources_sha16': kernel_sources_sha(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))}
    out.update(reduce_pmc(src))
    with open(os.path.join(dst, 'pmc_hbm_bytes.json'), 'w') as fh:
        json.dump(out, fh, indent=1)
    for name in ('bench.json', 'bench_under_rocprof.json', 'bench_one_stream.json', 'bench_under_rocprof_default.json'):
        p = os.path.join(src, name)
        if os.path.exists(p):
            shutil.copy(p, os.path.join(dst, name))
    # the kernel statistics of the default command (samples counted concurrently on three streams)
    stats = sorted(glob.glob(os.path.join(src, 'trace_default', '**', '*kernel_stats.csv'), recursive=True), key=os.path.getmtime, reverse=True)
    if stats:
        with open(stats[0]) as fh, open(os.path.join(dst, 'kernel_stats_default_3_streams.csv'), 'w') as out:
            rows = list(csv.reader(fh))
            w = csv.writer(out)
            w.writerow(rows[0])
            for row in rows[1:]:
                if row and ('k_' in row[0] or 'memset' in row[0].lower() or 'rocprim' in row[0]):
                    w.writerow(row)
    # the ingest kernels: one gzip stream and one BGZF file of 2 M reads inflated on the device (scratch/gunzip_rate.py,
    # scratch/inflate_rate.py under the profiler)
    for sub, name in (('trace_gunzip', 'ingest_gzip_kernel_stats.csv'), ('trace_bgzf', 'ingest_bgzf_kernel_stats.csv')):
        stats = sorted(glob.glob(os.path.join(src, sub, '**', '*kernel_stats.csv'), recursive=True), key=os.path.getmtime, reverse=True)
        if stats:
            with open(stats[0]) as fh, open(os.path.join(dst, name), 'w') as out:
                rows = list(csv.reader(fh))
                w = csv.writer(out)
                w.writerow(rows[0])
                for row in rows[1:]:
                    if row and 'k_' in row[0]:
                        w.writerow(row)
    stats = sorted(glob.glob(os.path.join(src, 'trace_cfg5', '**', '*kernel_stats.csv'), recursive=True), key=os.path.getmtime, reverse=True)
    if stats:
        with open(stats[0]) as fh, open(os.path.join(dst, 'cfg5_kernel_stats.csv'), 'w') as out:
            rows = list(csv.reader(fh))
            w = csv.writer(out)
            w.writerow(rows[0])
            for row in rows[1:]:
                if row and ('k_' in row[0] or 'memset' in row[0].lower() or 'rocprim' in row[0]):
                    w.writerow(row)
    stats = sorted(glob.glob(os.path.join(src, 'trace_cfg4', '**', '*kernel_stats.csv'), recursive=True), key=os.path.getmtime, reverse=True)
    if stats:
        with open(stats[0]) as fh, open(os.path.join(dst, 'cfg4_band_kernel_stats.csv'), 'w') as out:
            rows = list(csv.reader(fh))
            w = csv.writer(out)
            w.writerow(rows[0])
            for row in rows[1:]:
                if row and ('k_' in row[0] or 'memset' in row[0].lower() or 'rocprim' in row[0]):
                    w.writerow(row)
    for name in ('bench_cfg4.json', 'bench_cfg4_under_rocprof.json'):
        if os.path.exists(os.path.join(src, name)):
            shutil.copy(os.path.join(src, name), os.path.join(dst, name))
    for name in ('bench_cfg5.json', 'bench_cfg5_under_rocprof.json'):
        if os.path.exists(os.path.join(src, name)):
            shutil.copy(os.path.join(src, name), os.path.join(dst, name))
    for log in ('gunzip_rate.log', 'inflate_rate.log'):
        if os.path.exists(os.path.join(src, log)):
            shutil.copy(os.path.join(src, log), os.path.join(dst, log))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
