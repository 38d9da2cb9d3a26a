#!/usr/bin/env python
"""
Aggregates rocprofv3 output directories into one JSON summary:
  * every *_counter_collection.csv found below the given directories -> per-kernel sums of
    each counter (kernel names shortened to the function name) and the dispatch count;
  * every *_kernel_stats.csv -> copied rows (name, calls, total/avg ns, percentage).
Usage: pmc_summary.py OUT.json DIR [DIR ...]
"""
import csv
import json
import os
import re
import sys


def short(name):
    m = re.search(r'(k[234p]?_[a-z_]+|hy_[a-z_0-9]+)', name)
    return m.group(1) if m else name.split('(')[0]


def main():
    out_path, dirs = sys.argv[1], sys.argv[2:]
    counters, stats = {}, []
    for d in dirs:
        for root, _, files in os.walk(d):
            for f in files:
                path = os.path.join(root, f)
                if f.endswith('counter_collection.csv'):
                    seen = set()
                    with open(path) as fh:
                        for row in csv.DictReader(fh):
                            k = short(row['Kernel_Name'])
                            e = counters.setdefault(k, {})
                            e[row['Counter_Name']] = e.get(row['Counter_Name'], 0.) + \
                                float(row['Counter_Value'])
                            key = (row['Dispatch_Id'], k)
                            if key not in seen:
                                seen.add(key)
                                e['_dispatches_' + os.path.basename(d.rstrip('/'))] = \
                                    e.get('_dispatches_' + os.path.basename(d.rstrip('/')), 0) + 1
                                for col in ('VGPR_Count', 'Accum_VGPR_Count', 'SGPR_Count',
                                            'LDS_Block_Size', 'Workgroup_Size', 'Scratch_Size'):
                                    if col in row:
                                        e.setdefault('_' + col, row[col])
                elif f.endswith('kernel_stats.csv'):
                    with open(path) as fh:
                        for row in csv.DictReader(fh):
                            row = dict(row)
                            row['Name'] = short(row.get('Name', ''))
                            stats.append(row)
    # hash of the kernel sources the counters were measured on: bench.py quotes the HBM traffic
    # of this summary only for the same code (kernel_source_hash there)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    doc = {'counters': counters, 'kernel_stats': stats,
           'kernel_source_sha': bench.kernel_source_hash()}
    with open(out_path, 'w') as fh:
        json.dump(doc, fh, indent=1, sort_keys=True)
    print(json.dumps(doc, indent=1, sort_keys=True))


if __name__ == '__main__':
    main()
