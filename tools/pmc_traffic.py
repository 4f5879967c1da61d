"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (counter_collection.csv, one directory per pass) into the per-kernel table and
the per-hop traffic figure bench.py reports as roofline.traffic for the register-blocked schedule.
    python3 tools/pmc_traffic.py <fetch_dir> <write_dir> <calib_fetch_dir> <known_calib_bytes> <out_prefix>"""
import csv, glob, json, os, sys
from collections import defaultdict


def per_kernel(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == counter:
                acc[r['Kernel_Name']].append(float(r['Counter_Value']))
    return acc


fetch_dir, write_dir, calib_dir, known, out = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4]), sys.argv[5]
F, W, Cal = per_kernel(fetch_dir, 'FETCH_SIZE'), per_kernel(write_dir, 'WRITE_SIZE'), per_kernel(calib_dir, 'FETCH_SIZE')
cal_name = [k for k in Cal if 'spmm_blocked64_kernel' in k][0]
long_ = [(k, len(F[k]), sum(F[k]) / len(F[k]), sum(W[k]) / len(W[k]) if k in W else 0.0) for k in F if 'spmm_long_rows_kernel' in k]
cal_raw = sum(Cal[cal_name]) / len(Cal[cal_name]) * 1024.0
factor = known / cal_raw
rows = []
for k in sorted(F, key=lambda k: -sum(F[k])):
    if 'spmm' in k:
        rows.append((k, len(F[k]), sum(F[k]) / len(F[k]), sum(W[k]) / len(W[k]) if k in W else float('nan')))
hub = [r for r in rows if 'spmm_rows_kernel' in r[0]]
with open(out + '_per_kernel.csv', 'w') as fh:
    fh.write('kernel,launches,FETCH_SIZE_KB_avg_raw,WRITE_SIZE_KB_avg_raw\n')
    for k, n, f, w in rows:
        fh.write('"%s",%d,%.1f,%.1f\n' % (k, n, f, w))
# one full-graph hop = every spmm_blocked64_kernel launch of the hop (user rows + item rows) + the hub rows' chunk kernel;
# per-hop traffic = total over those kernels / number of hops (the AXPBY / LAYERSUM / ADAM variants are averaged together)
blk = [(k, n, f, w) for k, n, f, w in rows if 'spmm_blocked64_kernel' in k]
n_hops = sum(n for _, n, _, _ in blk) / 2.0
fetch_blk = sum(n * f for _, n, f, _ in blk) * 1024.0 / n_hops
write_blk = sum(n * w for _, n, _, w in blk) * 1024.0 / n_hops
res = {'workload': 'cfg2', 'kernel': 'spmm_blocked64_kernel<32,*> (two launches per hop: user rows, item rows; rows above the per-set threshold as strided pieces + spmm_long_rows_kernel combine)', 'hops': n_hops,
       'split_row_combine_kernel': {'fetch_raw_bytes_per_hop': sum(n * f for _, n, f, _ in long_) * 1024.0 / n_hops, 'write_raw_bytes_per_hop': sum(n * w for _, n, _, w in long_) * 1024.0 / n_hops},
       'fetch_raw_bytes_per_hop': fetch_blk, 'write_raw_bytes_per_hop': write_blk,
       'calibration': {'kernel': cal_name, 'known_bytes': known, 'raw_bytes': cal_raw, 'factor': factor,
                       'note': 'tools/pmc_calibrate.py: every operand row gathered once from a 1.07 GB table (4 B per lane, 256 B per wave load)'},
       'hub_rows_kernel': {'note': 'spmm_rows_kernel on the hub rows of the same hops (16-B-per-lane gathers: FETCH_SIZE doubled per the guide)',
                           'fetch_raw_bytes_per_hop': sum(n * f for _, n, f, _ in hub) * 1024.0 / n_hops, 'write_raw_bytes_per_hop': sum(n * w for _, n, _, w in hub) * 1024.0 / n_hops},
       'traffic_corrected_bytes': fetch_blk * factor + write_blk + 2.0 * sum(n * f for _, n, f, _ in hub) * 1024.0 / n_hops + sum(n * w for _, n, _, w in hub) * 1024.0 / n_hops
                                  + 2.0 * sum(n * f for _, n, f, _ in long_) * 1024.0 / n_hops + sum(n * w for _, n, _, w in long_) * 1024.0 / n_hops,
       'correction': 'FETCH_SIZE of the blocked kernel scaled by the factor measured on a known byte count in the same access pattern '
                     '(MI355X_MICROARCH.md HBM section: widths other than 16 B/lane must be calibrated); WRITE_SIZE taken as is; Infinity-Cache hits are '
                     'included in FETCH_SIZE; the hub rows of the hop (chunked CSR kernel, 7.8 % of the edges) are added with the x2 of the guide on their dwordx4 gathers'}
json.dump(res, open(out + '.json', 'w'), indent=1)
print(json.dumps(res, indent=1))
