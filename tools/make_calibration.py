"""Calibration of the rocprofv3 FETCH_SIZE / WRITE_SIZE counters (KB per dispatch) on kernels whose HBM bytes are known:
tools/gather_ubench.hip under the same two --pmc passes as the benchmark.
    python tools/make_calibration.py <fetch counter csv> <write counter csv> > profiles/r03_counter_calibration.json
factor = counter bytes / known bytes per launch (1.0 = the counter reports what the kernel moves)."""
import collections, csv, json, sys

N = 212992            # rows per launch of the 213 K-row cases (the first size the microbenchmark runs)
KNOWN = {             # kernel name prefix -> (bytes read, bytes written) per launch at n = N
    'k_gather<2>': (N * 64 + N * 4, N * 64), 'k_gather<4>': (N * 64 + N * 4, N * 64),
    'k_rmw': (N * 64 + N * 4, N * 64),
    'k_adamlike(': (N * (64 + 128 + 64) + N * 4, N * (64 + 128)),
    'k_touch<true>': (N * (64 + 128) + N * 4, 0), 'k_touch<false>': (N * 128 + N * 4, 0),
    'k_store_rows': (N * 64 + N * 4, N * (64 + 128)),
    'k_copy': (None, None),        # sizes vary per launch: 28 MB and 256 MB (read = written)
}


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == counter:
            acc[r['Kernel_Name']].append(float(r['Counter_Value']) * 1024.0)
    return acc


def main():
    fetch, write = per_kernel(sys.argv[1], 'FETCH_SIZE'), per_kernel(sys.argv[2], 'WRITE_SIZE')
    out = {'source': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE -- tools/ub_gather (two passes); '
                     'factor = counter bytes / known bytes; launches at n = 212,992 rows only (the smallest value per kernel)',
           'kernels': {}}
    for name, vals in fetch.items():
        short = name.replace('void ', '')
        key = next((k for k in KNOWN if short.startswith(k)), None)
        if key is None:
            continue
        w = write.get(name, [0.0])
        if key == 'k_copy':
            f_small = min(vals)
            out['kernels']['k_copy 28 MB (16 B/lane stream)'] = {
                'fetch_factor': round(f_small / (28 << 20), 3), 'write_factor': round(min(w) / (28 << 20), 3)}
            continue
        rd, wr = KNOWN[key]
        # the 213 K-row launches are the smaller group of values (the 1 M-row launches follow)
        small = sorted(vals)[:max(1, len(vals) // 2)]
        smallw = sorted(w)[:max(1, len(w) // 2)]
        e = {'known_read_bytes': rd, 'known_write_bytes': wr, 'fetch_bytes': round(sum(small) / len(small)),
             'fetch_factor': round(sum(small) / len(small) / rd, 3)}
        if wr:
            e['write_bytes'] = round(sum(smallw) / len(smallw))
            e['write_factor'] = round(sum(smallw) / len(smallw) / wr, 3)
        out['kernels'][short.split('(')[0]] = e
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
