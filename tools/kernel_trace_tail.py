"""Summarises the LAST n dispatches of one kernel in a rocprofv3 --kernel-trace CSV (…_kernel_trace.csv).

    rocprofv3 --kernel-trace --output-format csv -d DIR -o matrix -- python3 bench.py --matrix-only --steps 100 ...
    python tools/kernel_trace_tail.py DIR k_hamming_matrix 100 > profiles/r3_matrix_timed_only.json

bench.py --matrix-only launches 400 untimed pre-roll + W warm-up launches (clock settling) and then
the K timed ones, all the same kernel: `rocprofv3 --stats` averages over all of them, the tail of the trace is exactly the
timed region.  Prints one JSON object: average / median / min / max duration of the tail, of the head, and the roofline
fraction of the tail for the 20000 x 20000 u16 matrix (801.28 MB of algorithmic bytes)."""
import csv
import glob
import json
import os
import sys


def main():
    d, kernel, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
    paths = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if not paths:
        raise SystemExit(f"no *kernel_trace.csv under {d}")
    rows = []
    for row in csv.DictReader(open(paths[0])):
        if kernel in row["Kernel_Name"]:
            rows.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"])))
    rows.sort()
    dur = [(b - a) / 1e3 for a, b in rows]            # us
    if len(dur) < n:
        raise SystemExit(f"only {len(dur)} dispatches of {kernel}")
    tail, head = sorted(dur[-n:]), sorted(dur[:-n]) or [0.0]
    alg = 32 * 40000 + 2 * 20000 * 20000
    avg = sum(tail) / n
    print(json.dumps(dict(kernel=kernel, trace=os.path.basename(paths[0]), dispatches=len(dur), tail_n=n,
                          tail_us=dict(avg=avg, median=tail[n // 2], min=tail[0], max=tail[-1]),
                          head_us=dict(n=len(dur) - n, avg=sum(head) / len(head), min=head[0], max=head[-1]),
                          span_ms=(rows[-1][1] - rows[-n][0]) / 1e6,
                          algorithmic_bytes=alg, tail_GBps=alg / avg / 1e3, tail_frac_of_8TBps=alg / avg / 1e3 / 8000.0)))


if __name__ == "__main__":
    main()
