"""Timeline analysis of a rocprofv3 --kernel-trace CSV (kernel_trace.csv): how much of the wall time between the first and the last
kernel of the steady-state window has at least one kernel running, how much is idle, which kernels the idle gaps follow, and how the
time splits over queues (streams).   usage: python tools/trace_gaps.py <kernel_trace.csv> [length of the window at the END of the trace in ms, default 200]"""
import csv
import sys
from collections import defaultdict


def short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    for ch in "<(":
        i = name.find(ch)
        if i > 0:
            name = name[:i]
    return name[:48]


def main():
    path = sys.argv[1]
    win_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 200.0
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")))
    rows.sort()
    t0, t1 = rows[0][0], rows[-1][1]
    cut = max(r[1] for r in rows) - win_ms * 1e6
    rows = [r for r in rows if r[0] >= cut]
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    wall = t1 - t0
    busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
    gaps = []
    last_name = rows[0][2]
    for s, e, name, q in rows[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, last_name, name))
            cur_s, cur_e = s, e
            last_name = name
        else:
            if e > cur_e:
                cur_e, last_name = e, name
    busy += cur_e - cur_s
    ksum = sum(e - s for s, e, _, _ in rows)
    print("window %.3f ms, %d kernels: busy (union) %.3f ms = %.1f %%, idle %.3f ms in %d gaps; summed kernel time %.3f ms (concurrency %.2f)"
          % (wall / 1e6, len(rows), busy / 1e6, 100.0 * busy / wall, (wall - busy) / 1e6, len(gaps), ksum / 1e6, ksum / max(busy, 1)))
    perq = defaultdict(int)
    for s, e, _, q in rows:
        perq[q] += e - s
    print("per queue (ms):", ", ".join("%s %.2f" % (q, v / 1e6) for q, v in sorted(perq.items(), key=lambda kv: -kv[1])))
    hist = defaultdict(lambda: [0, 0])
    for g, a, b in gaps:
        k = "<2us" if g < 2000 else "<5us" if g < 5000 else "<10us" if g < 10000 else "<50us" if g < 50000 else ">=50us"
        hist[k][0] += 1
        hist[k][1] += g
    print("gap histogram:", ", ".join("%s: %d (%.3f ms)" % (k, v[0], v[1] / 1e6) for k, v in hist.items()))
    by = defaultdict(lambda: [0, 0])
    for g, a, b in gaps:
        key = short(a) + " -> " + short(b)
        by[key][0] += 1
        by[key][1] += g
    for k, v in sorted(by.items(), key=lambda kv: -kv[1][1])[:25]:
        print("  %8.3f ms  x%-5d %s" % (v[1] / 1e6, v[0], k))
    bk = defaultdict(lambda: [0, 0])
    for s, e, name, _ in rows:
        key = short(name)
        bk[key][0] += 1
        bk[key][1] += e - s
    print("kernels by time:")
    for k, v in sorted(bk.items(), key=lambda kv: -kv[1][1])[:25]:
        print("  %8.3f ms  x%-5d avg %7.1f us  %s" % (v[1] / 1e6, v[0], v[1] / v[0] / 1e3, k))


if __name__ == "__main__":
    main()
