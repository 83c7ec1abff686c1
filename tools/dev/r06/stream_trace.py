"""Two-stream view of a training step from rocprofv3 --kernel-trace CSVs (config 2): where the main stream waits, what the weight-gradient stream hides,
and how much each main-stream kernel stretches when the second stream runs beside it.
usage: stream_trace.py <trace_overlap_on.csv> [<trace_overlap_off.csv>]"""
import collections
import csv
import sys


def load(path):
    rows = list(csv.DictReader(open(path)))
    out = []
    for r in rows:
        out.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", ""), r.get("Queue_Id", "0")))
    out.sort()
    return out


def steps(iv):
    """Split at stem_fwd_kernel starts; return the list of steps (each a list of intervals), dropping the first and last partial ones."""
    idx = [i for i, k in enumerate(iv) if k[2].startswith("stem_fwd")]
    return [iv[idx[j]:idx[j + 1]] for j in range(len(idx) - 1)]


def main():
    on = steps(load(sys.argv[1]))
    on = on[len(on) // 2:]                      # steady state
    print("steps analysed: %d" % len(on))
    qcount = collections.Counter(k[3] for st in on for k in st)
    mainq = qcount.most_common(1)[0][0]
    print("queues:", dict(qcount), "main =", mainq)
    tot = collections.defaultdict(float)
    for st in on:
        t0, t1 = st[0][0], max(k[1] for k in st)
        main = [k for k in st if k[3] == mainq]
        side = [k for k in st if k[3] != mainq]
        tot["span"] += (t1 - t0) / 1e3
        tot["main_busy"] += sum(e - s for s, e, _, _ in main) / 1e3
        tot["side_busy"] += sum(e - s for s, e, _, _ in side) / 1e3
        # gaps on the main queue and how much of each a side kernel covers
        gaps = 0.0; covered = 0.0; ngap = 0
        for a, b in zip(main, main[1:]):
            g0, g1 = a[1], b[0]
            if g1 <= g0:
                continue
            gaps += (g1 - g0) / 1e3; ngap += 1
            for s, e, _, _ in side:
                lo, hi = max(s, g0), min(e, g1)
                if hi > lo:
                    covered += (hi - lo) / 1e3
        tot["main_gaps"] += gaps; tot["gaps_covered_by_side"] += covered; tot["n_gaps"] += ngap
        # overlap: time both queues have a kernel running
        ov = 0.0
        j = 0
        for s, e, _, _ in side:
            for ms, me, _, _ in main:
                lo, hi = max(s, ms), min(e, me)
                if hi > lo:
                    ov += (hi - lo) / 1e3
        tot["both_running"] += ov
        tot["side_after_main_end"] += max(0.0, (max([k[1] for k in side] or [0]) - max(k[1] for k in main)) / 1e3)
        tot["n_main"] += len(main); tot["n_side"] += len(side)
    n = len(on)
    for k, v in tot.items():
        print("%-24s %9.1f us/step" % (k, v / n) if not k.startswith("n_") else "%-24s %9.1f /step" % (k, v / n))
    if len(sys.argv) > 2:
        off = steps(load(sys.argv[2]))
        off = off[len(off) // 2:]
        def per_name(sts, q=None):
            d = collections.defaultdict(list)
            for st in sts:
                c = collections.Counter()
                for s, e, nm, qq in st:
                    if q is not None and qq != q:
                        continue
                    d[nm].append((e - s) / 1e3)
            return {k: (sum(v) / len(sts), len(v) / len(sts)) for k, v in d.items()}
        a, b = per_name(on), per_name(off)
        print("\nkernel                                  us/step overlap-on  overlap-off   stretch   launches")
        for k in sorted(a, key=lambda k: -a[k][0])[:28]:
            if k in b:
                print("%-40s %10.1f %12.1f %9.2f %9.1f" % (k[:40], a[k][0], b[k][0], a[k][0] / max(b[k][0], 1e-9), a[k][1]))
        print("sum of kernel durations: on %.1f us, off %.1f us; span on %.1f, off %.1f" % (
            sum(v[0] for v in a.values()), sum(v[0] for v in b.values()), tot["span"] / n,
            sum((max(k[1] for k in st) - st[0][0]) / 1e3 for st in off) / len(off)))
        # gap statistics of the single-stream run
        g = []
        for st in off:
            for x, y in zip(st, st[1:]):
                g.append((y[0] - x[1]) / 1e3)
        g.sort()
        print("overlap-off gaps: n/step %.0f  median %.2f us  mean %.2f  p90 %.2f  sum/step %.1f us" % (len(g) / len(off), g[len(g) // 2], sum(g) / len(g), g[int(len(g) * 0.9)], sum(g) / len(off)))


if __name__ == "__main__":
    main()
