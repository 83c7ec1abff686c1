"""Aggregate a bench.py --dump-launches CSV (class,label,us) per conv shape against its HBM / MFMA floor (triage tool)."""
import collections
import csv
import re
import sys


def main(path, steps=2, top=200, cls=None):
    rows = list(csv.DictReader(open(path)))
    agg = collections.OrderedDict()
    for r in rows:
        if cls and r["class"] != cls:
            continue
        a = agg.setdefault((r["class"], r["label"]), [0, 0.0])
        a[0] += 1
        a[1] += float(r["us"])
    out = []
    for (c, l), (n, us) in agg.items():
        m = re.search(r"k(\d+) s(\d).*?cin(\d+) cout(\d+) M(\d+)", l)
        k, s, cin, cout, M = map(int, m.groups())
        taps = k * k if k < 10 else (k // 10) * (k % 10)      # "k21" = 2x1 taps (stride-2 dgrad phases)
        byt = M * (cin + cout) * 2
        fl = 2.0 * M * cin * cout * taps
        if c != "conv_wgrad":
            if s == 2 and "direct" in l and "div1" in l:
                byt = M * (4 * cin + cout) * 2
            if "div2" in l:
                byt = M * (cin / 4 + cout) * 2
                fl /= 4
        elif s == 2:
            byt = M * (4 * cin + cout) * 2
        out.append((us / steps, n // steps, us / n, byt / 6.3e12 * 1e6, fl / 2.0e15 * 1e6, c, l))
    out.sort(reverse=True)
    print("total us/step %.1f" % sum(o[0] for o in out))
    for o in out[:top]:
        print("%8.1f us/step n=%-2d avg=%7.1f hbm=%6.1f mfma=%5.1f x%5.1f  %s" % (o[0], o[1], o[2], o[3], o[4], o[2] / max(o[3], o[4], 1e-9), o[6]))


if __name__ == "__main__":
    main(sys.argv[1], cls=sys.argv[2] if len(sys.argv) > 2 else None)
