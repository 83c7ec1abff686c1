"""Aggregate a --dump-launches CSV by layer label: launches, total us, share, TFLOP/s (conv labels carry k/cin/cout/M)."""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
flt = sys.argv[3] if len(sys.argv) > 3 else ""
agg = collections.defaultdict(lambda: [0, 0.0, 0])
tot = 0.0
for r in rows:
    l, us = r["label"], float(r["us"])
    tot += us
    m = re.search(r"^(\w+) k(\d)(\d) s(\d).*?cin(\d+) cout(\d+) M(\d+) acc(\d)", l)
    if m:
        kind, kh, kw, s, cin, cout, M, acc = m.groups()
        key = "%s %s k%s%s s%s cin%s cout%s M%s acc%s" % (r["class"], kind, kh, kw, s, cin, cout, M, acc)
        fl = 2 * int(M) * int(cout) * int(cin) * int(kh) * int(kw)
    else:
        key, fl = r["class"] + " " + l[:70], 0
    a = agg[key]
    a[0] += 1; a[1] += us; a[2] = fl
print("total %.1f us" % tot)
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if flt and flt not in k:
        continue
    print("%4d x %9.1f us (%4.1f%%) %5.0f TF/s  %s" % (v[0], v[1], 100 * v[1] / tot, v[2] * v[0] / v[1] / 1e6 if v[2] else 0, k))
    top -= 1
    if top <= 0:
        break
