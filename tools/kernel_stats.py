"""Collapse a rocprofv3 --stats kernel_stats.csv by kernel family (template instantiations merged) -> ms per step."""
import collections
import csv
import re
import sys


def main(path, steps):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        name = re.sub(r"<.*", "", r["Name"]).replace("void ", "").split("(")[0]
        agg[name][0] += int(r["Calls"])
        agg[name][1] += float(r["TotalDurationNs"])
    tot = sum(v[1] for v in agg.values())
    print("| kernel | launches/step | ms/step | avg us | % |\n|---|---|---|---|---|")
    for k, (n, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        if ns / tot < 0.002:
            continue
        print("| `%s` | %.1f | %.3f | %.1f | %.1f |" % (k, n / steps, ns / steps / 1e6, ns / n / 1e3, 100 * ns / tot))
    print("total %.3f ms/step" % (tot / steps / 1e6))


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]))
