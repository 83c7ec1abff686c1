import sys
import numpy as np
a = np.load(sys.argv[1]); b = np.load(sys.argv[2])
rows = []
for k in a.files:
    d = np.abs(a[k] - b[k]).max(); mx = np.abs(b[k]).max()
    if d > 0:
        rows.append((d / max(mx, 1e-30), k, d, mx))
rows.sort(reverse=True)
print(len(rows), "of", len(a.files), "differ")
for r in rows[:25]:
    print("%.3e %s %.3e %.3e" % r)
same = [k for k in a.files if not np.abs(a[k] - b[k]).max() > 0]
print("identical:", " ".join(same))
