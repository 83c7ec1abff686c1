"""How much of a training step has NO kernel running: union of the kernel intervals of a rocprofv3 --kernel-trace CSV against the span of the timed steps.
usage: gap_analysis.py <kernel_trace.csv> [skip_first_fraction]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40], r.get("Queue_Id", "")) for r in rows)
n = len(iv)
iv = iv[int(n * float(sys.argv[2]) if len(sys.argv) > 2 else n // 2):]     # the second half of the trace: steady-state steps
t0, t1 = iv[0][0], max(e for _, e, _, _ in iv)
busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
gaps = []
for s, e, nm, q in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append((s - cur_e, nm)); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = t1 - t0
print(f"kernels {len(iv)}  span {span/1e6:.3f} ms  busy (union) {busy/1e6:.3f} ms  idle {100*(span-busy)/span:.1f} %  sum of durations {sum(e-s for s,e,_,_ in iv)/1e6:.3f} ms")
import collections
g = sorted(gaps, reverse=True)
print("gaps: n=%d  median %.2f us  mean %.2f us  p90 %.2f us  largest %s" % (len(g), g[len(g)//2][0]/1e3, sum(x for x,_ in g)/len(g)/1e3, g[len(g)//10][0]/1e3, [(round(x/1e3,1), nm) for x, nm in g[:5]]))
