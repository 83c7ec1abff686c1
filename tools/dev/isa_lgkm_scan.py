"""Triage: linear scan of a hipcc -S dump for counted LGKM waits (s_waitcnt lgkmcnt(N), N > 0) issued while a scalar-memory load may
still be outstanding (SMEM returns out of order: a counted wait is then meaningless for the LDS reads it is supposed to cover).
Control flow is ignored (straight-line approximation): a hit needs a look at the code.  usage: isa_lgkm_scan.py file.s [kernel-substring]"""
import re, sys
path = sys.argv[1]; key = sys.argv[2] if len(sys.argv) > 2 else ""
kern = None; smem = 0; n = 0; hits = {}
for line in open(path):
    m = re.match(r"^(_Z\S+):", line)
    if m:
        kern = m.group(1); smem = 0; n = 0
        continue
    if kern is None or key not in kern:
        continue
    n += 1
    t = line.strip()
    if t.startswith("s_load") or t.startswith("s_buffer_load"):
        smem += 1
    w = re.search(r"lgkmcnt\((\d+)\)", t)
    if w and t.startswith("s_waitcnt"):
        c = int(w.group(1))
        if c == 0:
            smem = 0
        elif smem > 0:
            hits.setdefault(kern, []).append((n, c, smem))
    if t.startswith("s_endpgm"):
        kern = None
for k, v in hits.items():
    print(k[:90], len(v), v[:6])
print("kernels with hits:", len(hits))
