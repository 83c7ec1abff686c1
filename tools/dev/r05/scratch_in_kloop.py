"""Where a kernel's scratch (spill) traffic sits relative to its MFMA loop: python tools/dev/r05/scratch_in_kloop.py file.s <mangled-name-substring>"""
import re, sys
lines = open(sys.argv[1]).read().splitlines()
starts = [(k, l.split(':')[0]) for k, l in enumerate(lines) if re.match(r'^_Z\w+:', l)]
for (k0, name), nxt in zip(starts, starts[1:] + [(len(lines), '')]):
    if sys.argv[2] not in name: continue
    body = lines[k0:nxt[0]]
    end = next((k for k, l in enumerate(body) if l.startswith('.Lfunc_end')), len(body))
    body = body[:end]
    mf = [k for k, l in enumerate(body) if 'v_mfma' in l]
    if not mf: continue
    sc = [k for k, l in enumerate(body) if 'scratch_' in l]
    inside = [k for k in sc if mf[0] <= k <= mf[-1]]
    print(name[:62], 'lines', len(body), 'mfma', len(mf), 'range', mf[0], mf[-1], 'scratch ops', len(sc), 'inside the MFMA range', len(inside), inside[:12])
