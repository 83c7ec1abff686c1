"""Triage: print one kernel's ISA from a hipcc -S dump (build/isa/*.s).  usage: isa_extract.py file.s mangled-name-substring [grep-regex]"""
import re, sys
path, key = sys.argv[1], sys.argv[2]
pat = re.compile(sys.argv[3]) if len(sys.argv) > 3 else None
on = False
n = 0
for line in open(path):
    if not on and line.startswith("_Z") and key in line.split(":")[0] and line.rstrip().split(";")[0].strip().endswith(":"):
        on = True
    if on:
        n += 1
        if pat is None or pat.search(line):
            sys.stdout.write("%6d %s" % (n, line))
        if line.startswith(".Lfunc_end"):
            break
