#!/usr/bin/env python3
"""Share of a file's long lines (whitespace-normalised, >= 25 characters) that occur verbatim anywhere under the reference's
libs/ -- the check the round verdicts apply to the host layer.  usage: verbatim_check.py <file>..."""
import glob, os, re, sys
REF = "/root/reference/libs"
norm = lambda l: re.sub(r"\s+", "", l)
ref = set()
for f in glob.glob(os.path.join(REF, "**", "*"), recursive=True):
    if os.path.isfile(f) and f.endswith((".h", ".cpp", ".hpp", ".c", ".py", ".in")):
        try:
            for l in open(f, errors="ignore"):
                n = norm(l)
                if len(n) >= 25:
                    ref.add(n)
        except OSError:
            pass
for p in sys.argv[1:]:
    lines = [norm(l) for l in open(p, errors="ignore")]
    longl = [l for l in lines if len(l) >= 25 and not l.startswith("//")]
    hit = [l for l in longl if l in ref]
    print("%-40s %4d / %4d = %.1f%%" % (p, len(hit), len(longl), 100.0 * len(hit) / max(1, len(longl))))
    if os.environ.get("VERBOSE"):
        for l in hit:
            print("    ", l[:140])
