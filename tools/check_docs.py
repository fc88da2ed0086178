#!/usr/bin/env python3
"""The record must equal the measurements: the headline figures DESIGN.md and README.md state -- throughput, demod launch time, decode
step, fraction of the HBM roofline -- are compared with the newest committed bench line profiles/r<NN>_bench_b3584.json and must agree within 1 %
(the roofline fraction within 2 % of its value); every `profiles/...` file a document cites must exist.  Exit code 1 on any mismatch.

The documents mark each headline figure with a fixed phrase (the regular expressions below); a figure that is quoted must follow its phrase, so that
prose elsewhere in the documents can use other numbers (earlier rounds, other workloads) freely.
    python tools/check_docs.py [--verbose]"""
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# phrase -> (regex with one group = the number as written, how to get the measured value from the bench line, relative tolerance)
FIGURES = {
    "throughput": (r"\*\*(\d+(?:\.\d+)?) G IQ samples/s\*\*", lambda d: d["value"] / 1e3, 0.01),
    "demod launch": (r"demod launch (?:takes )?\*\*(\d+(?:\.\d+)?) ms\*\*", lambda d: d["kernel_ms"]["demod"], 0.01),
    "decode step": (r"decode step \*\*(\d+(?:\.\d+)?) ms\*\*", lambda d: d["kernel_ms"]["decode"], 0.01),
    "roofline fraction": (r"\*\*(\d+(?:\.\d+)?) % of the HBM roofline\*\*", lambda d: 100.0 * d["roofline"]["frac"], 0.02),
}


def newest_bench():
    best = None
    for f in glob.glob(os.path.join(ROOT, "profiles", "r*_bench_b3584.json")):
        m = re.match(r"r(\d+)_bench_b3584\.json$", os.path.basename(f))
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), f)
    if best is None:
        raise SystemExit("no profiles/r*_bench_b3584.json")
    return best[1], json.load(open(best[1]))


def main():
    verbose = "--verbose" in sys.argv
    path, d = newest_bench()
    bad = []
    for doc in ("DESIGN.md", "README.md"):
        txt = open(os.path.join(ROOT, doc)).read()
        for name, (rx, get, tol) in FIGURES.items():
            found = re.findall(rx, txt)
            if doc == "DESIGN.md" and not found:
                bad.append(f"{doc}: no '{name}' figure (pattern {rx})")
            want = get(d)
            for s in found:
                ok = abs(float(s) - want) <= tol * abs(want) + 0.051 * (10 ** -(len(s.split('.')[1]) if '.' in s else 0))      # (+ half a unit of the last digit written)
                if verbose or not ok:
                    print(f"{doc}: {name}: written {s}, measured {want:.4g} ({os.path.relpath(path, ROOT)}) {'ok' if ok else 'MISMATCH'}")
                if not ok:
                    bad.append(f"{doc}: {name} {s} != {want:.4g}")
        for ref in sorted(set(re.findall(r"`(profiles/[A-Za-z0-9_./{},*-]+)`", txt))):
            if any(c in ref for c in "{*"):
                continue
            if not os.path.exists(os.path.join(ROOT, ref)):
                bad.append(f"{doc}: cites {ref}, which does not exist")
    if bad:
        print("\n".join(bad))
        return 1
    print(f"check_docs: DESIGN.md and README.md agree with {os.path.relpath(path, ROOT)}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
