"""Share of a reference file's code lines that appear verbatim (whitespace-stripped) in one of ours.
    python tools/line_overlap.py  (build container only: reads /root/reference)"""
import sys

PAIRS = [("transflow_amd/flow.py", "transflow/flow/sources/source.py"),
         ("transflow_amd/archive.py", "transflow/flow/sources/archive.py"),
         ("transflow_amd/archive.py", "transflow/output/zip.py"),
         ("transflow_amd/archive.py", "transflow/output/numpy.py"),
         ("transflow_amd/archive.py", "transflow/utils.py"),
         ("transflow_amd/masks.py", "transflow/utils.py"),
         ("transflow_amd/compositor.py", "transflow/compositor/compositor.py"),
         ("transflow_amd/config.py", "transflow/config.py")]


def code_lines(path):
    out = []
    for line in open(path, encoding="utf8"):
        s = line.strip()
        if len(s) < 12 or s.startswith(("#", '"""', "import ", "from ")):
            continue
        out.append(s)
    return out


for ours, theirs in PAIRS:
    mine = set(code_lines(ours))
    ref = code_lines("/root/reference/" + theirs)
    hit = sum(1 for l in ref if l in mine)
    print(f"{ours:32s} vs {theirs:40s} {hit:4d} / {len(ref):4d} = {100.0 * hit / max(1, len(ref)):5.1f} %")
