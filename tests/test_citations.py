"""Every `file.ext:line[-line]` citation of the reference in our headers, sources, tests and docs must point at
an existing file of /root/reference with that many lines.  Runs where the reference is mounted (the build
container); skipped on the GPU box, which does not have it."""
import collections
import glob
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not mounted here")
def test_reference_citations_resolve():
    files = set()
    for pat in ["include/**/*.h", "include/**/*.hpp", "cudf_amd/**/*.py", "cudf_amd/**/*.hip", "cudf_amd/**/*.hpp",
                "cudf_amd/**/*.cpp", "oracle/*.py", "oracle/*.c", "tests/**/*.py", "tests/**/*.cpp", "DESIGN.md",
                "INTEGRATION.md", "README.md", "bench.py"]:
        files.update(glob.glob(os.path.join(ROOT, pat), recursive=True))
    rx = re.compile(r"([A-Za-z0-9_./-]+\.(?:cu|cuh|hpp|cpp|h|pyx|pxd|py|cmake|java)):(\d+)(?:-(\d+))?")
    index = collections.defaultdict(list)
    for root, _, fs in os.walk(REF):
        if "/.git" in root:
            continue
        for f in fs:
            index[f].append(os.path.join(root, f))
    ours = {os.path.basename(p) for p in glob.glob(os.path.join(ROOT, "**", "*"), recursive=True)}
    checked, bad = 0, []
    for f in sorted(files):
        for m in rx.finditer(open(f, errors="ignore").read()):
            path, last = m.group(1), int(m.group(3) or m.group(2))
            base = os.path.basename(path)
            cands = [p for p in index.get(base, []) if p.endswith(path.lstrip("./"))] or index.get(base, [])
            if not cands:
                if base not in ours:  # a citation of one of our own files is not a reference citation
                    bad.append((os.path.relpath(f, ROOT), m.group(0), "no such reference file"))
                continue
            checked += 1
            if not any(last <= sum(1 for _ in open(c, errors="ignore")) + 2 for c in cands):
                bad.append((os.path.relpath(f, ROOT), m.group(0), "line number beyond the end of the file"))
    assert checked > 300 and not bad, bad[:20]
