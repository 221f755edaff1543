#!/usr/bin/env python3
"""Regenerate tests/golden/ctest_cases.json and tests/golden/data/*.mha.

Run in the build container only (needs /root/reference).  It extracts DATA, not
code: the 19 live ADD_TEST argument rows of the reference's
Testing/CMakeLists.txt:10-331 (input, iso, expected #points, expected #cells,
triangle flag, projection flag and the four projection knobs) and byte copies
of the eleven MetaImage volumes those rows name (Data/*.mha).  These are the
only known-answer vectors the reference holds for the hot path
(Testing/CuberilleTest01.cxx:193-204 asserts nothing but the two counts).
"""
import json
import os
import re
import shutil

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    text = open(os.path.join(REF, "Testing", "CMakeLists.txt")).read()
    cases = []
    for block in re.findall(r"^ADD_TEST\((.*?)^\)", text, flags=re.S | re.M):
        toks = []
        for line in block.strip().splitlines():
            line = line.split("#")[0].strip()
            if line:
                toks.append(line)
        name, exe, fn, inp, out = toks[:5]
        iso, npts, ncells, tri, proj = (int(t) for t in toks[5:10])
        thr, step, relax = (float(t) for t in toks[10:13])
        max_steps = int(toks[13])
        assert exe == "CuberilleTest01" and fn == "Test01"
        cases.append(dict(name=name, input=os.path.basename(inp), iso=iso, points=npts,
                          cells=ncells, triangles=tri, project=proj, threshold=thr,
                          step=step, relax=relax, max_steps=max_steps))
    assert len(cases) == 19, len(cases)
    with open(os.path.join(HERE, "ctest_cases.json"), "w") as f:
        json.dump(cases, f, indent=1)
        f.write("\n")
    os.makedirs(os.path.join(HERE, "data"), exist_ok=True)
    for fn in sorted({c["input"] for c in cases}):
        shutil.copyfile(os.path.join(REF, "Data", fn), os.path.join(HERE, "data", fn))
        os.chmod(os.path.join(HERE, "data", fn), 0o644)
    print("wrote", len(cases), "cases")


if __name__ == "__main__":
    main()
