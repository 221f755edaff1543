#!/usr/bin/env python3
"""Per-kernel register / scratch / occupancy table of csrc/cuberille_kernels.hip as hipcc reports it for gfx950
(-Rpass-analysis=kernel-resource-usage).  Usage: python profiles/resource_usage.py [filter-substring]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "midas-journal-740_amd", "csrc", "cuberille_kernels.hip")


def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                          "-fno-fast-math", "-c", SRC, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"],
                         capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in out.splitlines():
        m = re.search(r"remark: (?:\[[^\]]*\] )?\s*([A-Za-z ]+(?:\[[^\]]*\])?): (.*?) \[-Rpass", line)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2).strip()
        if k == "Function Name":
            cur = {"name": v}
            rows.append(cur)
        elif cur is not None:
            cur[k] = v
    names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows),
                           capture_output=True, text=True).stdout.splitlines()
    # (sspill: scalar registers spilled into the lanes of a vector register -- every use inside a loop is a v_readlane; the walk
    #  kernel lost 5 % to 62 of them before it got its forms by geometry, round 5)
    print("%-70s %5s %5s %6s %6s %7s %4s %5s" % ("kernel", "VGPR", "SGPR", "sspill", "vspill", "scratch", "occ", "LDS"))
    for r, n in zip(rows, names):
        n = re.sub(r"^void cuberille::", "", n)
        n = re.sub(r"\(.*$", "", n)
        if flt and flt not in n:
            continue
        print("%-70s %5s %5s %6s %6s %7s %4s %5s" % (n[:70], r.get("VGPRs"), r.get("TotalSGPRs", r.get("SGPRs")), r.get("SGPRs Spill"),
                                                       r.get("VGPRs Spill"), r.get("ScratchSize [bytes/lane]"),
                                                       r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]")))


if __name__ == "__main__":
    main()
