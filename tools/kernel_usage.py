"""Summarise hipcc's -Rpass-analysis=kernel-resource-usage remarks (grtrans_amd/csrc/kernel_resource_usage.txt or a file /
git blob given as argument) as one line per kernel: VGPRs, AGPRs, scratch bytes per lane, occupancy, LDS."""
import re
import subprocess
import sys


def parse(txt):
    out, cur = {}, None
    for ln in txt.splitlines():
        m = re.search(r"Function Name: (\S+)", ln)
        if m:
            cur = m.group(1)
            out[cur] = {}
            continue
        # "file:l:c: remark:     VGPRs: 256 [-Rpass...]" (plain compile) or "remark: file:l:c:     VGPRs: 256 [-Rpass...]" (-save-temps)
        m = re.search(r"[:\s]\s+([A-Za-z][A-Za-z ]*?)(?: \[[^\]]*\])?: (\S+) \[-Rpass-analysis", ln)
        if m and cur:
            out[cur][m.group(1).strip()] = m.group(2)
    return out


def demangle(n):
    s = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    s = re.sub(r"\(.*$", "", s).replace("void ", "")
    return s


if __name__ == "__main__":
    path = sys.argv[1] if len(sys.argv) > 1 else "grtrans_amd/csrc/kernel_resource_usage.txt"
    if ":" in path and not path.startswith("/"):
        txt = subprocess.run(["git", "show", path], capture_output=True, text=True).stdout
    else:
        txt = open(path).read()
    filt = sys.argv[2] if len(sys.argv) > 2 else ""
    for k, v in parse(txt).items():
        name = demangle(k)
        if filt in name:
            print("%-62s VGPR %3s AGPR %3s scratch %5s occ %s LDS %6s sgprspill %s" % (name[:62], v.get("VGPRs"), v.get("AGPRs"), v.get("ScratchSize"),
                                                                                      v.get("Occupancy"), v.get("LDS Size"), v.get("SGPRs Spill")))
