"""Print registers, spills, LDS and occupancy per kernel from hipcc's -Rpass-analysis=kernel-resource-usage remarks.
usage: hipcc ... --cuda-device-only -Rpass-analysis=kernel-resource-usage -c x.hip 2> remarks.txt; python tools/kernel_resources.py remarks.txt"""
import re, sys
KEEP = {"VGPRs": "vgpr", "AGPRs": "agpr", "TotalSGPRs": "sgpr", "Occupancy [waves/SIMD]": "occ", "SGPRs Spill": "sspill",
        "VGPRs Spill": "vspill", "LDS Size [bytes/block]": "lds", "ScratchSize [bytes/lane]": "scratch"}
for f in sys.argv[1:]:
    cur = None
    for l in open(f):
        m = re.search(r"remark: \s*(.*?)\s*\[-Rpass", l)
        if not m:
            continue
        t = m.group(1)
        if t.startswith("Function Name"):
            if cur:
                print(cur)
            n = t.split(": ")[1]
            m2 = re.search(r"(k_[a-z_0-9]+)INS_6JobLdsI([ht])Li(\d+)ELi(\d+)ELb(\d)ELb(\d)", n)
            cur = ("%s<%s,%s,%s,LR=%s,INL=%s>" % (m2.group(1), "u8" if m2.group(2) == "h" else "u16", *m2.groups()[2:])) if m2 else n[:60]
            cur = "%-44s" % cur
        elif ": " in t and t.split(": ")[0].strip() in KEEP:
            k, v = t.split(": ")
            cur += " %s=%s" % (KEEP[k.strip()], v)
    if cur:
        print(cur)
