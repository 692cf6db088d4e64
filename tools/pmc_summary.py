#!/usr/bin/env python
"""Aggregate rocprofv3 --pmc counter_collection.csv files: mean counter value per launch, per (kernel, grid size)."""
import collections
import csv
import glob
import sys


def short(name):
    name = name.split("(")[0]
    for key in ("conv_nt3", "conv_nt2r", "conv_tn9", "conv_tn8x", "conv_nt_glds", "conv_nt_kernel", "conv_tn256", "conv_tn_kernel", "bn_pool_bwd", "bn_drop_pool_gmax",
                "bn_drop_pool_fwd", "conv1_fused_bwd", "conv1_fused_fwd", "global_maxpool", "slab_stage", "colreduce", "whiten", "dense", "adam",
                "siamese"):
        if key in name:
            extra = ""
            if key == "conv_nt3":   # conv_nt3_kernel<T, EPI, chunks, pipe>: mangled ...Li<EPI>ELi<chunks>ELb1E
                import re
                m = re.search(r"Li(\d)ELi(\d+)E", name) or re.search(r", (?:\(int\))?(\d), (?:\(int\))?(\d+),", name)
                epi, ch = (m.group(1), m.group(2)) if m else ("?", "?")
                extra = "<%s, K-side %s ch>" % ({"3": "fwd+fold", "1": "dgrad", "2": "fwd+pool"}.get(epi, epi), int(ch) * 32 if ch != "?" else ch)
            elif "conv_nt" in name:   # mangled: ...ILi0E / Li1E ...; demangled: <T, 0> / <T, 1>
                extra = ("<fwd>" if ("Li0E" in name or ", 0>" in name or "(int)0>" in name) else
                         "<fwd+fold>" if ("Li3E" in name or ", 3>" in name or "(int)3>" in name) else "<dgrad>")
            if "bn_pool_bwd" in name:
                extra = "<apply>" if ("Lb1E" in name or "bool, E>" in name or "apply" in name) else "<reduce>"
            return key + extra
    return name[-40:]


def main(paths):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for pth in paths:
        for f in glob.glob(pth + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                key = (short(r["Kernel_Name"]), r["Grid_Size"])
                agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    counters = sorted({c for v in agg.values() for c in v})
    print(",".join(["kernel", "grid"] + counters))
    rows = []
    for (k, g), cs in agg.items():
        rows.append([k, g] + ["%.4g" % (sum(cs[c]) / len(cs[c])) if c in cs else "" for c in counters])
    rows.sort(key=lambda r: -float(r[2 + counters.index("SQ_WAVE_CYCLES")] or 0) if "SQ_WAVE_CYCLES" in counters else 0)
    for r in rows[:30]:
        print(",".join(r))


if __name__ == "__main__":
    main(sys.argv[1:])
