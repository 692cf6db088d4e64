#!/usr/bin/env python
"""Build profiles/pmc_traffic.json from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, collected separately as
MI355X_MICROARCH.md prescribes):   python tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json>

HBM bytes per launch = 2 * FETCH_SIZE[KB] * 1024  (gfx950 rocprofv3 reports exactly half of a wide coalesced streaming
read) + WRITE_SIZE[KB] * 1024, averaged over the launches of one (kernel, grid) family.  The wgrad launches of the
three conv blocks have distinct grids, which is how a family is mapped back to its conv shape; bench.py looks the dominant
kernel's shape up in this file to fill roofline.traffic."""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def per_order(d, counter, name_has, period):
    """Mean counter value of the i-th (mod period) dispatch of the kernels whose name contains all of name_has."""
    rows = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and any(all(h in r["Kernel_Name"] for h in alt) for alt in name_has):
                rows.append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    rows.sort()
    out = [[] for _ in range(period)]
    for i, (_, v) in enumerate(rows):
        out[i % period].append(v)
    return [sum(v) / len(v) if v else None for v in out]


def per_family(d, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                agg[(r["Kernel_Name"].split("(")[0], int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


def main(fetch_dir, write_dir, out):
    from voicemap_amd import _lib
    lib = _lib.lib()
    fetch, write = per_family(fetch_dir, "FETCH_SIZE"), per_family(write_dir, "WRITE_SIZE")
    res = {"note": "hbm_bytes = 2*FETCH_SIZE + WRITE_SIZE (KB -> bytes); gfx950 FETCH_SIZE counts half of a wide streaming read",
           "kernels": {}}
    n = 256
    for (L, cin, cout) in [(3000, 128, 256), (1500, 256, 384), (750, 384, 512)]:
        # slabs of the plain entry point, or of the folded one (every slab inside one of the two towers)
        grids = []
        for splits in (lib.query("vm_conv_wgrad_splits", n, L, cin, cout),
                       lib.query("vm_conv_wgrad_fold_workspace_bytes", n, n // 2, L, cin, cout) // (3 * cin * cout * 4)):
            grids.append(splits * (-(-cin // 128)) * (-(-cout // 128)) * 512)  # conv_tn9_kernel / conv_tn8x_kernel: (3 taps x 128 ci) x 128 co tiles
        # the folded entry point's grid first (the default step); the plain one only where no launch has that grid -- with the
        # stage-granular splits one shape's plain grid can equal another shape's folded grid (21 x 12 tiles = 42 x 6)
        for want in (grids[1], grids[0]):
            hit = [(name, g) for (name, g) in fetch if ("conv_tn9" in name or "conv_tn8x" in name) and g == want]
            if hit:
                name, g = hit[0]
                fv, wv = fetch[(name, g)], write.get((name, g), 0.0)
                res["kernels"]["vm_conv_wgrad|%d|%d|%d|%d" % (n, L, cin, cout)] = {
                    "fetch_kb": fv, "write_kb": wv, "hbm_bytes": 2 * fv * 1024 + wv * 1024, "grid": g}
                break
    # forward / dgrad launches share one grid size: told apart by dispatch order (forward: blocks 2,3,4; dgrad: 4,3,2)
    shapes = [(3000, 128, 256), (1500, 256, 384), (750, 384, 512)]
    # (rocprofv3 prints some instantiations demangled, with the epilogue enum elided: "<bool _Accum, int, E, 128, false>")
    # (Li3 = the forward on the pool extremes with the BatchNorm affine folded in, vm_conv_fwd_fold: what a default step launches)
    # (round 4: with packed weights the same launches are conv_nt3_kernel<T, EPI, chunks, true>: "Li3ELi<chunks>E" mangled)
    for entry, has, order in (("vm_conv_fwd", [("conv_nt2r", "Li0E"), ("conv_nt2r_kernel<", ", 0>"), ("conv_nt2r_kernel<", "(int)0>"),
                                               ("conv_nt2r", "Li3E"), ("conv_nt2r_kernel<", ", 3>"), ("conv_nt2r_kernel<", "(int)3>"),
                                               ("conv_nt3", "Li3ELi"), ("conv_nt3_kernel<", ", 3, "), ("conv_nt3_kernel<", "(int)3, ")], shapes),
                              ("vm_conv_dgrad", [("conv_nt2r", "Li1E"), ("conv_nt2r_kernel<", ", 1>"), ("conv_nt2r_kernel<", "(int)1>"),
                                                 ("conv_nt3", "Li1ELi"), ("conv_nt3_kernel<", ", 1, "), ("conv_nt3_kernel<", "(int)1, ")],
                               shapes[::-1])):
        fv, wv = per_order(fetch_dir, "FETCH_SIZE", has, 3), per_order(write_dir, "WRITE_SIZE", has, 3)
        for (L, cin, cout), f_, w_ in zip(order, fv, wv):
            if f_ is not None and w_ is not None:
                res["kernels"]["%s|%d|%d|%d|%d" % (entry, n, L, cin, cout)] = {
                    "fetch_kb": f_, "write_kb": w_, "hbm_bytes": 2 * f_ * 1024 + w_ * 1024}
    # everything else: per (kernel, grid) family, for the record
    fam = {}
    for (name, g), fv in fetch.items():
        if "vm" in name:
            wv = write.get((name, g), 0.0)
            fam["%s|grid%d" % (name[-70:], g)] = {"fetch_kb": fv, "write_kb": wv, "hbm_bytes": 2 * fv * 1024 + wv * 1024}
    res["families"] = fam
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    print("wrote", out, "with", len(res["kernels"]), "wgrad shapes")


if __name__ == "__main__":
    main(*sys.argv[1:4])
