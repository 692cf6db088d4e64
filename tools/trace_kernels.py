"""Per-dispatch durations from a rocprofv3 --kernel-trace --output-format csv run, grouped by (kernel, grid):
python tools/trace_kernels.py <dir> [substr]"""
import collections, csv, glob, sys
d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(list)
order = {}
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if sub and sub not in name:
            continue
        key = (name.split("(")[0][-60:], r.get("Grid_Size", r.get("Grid_Size_X", "")), r.get("Workgroup_Size", r.get("Workgroup_Size_X", "")))
        agg[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        order.setdefault(key, int(r["Start_Timestamp"]))
for key in sorted(agg, key=lambda k: order[k]):
    v = sorted(agg[key])
    print("%-62s grid %-8s wg %-4s n %3d  med %8.1f us  min %8.1f" % (key[0], key[1], key[2], len(v), v[len(v) // 2], v[0]))
