"""Per-dispatch durations from a rocprofv3 --kernel-trace --output-format csv run, grouped by (kernel, grid):
python tools/trace_kernels.py <dir> [substr] [period]
With a period (e.g. 3 for the three conv blocks of a step) launches of one kernel are additionally split by their position
in the launch order modulo the period -- forward/dgrad launches of different layers share one grid size."""
import collections, csv, glob, sys
d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
period = int(sys.argv[3]) if len(sys.argv) > 3 else 0
agg = collections.defaultdict(list)
order = {}
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    rows += [r for r in csv.DictReader(open(f)) if not sub or sub in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seen = collections.Counter()
for r in rows:
    name = r["Kernel_Name"]
    key = (name.split("(")[0][-60:], r.get("Grid_Size", r.get("Grid_Size_X", "")), r.get("Workgroup_Size", r.get("Workgroup_Size_X", "")))
    if period:
        seen[key] += 1
        key = key + ("#%d" % ((seen[key] - 1) % period),)
    agg[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    order.setdefault(key, int(r["Start_Timestamp"]))
for key in sorted(agg, key=lambda k: order[k]):
    v = sorted(agg[key])
    tag = key[3] if len(key) > 3 else ""
    print("%-62s grid %-8s wg %-4s %-3s n %3d  mean %8.1f us  med %8.1f  min %8.1f" % (key[0], key[1], key[2], tag, len(v), sum(v) / len(v),
                                                                                   v[len(v) // 2], v[0]))
