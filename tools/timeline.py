"""Timeline of one training step from a rocprofv3 --kernel-trace CSV: per queue busy time, idle gaps, and the kernels in launch order.
usage: python tools/timeline.py <trace dir> [steps_to_skip]"""
import csv, glob, sys, collections
d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# one step = from one decimate/whiten_stats kernel to the next
marks = [i for i, r in enumerate(rows) if "whiten_stats" in r["Kernel_Name"]]
i0, i1 = marks[-3], marks[-2]
step = rows[i0:i1]
t0 = int(step[0]["Start_Timestamp"])
wall = int(rows[i1]["Start_Timestamp"]) - t0
print("step wall %.1f us, %d kernels" % (wall / 1e3, len(step)))
byq = collections.defaultdict(list)
for r in step:
    byq[r["Queue_Id"]].append(r)
for q, rs in byq.items():
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs)
    print("queue %s: %d kernels, busy %.1f us" % (q, len(rs), busy / 1e3))
# union busy
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in step)
u, cs, ce = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > ce:
        u += ce - cs; cs, ce = s, e
    else:
        ce = max(ce, e)
u += ce - cs
print("union busy %.1f us -> idle %.1f us" % (u / 1e3, (wall - u) / 1e3))
prev_end = {}
for r in step:
    q = r["Queue_Id"]; s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end[q]) / 1e3 if q in prev_end else 0.0
    prev_end[q] = e
    name = r["Kernel_Name"].replace("vm::", "").split("(")[0][:60]
    print("%8.1f q%s gap %6.1f dur %7.1f  %s" % ((s - t0) / 1e3, q, gap, (e - s) / 1e3, name))
