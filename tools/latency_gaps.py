"""tools/latency_gaps.py TRACE_DIR — per-kernel duration and the gap to the previous kernel from a rocprofv3 kernel trace of
tools/latency_trace.py (the last 20 calls)."""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "ssd::" in r["Kernel_Name"]]
names = []
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
    if n not in names:
        names.append(n)
per = len(names)
calls = [rows[i:i + per] for i in range(0, len(rows), per)][-20:]
dur = collections.defaultdict(float); gap = collections.defaultdict(float); tot = 0.0
for c in calls:
    for i, r in enumerate(c):
        n = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        dur[n] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / len(calls)
        if i:
            gap[n] += (int(r["Start_Timestamp"]) - int(c[i - 1]["End_Timestamp"])) / len(calls)
    tot += (int(c[-1]["End_Timestamp"]) - int(c[0]["Start_Timestamp"])) / len(calls)
for n in names:
    print("%-28s %8.2f us   gap before %6.2f us" % (n, dur[n] / 1e3, gap[n] / 1e3))
print("first start -> last end: %.2f us; sum of kernels %.2f us" % (tot / 1e3, sum(dur.values()) / 1e3))
