"""Per-dispatch durations of one kernel from a rocprofv3 --kernel-trace csv, in dispatch order: is a long tail in the kernel
stats (max far above the average) the first launches of the process (clock ramp, cold caches), or scattered over the run?
usage: dispatch_times.py <dir or kernel_trace.csv> <kernel-name regex>"""
import csv, glob, os, re, sys
root, pat = sys.argv[1], re.compile(sys.argv[2])
files = [root] if root.endswith(".csv") else glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        if pat.search(r["Kernel_Name"]):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows.sort()
d = [(e - s) / 1e3 for s, e in rows]
if not d:
    sys.exit("no dispatch matches")
t0 = rows[0][0]
print("dispatches %d  avg %.1f us  min %.1f  max %.1f" % (len(d), sum(d) / len(d), min(d), max(d)))
print("first 12 (us): " + " ".join("%.1f" % x for x in d[:12]))
print("last 12  (us): " + " ".join("%.1f" % x for x in d[-12:]))
srt = sorted(d)
print("percentiles: p10 %.1f p50 %.1f p90 %.1f p99 %.1f" % tuple(srt[min(len(srt) - 1, int(len(srt) * q))] for q in (0.1, 0.5, 0.9, 0.99)))
slow = [(i, x, (rows[i][0] - t0) / 1e6) for i, x in enumerate(d) if x > 1.1 * srt[len(srt) // 2]]
print("dispatches more than 10 %% above the median: %d of %d; (index, us, ms since the first dispatch): %s" % (len(slow), len(d), " ".join("(%d, %.1f, %.2f)" % s for s in slow[:40])))
gaps = [(rows[i + 1][0] - rows[i][1]) / 1e3 for i in range(len(rows) - 1)]
if gaps:
    g = sorted(gaps)
    print("gap to the next dispatch of this kernel (us): p10 %.1f p50 %.1f p90 %.1f" % (g[len(g) // 10], g[len(g) // 2], g[len(g) * 9 // 10]))
