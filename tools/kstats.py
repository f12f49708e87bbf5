"""Print the top rows of a rocprofv3 kernel_stats.csv compactly. usage: python tools/kstats.py FILE [N]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
    n = r["Name"]
    n = n[n.find("::") + 2:][:70] if "anonymous" in n else n[:70]
    print("%-72s %6s %9.1f ms avg %8.1f us %s%%" % (n, r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
