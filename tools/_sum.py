import csv, re, glob, sys
f = glob.glob(sys.argv[1] + "/*kernel_stats.csv")[0]
n = float(sys.argv[2])
tot = 0
for r in csv.DictReader(open(f)):
    name = re.sub(r"\(anonymous namespace\)::", "", r["Name"]).split("(")[0][:30]
    print("%-30s calls %4d avg %8.2f us" % (name, int(r["Calls"]), float(r["AverageNs"]) / 1e3))
    tot += float(r["TotalDurationNs"])
print("total per call us", tot / n / 1e3)
