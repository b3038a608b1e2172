import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*hip_api_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        rows.append((int(r["Start_Timestamp"]), r["Function"], d))
rows.sort()
t0 = rows[0][0]
# only after the ROM set-up: find first hipExtLaunchKernel? print all long calls with time
for s, fn, d in rows:
    if d > 0.3:
        print("%10.3f ms  %-40s %8.3f ms" % ((s - t0) / 1e6, fn, d))
