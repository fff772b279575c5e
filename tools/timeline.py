"""Print the kernel timeline of a rocprofv3 --kernel-trace CSV (start offset, duration, name) for the last N dispatches."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-int(sys.argv[2]):] if len(sys.argv) > 2 else rows
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.3f ms  +%8.3f ms  %s" % ((s - t0) / 1e6, (e - s) / 1e6, r["Kernel_Name"].split("(")[0][:70]))
