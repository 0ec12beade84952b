import csv, glob, sys, collections
d = sys.argv[1]
trace = {r["Dispatch_Id"]: r for r in csv.DictReader(open(glob.glob(d + "/**/p_kernel_trace.csv", recursive=True)[0]))}
rows = []
for r in csv.DictReader(open(glob.glob(d + "/**/p_counter_collection.csv", recursive=True)[0])):
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE": continue
    t = trace[r["Dispatch_Id"]]
    us = (int(t["End_Timestamp"]) - int(t["Start_Timestamp"])) / 1e3
    if us < 100: continue
    rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"][:40], us, float(r["Counter_Value"]) / 8 / us / 1e3))
for i, n, us, ghz in sorted(rows):
    print(f"{i:5d} {n:40s} {us:9.1f} us   {ghz:.3f} GHz")
