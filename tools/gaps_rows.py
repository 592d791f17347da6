import csv, glob, sys, re
f = glob.glob(sys.argv[1] + "/*kernel_trace.csv")[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0].split("<")[0], r.get("Queue_Id"), r.get("Stream_Id"), r.get("Grid_Size"), r.get("Workgroup_Size"), r.get("LDS_Block_Size"), r.get("Scratch_Size"), r.get("VGPR_Count")) for r in csv.DictReader(open(f))]
rows.sort()
i0 = int(len(rows) * 0.7)
t0 = rows[i0][0]
for r in rows[i0:i0 + 14]:
    print("%-16s start %9.2f end %9.2f dur %7.2f  q %s grid %s wg %s lds %s scr %s vgpr %s" % (r[2][:16], (r[0] - t0) / 1e3, (r[1] - t0) / 1e3, (r[1] - r[0]) / 1e3, r[3], r[5], r[6], r[7], r[8], r[9]))
