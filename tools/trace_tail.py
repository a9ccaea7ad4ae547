import csv,glob,sys
tag=sys.argv[1]; n=int(sys.argv[2]) if len(sys.argv)>2 else 40
f=glob.glob(f"gpurun_out/prof_{tag}/trace/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
rows=[r for r in rows if "copyBuffer" not in r["Kernel_Name"]]
t0=None
for r in rows[-n:]:
    s=int(r["Start_Timestamp"]); e=int(r["End_Timestamp"])
    if t0 is None: t0=s
    print("%9.1f %8.1f  %s" % ((s-t0)/1e3,(e-s)/1e3,r["Kernel_Name"][:70]))
