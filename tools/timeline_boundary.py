import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
names=[r["Kernel_Name"] for r in rows]
adam=[i for i,n in enumerate(names) if n.startswith("adam_multi_kernel")]
k=12
t0=int(rows[adam[k-1]]["End_Timestamp"])
for r in rows[adam[k-1]-2:adam[k-1]+int(sys.argv[2])]:
    s,e=(int(r["Start_Timestamp"])-t0)/1e3,(int(r["End_Timestamp"])-t0)/1e3
    print("%8.1f %8.1f q%s %-60s g%s"%(s,e,r["Queue_Id"],r["Kernel_Name"][:60],r["Grid_Size_X"]))
print("step wall", (int(rows[adam[k]]["End_Timestamp"])-t0)/1e3)
