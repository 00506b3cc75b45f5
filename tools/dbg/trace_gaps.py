import csv,glob,sys,collections
f=sorted(glob.glob(sys.argv[1]+"/*/*kernel_trace.csv"))[0]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# take the last 1/3 of launches (steady state forwards)
n=len(rows); rows=rows[n//2:]
busy=sum(int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in rows)
span=int(rows[-1]["End_Timestamp"])-int(rows[0]["Start_Timestamp"])
gaps=[int(b["Start_Timestamp"])-int(a["End_Timestamp"]) for a,b in zip(rows,rows[1:])]
import statistics
print("launches",len(rows),"busy ms",busy/1e6,"span ms",span/1e6,"idle %",100*(1-busy/span),"median gap us",statistics.median(gaps)/1e3,"mean gap us",sum(g for g in gaps if g>0)/len(gaps)/1e3)
c=collections.Counter(); t=collections.Counter()
for r in rows:
    k=r["Kernel_Name"][:70]; c[k]+=1; t[k]+=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
for k,v in t.most_common(14): print(f"{v/1e6:8.2f} ms {c[k]:5d} x {v/c[k]/1e3:8.1f} us  {k}")
