import csv, glob, sys, collections, re
rows=[]
for f in glob.glob(sys.argv[1]+'/**/*kernel_trace.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
d=collections.defaultdict(list)
for r in rows:
    k=re.sub(r"\(.*$","",r["Kernel_Name"].replace("void ","").replace("(anonymous namespace)::",""))
    d[k].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
tot=sum(sum(v) for v in d.values())
for k,v in sorted(d.items(), key=lambda kv:-sum(kv[1]))[:int(sys.argv[2]) if len(sys.argv)>2 else 25]:
    v2=sorted(v)
    print(f"{k[:70]:70s} n={len(v):5d} med={v2[len(v2)//2]:9.1f} us  sum={sum(v)/1e3:9.2f} ms")
