import csv,re,sys,collections
d=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    k=re.sub(r"\(anonymous namespace\)::|void ","",r["Kernel_Name"]).split("(")[0]
    if 'field' in k: d[k].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in d.items(): v=v[10:]; print(f"{k:45s} n={len(v)} avg {sum(v)/len(v):7.1f} min {min(v):7.1f}")
