import csv, collections, sys
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    if "trunk" in k: print(sys.argv[1].split("/")[-2], k, {c: round(sum(x)/len(x)/1e6,1) for c,x in v.items()})
