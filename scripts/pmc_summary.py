import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for path in sys.argv[2:]:
    for r in csv.DictReader(open(path)):
        if sys.argv[1] in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:50]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in agg.items():
    print(k)
    for c, x in sorted(v.items()):
        print(f"   {c:28s} {x:18.0f}")
