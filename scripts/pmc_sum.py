"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel name, mean counter value per dispatch."""
import collections, csv, glob, sys
for path in sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        if len(sys.argv) > 2 and sys.argv[2] not in k:
            continue
        print(k)
        for c, v in cs.items():
            print("   %-34s %16.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
