"""Per-kernel totals of every counter found in rocprofv3 --pmc csv outputs (development aid). Usage: pmc_table.py dir..."""
import collections, csv, glob, sys
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            k = "k_" + k.split("k_")[1].split("(")[0][:24] if "k_" in k else k[:24]
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in tot.items():
    if not k.startswith("k_"): continue
    print(k)
    for a, b in sorted(v.items()): print(f"    {a:40s} {b:.4g}")
