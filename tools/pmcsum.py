"""Sum rocprofv3 PMC counters per kernel: tools/pmcsum.py DIR [regex]"""
import csv, collections, re, sys, glob
pat = re.compile(sys.argv[2] if len(sys.argv) > 2 else "tehmm")
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void tehmm::", "")
        if pat.search(k):
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[(k, r["Counter_Name"])] += 1
for k, v in agg.items():
    n = max(cnt[(k, c)] for c in v)
    print(k, "(dispatches %d)" % n)
    wc = v.get("SQ_WAVE_CYCLES", 0)
    for c, x in sorted(v.items()):
        print("    %-28s %.4g %s" % (c, x, ("%.1f%%" % (100 * x / wc)) if wc and c.startswith("SQ_") and "INSTS" not in c else ""))
