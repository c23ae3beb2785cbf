import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"]
    if not ("-a" in sys.argv) and ("conv_" in n or "rocclr" in n or "maxpool" in n or "pack_input" in n):
        continue
    short = re.sub(r"\(.*", "", n)[:70]
    print("%9.1f us avg %6s calls  %s" % (float(r["AverageNs"]) / 1e3, r["Calls"], short))
