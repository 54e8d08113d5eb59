import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if any(k in r["Name"] for k in ("reduce1","reduce2","accum30")): print(r["Name"][:40].ljust(40), r["Calls"].rjust(4), "%9.3f ms" % (float(r["TotalDurationNs"])/1e6))
