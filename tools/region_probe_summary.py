#!/usr/bin/env python3
"""region_probe_summary.py LOG COUNTER_DIR...: per probe_write_kernel dispatch (3 per candidate, in candidate order) the
PMC counters collected by rocprofv3, next to the probe rates the library reported."""
import csv, glob, os, re, sys
from collections import defaultdict
log = open(sys.argv[1]).read()
m = re.search(r"probe write rate \(GB/s\):([0-9 ]+)", log)
rates = [int(x) for x in m.group(1).split()] if m else []
print("rates (GB/s):", rates)
for d in sys.argv[2:]:
	for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
		per = defaultdict(lambda: defaultdict(float))
		for r in csv.DictReader(open(f)):
			if "probe_write_kernel" in r["Kernel_Name"]:
				per[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
		ids = sorted(per)
		names = sorted({k for v in per.values() for k in v})
		print("  dispatch cand " + " ".join("%28s" % n for n in names))
		for i, did in enumerate(ids):
			print("  %8d %4d " % (did, i // 3) + " ".join("%28.0f" % per[did].get(n, 0) for n in names))
