#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc passes (counter_collection.csv files) into HBM bytes per launch per kernel.

    python tools/pmc_summary.py OUT.json NOTE  WRITE_SIZE=<dir-or-csv>  FETCH_SIZE=<dir-or-csv>  [record=<config>,<batch_rows>,<n_gpus>,<source path>]

With record=..., the splice figure is also written into profiles/pmc_traffic.json, stamped with the git blob hashes of the
kernel sources as they are now (bench.py prints roofline.traffic only while those hashes still match).

Counter unit is KiB.  On gfx950 FETCH_SIZE reports half of what wide coalesced reads move
(/opt/skills/guides/MI355X_MICROARCH.md, HBM section): hbm_bytes = 1024 * (WRITE_SIZE + 2 * FETCH_SIZE)."""
import csv, glob, json, os, re, sys
from collections import defaultdict


def find_csv(path):
	if os.path.isfile(path):
		return path
	hits = sorted(glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True))
	if not hits:
		sys.exit("no counter_collection.csv under " + path)
	return hits[-1]


def short(name):
	return re.sub(r"\(.*$", "", name).replace("void ", "").strip()


def main():
	out_path, note = sys.argv[1], sys.argv[2]
	kernels = defaultdict(dict)
	record = None
	for arg in sys.argv[3:]:
		counter, path = arg.split("=", 1)
		if counter == "record":
			record = path.split(",", 3)
			continue
		per_dispatch = defaultdict(float)     # (dispatch id, kernel) -> value summed over the counter's instances
		with open(find_csv(path)) as f:
			for row in csv.DictReader(f):
				if row["Counter_Name"] != counter:
					continue
				per_dispatch[(row["Dispatch_Id"], short(row["Kernel_Name"]))] += float(row["Counter_Value"])
		sums, counts = defaultdict(float), defaultdict(int)
		for (_, k), v in per_dispatch.items():
			sums[k] += v
			counts[k] += 1
		for k in sums:
			kernels[k][counter + "_KiB_per_launch_avg"] = sums[k] / counts[k]
			kernels[k]["launches"] = counts[k]
	for k, d in kernels.items():
		if "WRITE_SIZE_KiB_per_launch_avg" in d and "FETCH_SIZE_KiB_per_launch_avg" in d:
			d["hbm_bytes_per_launch"] = 1024.0 * (d["WRITE_SIZE_KiB_per_launch_avg"] + 2.0 * d["FETCH_SIZE_KiB_per_launch_avg"])
	result = {"_note": note, "kernels": dict(sorted(kernels.items()))}
	splice = [v for k, v in kernels.items() if k.startswith("v2m::splice_aligned_kernel") and "hbm_bytes_per_launch" in v]
	if splice:
		n = sum(v["launches"] for v in splice)
		result["splice_aligned_kernel_hbm_bytes_per_launch"] = sum(v["hbm_bytes_per_launch"] * v["launches"] for v in splice) / n
	with open(out_path, "w") as f:
		json.dump(result, f, indent=1)
	if record and "splice_aligned_kernel_hbm_bytes_per_launch" in result:
		root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
		sys.path.insert(0, root)
		import bench
		path = os.path.join(root, "profiles", "pmc_traffic.json")
		with open(path) as f:
			rec = json.load(f)
		rec[record[0]] = {"batch_rows": int(record[1]), "n_gpus": int(record[2]), "hbm_bytes_per_launch": result["splice_aligned_kernel_hbm_bytes_per_launch"],
			"source": record[3], "kernel_sources": bench.kernel_source_stamp()}
		with open(path, "w") as f:
			json.dump(rec, f, indent=1)
	print(json.dumps({k: v for k, v in result.items() if k != "kernels"}))


if __name__ == "__main__":
	main()
