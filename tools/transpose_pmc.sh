#!/bin/bash
# What the memory side sees of the transposes (round 5, late): L2 <-> memory requests, L2 hits, L1 -> L2 requests, stalls and instruction counts of stream16 and
# lines8 on config 3's dense matrix and on the same matrix padded to line-aligned columns, forward and inverse, each group of counters in its own --pmc pass
# (tools/transpose_pmc_run.py fixes the dispatch order).  GPU box, repository root.  -> gpurun_out/transpose_pmc/summary.txt
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
# build first, with no profiler around (a compiler launcher under rocprofv3's preload is the forbidden exec after GPU init)
python3 -c 'import __graft_entry__ as g; g.build()' > /dev/null || exit 1
OUT=gpurun_out/transpose_pmc; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 200 python3 tools/transpose_pmc_run.py > $OUT/plain_run.txt 2> $OUT/plain_run.err || { tail -5 $OUT/plain_run.err; exit 1; }
rocprofv3 -L 2>/dev/null | grep -o -E "TCC_EA0_(WR|RD)REQ[A-Za-z0-9_]*|TCC_(HIT|MISS|REQ|WRITE|READ|BUSY|TAG_STALL|NORMAL_WRITEBACK|ALL_TC_OP_WB)[A-Za-z0-9_]*|TCP_TCC_[A-Za-z0-9_]*|TCP_PENDING_STALL[A-Za-z0-9_]*|TA_BUSY[A-Za-z0-9_]*" | sort -u | tr '\n' ' ' > $OUT/counters_available.txt
i=0
for counters in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum" "TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum" "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS"; do
	i=$((i+1))
	timeout -k 10 200 rocprofv3 --pmc $counters --output-format csv -d $OUT/pmc_$i -o p -- python3 tools/transpose_pmc_run.py > $OUT/pmc_$i.txt 2> $OUT/pmc_$i.err || { echo "pass $i ($counters) failed"; tail -3 $OUT/pmc_$i.err; }
done
python3 - "$OUT" <<'PY' | tee $OUT/summary.txt
import csv, glob, collections, sys
out = sys.argv[1]
print(open(out + "/plain_run.txt").read().strip())
print("counters the box lists:", open(out + "/counters_available.txt").read().strip())
groups = [("5056x1000000 dense", "stream16"), ("5056x1000000 dense", "lines8"), ("5120x1000448 line-aligned columns", "stream16"), ("5120x1000448 line-aligned columns", "lines8")]
table = collections.defaultdict(dict)
for f in sorted(glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True)):
	per = collections.defaultdict(lambda: collections.defaultdict(float)); names = {}
	for r in csv.DictReader(open(f)):
		if "transpose_bits" in r["Kernel_Name"]:
			per[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"]); names[int(r["Dispatch_Id"])] = r["Kernel_Name"]
	ids = sorted(per)
	if len(ids) != 6 * len(groups):
		print("!! %s: %d transpose dispatches, expected %d" % (f, len(ids), 6 * len(groups))); continue
	for g, (shape, kernel) in enumerate(groups):
		for d, direction in ((0, "forward"), (1, "inverse")):
			mine = [ids[6 * g + 2 * rep + d] for rep in (1, 2)]           # (the first repetition warms up)
			for c in per[mine[0]]:
				table[(shape, kernel, direction)][c] = sum(per[i][c] for i in mine) / len(mine)
			table[(shape, kernel, direction)]["kernel"] = names[mine[0]].split("(")[0][:90]
for key in table:
	print("%s, %s, %s   [%s]" % (key + (table[key].pop("kernel"),)))
	for c in sorted(table[key]): print("   %-28s %16.0f" % (c, table[key][c]))
PY
rm -rf $OUT/pmc_[0-9]
