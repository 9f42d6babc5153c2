#!/bin/bash
# The unaligned stream-out beside the aligned kernel on the same rows under the L2's request counters (round 5, late; the lens that found lines16:
# profiles/r05/transpose_pmc.txt): L1 -> L2 write / read requests, L2 requests / hits / misses, L2 -> memory write requests.  GPU box, repository root.
#   tools/unaligned_req_pmc.sh   -> gpurun_out/unaligned_req/summary.txt
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
python3 -c 'import __graft_entry__ as g; g.build()' > /dev/null || exit 1     # build first, with no profiler around
OUT=gpurun_out/unaligned_req; rm -rf $OUT; mkdir -p $OUT
for cfg in "config5 244" "config3 620"; do
set -- $cfg
i=0
for counters in "TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU"; do
	i=$((i+1))
	timeout -k 10 300 rocprofv3 --pmc $counters --output-format csv -d $OUT/p_$1_$i -o p -- python3 tools/unaligned_bench.py $1 $2 > $OUT/run_$1_$i.txt 2> $OUT/run_$1_$i.err || { echo "pass $i ($counters) failed"; tail -3 $OUT/run_$1_$i.err; }
done
python3 - "$OUT" "$1" <<'PY' | tee -a $OUT/summary.txt
import csv, glob, collections, sys
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(sys.argv[1] + "/p_" + sys.argv[2] + "_*/**/*counter_collection.csv", recursive=True):
	per = collections.defaultdict(float)
	for r in csv.DictReader(open(f)):
		k = "splice_unaligned_kernel" if "splice_unaligned" in r["Kernel_Name"] else "splice_aligned_kernel" if "splice_aligned" in r["Kernel_Name"] else "count_unaligned_kernel" if "count_unaligned" in r["Kernel_Name"] else None
		if k: per[(k, r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
	for (k, d, c), v in per.items():
		tot[k][c] += v; n[k][c] += 1
print(open(sys.argv[1] + "/run_" + sys.argv[2] + "_1.txt").read().strip())
for k in sorted(tot):
	print(sys.argv[2], k)
	for c in sorted(tot[k]): print("   %-28s %16.0f per launch (%d launches)" % (c, tot[k][c] / n[k][c], n[k][c]))
PY
rm -rf $OUT/p_$1_[0-9]
done
