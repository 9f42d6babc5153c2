#!/bin/bash
# SQ counters of splice_unaligned_kernel beside splice_aligned_kernel on the same rows (tools/unaligned_bench.py), in separate --pmc
# passes: is the dense graph's --unaligned leg (config 5) bound by VALU issue?  GPU box, repository root.
#   tools/unaligned_pmc.sh [config5] [244]
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
# build first, with no profiler around (a compiler launcher under rocprofv3's preload is the forbidden exec after GPU init)
python3 -c 'import __graft_entry__ as g; g.build()' > /dev/null
CFG=${1:-config5}; ROWS=${2:-244}
OUT=gpurun_out/unaligned_pmc
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 python3 tools/unaligned_bench.py $CFG $ROWS > $OUT/plain_run.txt 2> $OUT/plain_run.err || { tail -5 $OUT/plain_run.err; exit 1; }
i=0
for counters in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_THREAD_CYCLES_VALU SQ_IFETCH"; do
	i=$((i+1))
	timeout -k 10 300 rocprofv3 --pmc $counters --output-format csv -d $OUT/pmc_$i -o p -- python3 tools/unaligned_bench.py $CFG $ROWS > $OUT/pmc_$i.txt 2> $OUT/pmc_$i.err || { echo "pass $i ($counters) failed"; tail -3 $OUT/pmc_$i.err; }
done
python3 - "$OUT" <<'PY' | tee $OUT/summary.txt
import csv, glob, collections, sys
out = sys.argv[1]
print(open(out + "/plain_run.txt").read().strip())
for kernel in ("splice_unaligned", "splice_aligned_kernel", "count_unaligned_kernel"):
	tot = collections.defaultdict(float); n = collections.defaultdict(int)
	for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
		per = collections.defaultdict(float)
		for r in csv.DictReader(open(f)):
			if kernel in r["Kernel_Name"]:
				per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
		for (d, c), v in per.items():
			tot[c] += v; n[c] += 1
	print(kernel)
	for c in sorted(tot): print("   %-28s %16.0f per launch (%d launches)" % (c, tot[c] / n[c], n[c]))
PY
rm -rf $OUT/pmc_[0-9]
