#!/bin/bash
# The founder search's two kernels at BASELINE config 4 (GPU box, repository root): durations (rocprofv3 --kernel-trace --stats) and, in
# separate --pmc passes, the SQ counters that say what bounds them.  Output under gpurun_out/founder_pmc/ (summary: summary.txt).
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
# Build BEFORE the first profiler line, with no profiler around: under rocprofv3 (--pmc above all) the preload has initialised the GPU before
# python starts, and a compiler launcher started from there would be the exec-after-GPU-init hop this pool forbids.  bench.py / build.py / the
# oracle binding refuse to compile when they find themselves stale under a profiler, so a missed build ends in a message, not in a compile.
python3 -c 'import __graft_entry__ as g; g.build()' > /dev/null
OUT=gpurun_out/founder_pmc
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 tools/founder_kernels_bench.py config3 2 > $OUT/run.txt 2> $OUT/run.err || { tail -5 $OUT/run.err; exit 1; }
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/trace
i=0
for counters in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY"; do
	i=$((i+1))
	timeout -k 10 300 rocprofv3 --pmc $counters --output-format csv -d $OUT/pmc_$i -o p -- python3 tools/founder_kernels_bench.py config3 1 > $OUT/pmc_$i.txt 2> $OUT/pmc_$i.err || { tail -3 $OUT/pmc_$i.err; }
done
python3 - <<'PY' | tee gpurun_out/founder_pmc/summary.txt
import csv, glob, collections
print(open("gpurun_out/founder_pmc/run.txt").read().strip())
for r in csv.DictReader(open("gpurun_out/founder_pmc/kernel_stats.csv")):
	if "pbwt" in r["Name"]:
		print("%-28s %s calls, average %.3f ms" % (r["Name"].split("(")[0].replace("v2m::", ""), r["Calls"], float(r["AverageNs"]) / 1e6))
for kernel in ("pbwt_cut_trials_kernel", "pbwt_cut_records_kernel"):
	tot = collections.defaultdict(float); n = collections.defaultdict(int)
	for f in glob.glob("gpurun_out/founder_pmc/pmc_*/**/*counter_collection.csv", recursive=True):
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
# which sources these figures belong to: git blob hashes of everything libv2m_hip.so is compiled from (as profiles/pmc_traffic.json carries them)
python3 - <<'PY'
import json, sys
sys.path.insert(0, ".")
import bench
json.dump({"_note": "kernel sources (git blob hashes) of the library that tools/founder_pmc.sh measured: kernel_stats.csv and summary.txt next to this file", "kernel_sources": bench.kernel_source_stamp()},
	open("gpurun_out/founder_pmc/provenance.json", "w"), indent=1)
PY
