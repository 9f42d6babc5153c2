#!/bin/bash
# SQ / LDS counters of the splice kernel on two workloads (GPU box, repository root): where does a dense graph lose?
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/splice_pmc
rm -rf $OUT; mkdir -p $OUT
for cfg in config3 config5; do
	i=0
	for counters in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY"; do
		i=$((i+1))
		timeout -k 10 300 rocprofv3 --pmc $counters --output-format csv -d $OUT/${cfg}_$i -o p -- python3 bench.py --config $cfg --steps 1 --warmup 0 --cpu-baseline-rows 0 --verify-rows 0 --output-candidates 1 > $OUT/${cfg}_$i.json 2> $OUT/${cfg}_$i.err || { tail -3 $OUT/${cfg}_$i.err; }
	done
done
python3 - <<'PY'
import csv, glob, collections
for cfg in ("config3", "config5"):
	tot = collections.defaultdict(float); n = collections.defaultdict(int)
	for f in glob.glob("gpurun_out/splice_pmc/%s_*/**/*counter_collection.csv" % cfg, recursive=True):
		per = collections.defaultdict(float)
		for r in csv.DictReader(open(f)):
			if "splice_aligned_kernel" in r["Kernel_Name"]:
				per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
		for (d, c), v in per.items():
			tot[c] += v; n[c] += 1
	print(cfg)
	for c in sorted(tot): print("   %-28s %16.0f per launch (%d launches)" % (c, tot[c] / n[c], n[c]))
PY
