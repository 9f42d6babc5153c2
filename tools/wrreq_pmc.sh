cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/wrreq; rm -rf $OUT; mkdir -p $OUT
rocprofv3 -L 2>/dev/null | grep -o -E "TCC_EA0_WRREQ[A-Za-z0-9_]*|TCC_EA0_WR_UNCACHED[A-Za-z0-9_]*|TCC_WRITE[A-Za-z0-9_]*|TCC_EA0_ATOMIC[A-Za-z0-9_]*|TCP_TCC_WRITE[A-Za-z0-9_]*" | sort -u > $OUT/counters.txt
cat $OUT/counters.txt | tr '\n' ' '; echo
for cfg in "config5 244" "config3 620"; do
set -- $cfg
timeout -k 10 300 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $OUT/p_$1 -o p -- python3 tools/unaligned_bench.py $1 $2 > $OUT/run_$1.txt 2> $OUT/run_$1.err || tail -3 $OUT/run_$1.err
python3 - "$OUT/p_$1" "$1" <<'PY'
import csv, glob, collections, sys
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
	per = collections.defaultdict(float)
	for r in csv.DictReader(open(f)):
		k = "unaligned" if "splice_unaligned" in r["Kernel_Name"] else "aligned" if "splice_aligned" in r["Kernel_Name"] else None
		if k: per[(k, r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
	for (k, d, c), v in per.items():
		tot[k][c] += v; n[k][c] += 1
for k in tot:
	print(sys.argv[2], k, {c: round(tot[k][c] / n[k][c]) for c in tot[k]}, "launches", {c: n[k][c] for c in tot[k]})
PY
done
