#!/bin/bash
# PMC comparison of fast and slow output buffers (GPU box, repository root).
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/region
rm -rf $OUT; mkdir -p $OUT
i=0
for counters in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE" "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum"; do
	i=$((i+1))
	timeout -k 10 200 rocprofv3 --pmc $counters --output-format csv -d $OUT/p$i -o p -- python3 tools/region_probe.py 4 63 > $OUT/log$i.txt 2> $OUT/err$i.txt || { tail -5 $OUT/err$i.txt; exit 1; }
	python3 tools/region_probe_summary.py $OUT/log$i.txt $OUT/p$i
done
