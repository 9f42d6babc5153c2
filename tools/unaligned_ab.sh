#!/bin/bash
# A/B of the unaligned stream-out kernels (GPU box, repository root): parity first (the unaligned tests and a short fuzz), then
# tools/unaligned_bench.py on config 5 (244 rows) and config 3 (620 rows), one process per variant and config.
#   V2M_UNALIGNED_KERNEL=wave       every wave packs its own short chunks (rounds 2-5)
#   V2M_UNALIGNED_KERNEL=shared64   one packing pass per workgroup and row tile, a row later, queue of 64
#   (default)                       the same with a queue of 128
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/unaligned_ab; mkdir -p $OUT
if [ "$1" != "noparity" ]; then
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "unaligned or fuzz or more_rows" > $OUT/parity.txt 2>&1; rc=$?; tail -3 $OUT/parity.txt
[ $rc = 0 ] || exit $rc
fi
for cfg in "config5 244" "config3 620"; do
	for v in wave shared shared64 wave shared; do
		set -- $v
		echo "== $cfg V2M_UNALIGNED_KERNEL=$1 (tuning library)"
		V2M_HIP_LIBRARY=$PWD/vcf2multialign_amd/libv2m_hip_tuning.so V2M_UNALIGNED_KERNEL=$1 timeout -k 10 300 python3 tools/unaligned_bench.py $cfg 2>&1 | grep -v amdgpu.ids
	done
done > $OUT/bench.txt 2>&1
cat $OUT/bench.txt
