#!/bin/bash
# A fuzz soak that finishes: tests/test_gpu_parity.py::test_fuzz_small_graphs (random small inputs: aligned, unaligned and founder rows against the
# oracle) over as many seeds as the box in hand does in BUDGET seconds, sized from a 100-seed pilot's rate with 25 % head-room -- rounds 3 and 4 each
# had one soak cut off by its own time limit because it was sized from another box's rate.  GPU box, repository root.
#   tools/fuzz_soak.sh [BUDGET seconds = 480]      -> gpurun_out/fuzz_soak.txt
BUDGET=${1:-480}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/fuzz_soak.txt
mkdir -p gpurun_out
t0=$(date +%s.%N)
V2M_FUZZ_SEEDS=100 timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k test_fuzz_small_graphs > gpurun_out/fuzz_pilot.txt 2>&1 || { echo "pilot failed"; tail -5 gpurun_out/fuzz_pilot.txt; exit 1; }
t1=$(date +%s.%N)
# (the pilot's time includes interpreter start-up and the library's load, which makes the estimate a conservative one)
SEEDS=$(python3 -c "import sys; dt = $t1 - $t0; rate = 100 / dt; print(max(100, int(0.75 * rate * $BUDGET) // 100 * 100))")
echo "pilot: 100 seeds in $(python3 -c "print('%.1f' % ($t1 - $t0))") s -> $SEEDS seeds for a budget of $BUDGET s with 25 % head-room" | tee $OUT
V2M_FUZZ_SEEDS=$SEEDS timeout -k 10 $((BUDGET + BUDGET / 2)) python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k test_fuzz_small_graphs 2>&1 | tail -4 >> $OUT
echo "soak rc=${PIPESTATUS[0]}" >> $OUT
cat $OUT
