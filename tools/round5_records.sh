#!/bin/bash
# Round 5's records on the final code (GPU box, repository root); copy what is wanted from gpurun_out/ into profiles/r05/ afterwards.
#   gpurun_out/refresh/*   tools/refresh_profiles.sh: config-3 bench, its profiled pair, PMC traffic (and profiles/pmc_traffic.json)
#   gpurun_out/r05r/*      config 5 bench, every row of config 5 against the oracle (both modes), founder kernels, five-rank rehearsal, command-line end to end
# Two parts (a gpurun call lasts 20 minutes at most): `round5_records.sh 1` = refresh, config 5 bench, founder kernels, rehearsal, command line;
# `round5_records.sh 2` = every row of config 5 (both modes: ~17 min of oracle time on 16 threads, which fills a gpurun call of 20 minutes: the fuzz soak, tools/fuzz_soak.sh, is a call of its own).
set -o pipefail
mkdir -p gpurun_out/r05r
PART=${1:-1}
if [ "$PART" = "2" ]; then
V2M_FULL_CONFIG5=both timeout -k 10 1150 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -s -k test_config5_every_row > gpurun_out/r05r/config5_every_row.txt 2>&1; echo "config5 every row rc=$?"; tail -3 gpurun_out/r05r/config5_every_row.txt
exit 0
fi
PROFILE_DEST=profiles/r05 bash tools/refresh_profiles.sh > gpurun_out/r05r/refresh.log 2>&1; echo "refresh rc=$?"
cp profiles/pmc_traffic.json gpurun_out/refresh/pmc_traffic.json
timeout -k 10 600 python bench.py --config config5 --steps 2 --verify-rows 1 > gpurun_out/r05r/bench_config5.json 2> gpurun_out/r05r/bench_config5.err; echo "config5 rc=$?"; tail -3 gpurun_out/r05r/bench_config5.err
bash tools/founder_pmc.sh > gpurun_out/r05r/founder_pmc.log 2>&1; echo "founder pmc rc=$?"
timeout -k 10 400 python bench.py --gpus 5 --force-device 0 --output-candidates 1 --cpu-baseline-rows 64 > gpurun_out/r05r/bench_n5_rehearsal.json 2> gpurun_out/r05r/bench_n5_rehearsal.err; echo "n5 rc=$?"
V2M_FOUNDER_TIMING=1 timeout -k 10 300 python tools/e2e_cli_config4.py config3 founders > gpurun_out/r05r/e2e_config4_cli.txt 2>&1; echo "e2e config4 rc=$?"; tail -2 gpurun_out/r05r/e2e_config4_cli.txt
timeout -k 10 300 python tools/e2e_cli_config4.py config3 haplotypes > gpurun_out/r05r/e2e_config3_cli.txt 2>&1; echo "e2e config3 rc=$?"; tail -2 gpurun_out/r05r/e2e_config3_cli.txt
