#!/bin/bash
# Round 4's records on the final code (GPU box, repository root); copy what is wanted from gpurun_out/ into profiles/r04/ afterwards.
#   gpurun_out/refresh/*      tools/refresh_profiles.sh: config-3 bench, its profiled pair, PMC traffic (and profiles/pmc_traffic.json)
#   gpurun_out/r04c/*         two-rank rehearsal, config 5, founder kernels (tools/founder_pmc.sh), command-line end to end (configs 3, 4; 3 from VCF text)
set -o pipefail
mkdir -p gpurun_out/r04c
PROFILE_DEST=profiles/r04 bash tools/refresh_profiles.sh > gpurun_out/r04c/refresh.log 2>&1; echo "refresh rc=$?"
cp profiles/pmc_traffic.json gpurun_out/refresh/pmc_traffic.json
timeout -k 10 400 python bench.py --gpus 2 --force-device 0 --output-candidates 1 > gpurun_out/r04c/bench_n2_rehearsal.json 2> gpurun_out/r04c/bench_n2_rehearsal.err; echo "n2 rc=$?"
if [ -z "$SKIP_CONFIG5" ]; then
	timeout -k 10 600 python bench.py --config config5 --steps 2 > gpurun_out/r04c/bench_config5.json 2> gpurun_out/r04c/bench_config5.err; echo "config5 rc=$?"; tail -3 gpurun_out/r04c/bench_config5.err
fi
bash tools/founder_pmc.sh > gpurun_out/r04c/founder_pmc.log 2>&1; echo "founder pmc rc=$?"
V2M_FOUNDER_TIMING=1 timeout -k 10 300 python tools/e2e_cli_config4.py config3 founders > gpurun_out/r04c/e2e_config4_cli.txt 2>&1; echo "e2e config4 rc=$?"; tail -2 gpurun_out/r04c/e2e_config4_cli.txt
timeout -k 10 300 python tools/e2e_cli_config4.py config3 haplotypes > gpurun_out/r04c/e2e_config3_cli.txt 2>&1; echo "e2e config3 rc=$?"; tail -2 gpurun_out/r04c/e2e_config3_cli.txt
TMPDIR=/dev/shm V2M_READER_TIMING=1 timeout -k 10 500 python tools/e2e_cli.py config3 > gpurun_out/r04c/e2e_config3_text_cli.txt 2>&1; echo "e2e config3 text rc=$?"; tail -3 gpurun_out/r04c/e2e_config3_text_cli.txt; rm -f /dev/shm/config3.* 2>/dev/null
