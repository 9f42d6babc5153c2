#!/bin/bash
# Re-creates the judged profile set of bench.py's default run (config 3, one GPU) under gpurun_out/refresh/:
#   pmc_hbm.json                     HBM bytes per launch from two separate --pmc passes (one step each)
#   pmc_traffic.json                 the record bench.py reads roofline.traffic from, stamped with the kernel sources' hashes
#   bench.json                       plain run (after the counter passes, so that it carries the traffic figure)
#   bench_under_rocprof.json         the same command minus the legs that launch the aligned kernel on other row counts (--e2e-gb 0 --unaligned-rows 0)
#                                    under rocprofv3 --kernel-trace --stats
#   kernel_stats.csv                 its per-kernel summary (the one roofline.avg_launch_ms must agree with)
#   bench_under_rocprof_full_command.json, kernel_stats_full_command.csv   the whole default command under the profiler (every kernel of every leg)
# Run on the GPU box from the repository root; copy the files into profiles/rNN/ (and pmc_traffic.json into profiles/) afterwards.
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
# Build BEFORE the first profiler line, with no profiler around: under rocprofv3 (--pmc above all) the preload has initialised the GPU before
# python starts, and a compiler launcher started from there would be the exec-after-GPU-init hop this pool forbids.  bench.py / build.py / the
# oracle binding refuse to compile when they find themselves stale under a profiler, so a missed build ends in a message, not in a compile.
python3 -c 'import __graft_entry__ as g; g.build()' > /dev/null
OUT=gpurun_out/refresh
mkdir -p $OUT
# the counter passes run with the store flavour fixed, so that every splice launch they see is one of the step's own
# (the flavour calibration would add four same-size launches, two of them with plain stores)
PMC_ARGS="--steps 1 --warmup 0 --cpu-baseline-rows 0 --verify-rows 0 --unaligned-rows 0 --transpose-extras 0 --cpu-transpose 0 --e2e-gb 0"
V2M_NT_STORES=1 timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_w -o w -- python3 bench.py $PMC_ARGS > $OUT/pmc_w.json 2> $OUT/pmc_w.err
echo "pmc write done"
V2M_NT_STORES=1 timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_f -o f -- python3 bench.py $PMC_ARGS > $OUT/pmc_f.json 2> $OUT/pmc_f.err
echo "pmc fetch done"
python3 tools/pmc_summary.py $OUT/pmc_hbm.json "rocprofv3 --pmc WRITE_SIZE and --pmc FETCH_SIZE (separate passes, one step of bench.py's default run each; counter unit KiB; hbm_bytes = 1024*(WRITE_SIZE + 2*FETCH_SIZE), the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md). Store flavour fixed (V2M_NT_STORES=1): the splice average covers the step's own launches only." WRITE_SIZE=$OUT/pmc_w FETCH_SIZE=$OUT/pmc_f "record=config3,$(python3 -c "import json;print(json.loads(open('$OUT/pmc_w.json').read().strip().splitlines()[-1])['config']['batch_rows'])"),1,${PROFILE_DEST:-profiles/r04}/config3_1gpu_pmc_hbm.json"
cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
rm -rf $OUT/pmc_w $OUT/pmc_f
timeout -k 10 400 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench done"
# The judged per-kernel summary: the timed region's own launches.  Two legs of the default command launch splice_aligned_kernel on other
# row counts -- the end-to-end leg ~500 times on the sink path's 5-row slices, the unaligned leg's like-for-like yardstick on 256 and 620 rows --
# and would average into the figure, so this run leaves them out (--e2e-gb 0 --unaligned-rows 0): every splice_aligned_kernel<true> launch it
# sees writes 626-627 rows, as the launches behind roofline.avg_launch_ms do.  A second summary of the whole default command sits beside it.
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 bench.py --e2e-gb 0 --unaligned-rows 0 > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/trace
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 bench.py > $OUT/bench_under_rocprof_full_command.json 2> $OUT/trace_full.err
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_full_command.csv
rm -rf $OUT/trace
echo "trace done"
tail -1 $OUT/bench.json | cut -c1-300
