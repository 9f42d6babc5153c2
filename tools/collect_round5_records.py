#!/usr/bin/env python3
"""Copies what tools/round5_records.sh (part 1) left under gpurun_out/ into profiles/r05/ (and the traffic record into profiles/), checks that the
traffic stamp equals the kernel sources' current git blob hashes, and prints the figures the documents quote.  Build container, repository root."""
import json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
D, R = "profiles/r05", "gpurun_out/refresh"
pairs = [(R + "/bench.json", D + "/config3_1gpu_bench.json"), (R + "/bench_under_rocprof.json", D + "/config3_1gpu_bench_under_rocprof.json"),
	(R + "/bench_under_rocprof_full_command.json", D + "/config3_1gpu_bench_under_rocprof_full_command.json"), (R + "/kernel_stats.csv", D + "/config3_1gpu_kernel_stats.csv"),
	(R + "/kernel_stats_full_command.csv", D + "/config3_1gpu_kernel_stats_full_command.csv"), (R + "/pmc_hbm.json", D + "/config3_1gpu_pmc_hbm.json"),
	(R + "/pmc_traffic.json", "profiles/pmc_traffic.json"), ("gpurun_out/r05r/bench_config5.json", D + "/config5_1gpu_bench.json"),
	("gpurun_out/r05r/bench_n5_rehearsal.json", D + "/bench_n5_config3_rehearsal.json"), ("gpurun_out/r05r/e2e_config3_cli.txt", D + "/e2e_config3_cli.txt"),
	("gpurun_out/r05r/e2e_config4_cli.txt", D + "/e2e_config4_cli.txt"), ("gpurun_out/founder_pmc/summary.txt", D + "/founder_kernels_counters.txt"),
	("gpurun_out/founder_pmc/kernel_stats.csv", D + "/config4_founder_kernel_stats.csv"), ("gpurun_out/founder_pmc/provenance.json", D + "/config4_founder_kernel_stats.provenance.json")]
for a, b in pairs:
	shutil.copyfile(a, b)
stamp = json.load(open("profiles/pmc_traffic.json"))["config3"]["kernel_sources"]
bad = [f for f, h in stamp.items() if subprocess.run(["git", "hash-object", f], stdout=subprocess.PIPE).stdout.decode().strip() != h]
print("traffic stamp:", "equals the tree" if not bad else "DIFFERS for " + ", ".join(bad))
def last(f): return json.loads(open(f).read().strip().splitlines()[-1])
d = last(D + "/config3_1gpu_bench.json"); r, t, u, e, c = d["roofline"], d["roofline_transpose"], d["unaligned"], d["end_to_end"], d["cpu_baseline"]
print("config 3: %.0f Gbases/s, %.2f ms per step; splice %.3f ms = %.1f %%, traffic %.2f GB, memset %.2f TB/s" % (d["value"], d["ms_per_step"], r["avg_launch_ms"], 100 * r["frac"], (r["traffic"] or 0) / 1e9, r["memset_same_buffer_GBs"] / 1e3))
print("  transpose %.4f ms = %.1f %%, dense %.1f %% / %.1f %%, CPU transpose %.2f s" % (t["avg_launch_ms"], 100 * t["frac"], 100 * t["after_timing"]["dense_forward_frac"], 100 * t["after_timing"]["inverse_frac"], t["cpu_baseline"]["seconds"]))
print("  unaligned %s = %.1f %%, aligned same rows %.3f ms -> %.3fx; first 256 rows %.1f %%, %.3fx" % (u["kernels_ms"], 100 * u["roofline"]["frac"], u["aligned_kernel_same_rows_ms"], u["time_per_base_vs_aligned_kernel_same_rows"],
	100 * u["first_rows_only"]["roofline"]["frac"], u["first_rows_only"]["time_per_base_vs_aligned_kernel_same_rows"]))
print("  end to end %.2f GB/s = %.1f %%; CPU %.2f Gbases/s (%.1f on the quota cores); parity %d of %d rows in %s s" % (e["value"], 100 * e["roofline"]["frac"], c["value"], c["rows_dealt_to_threads"]["value"], d["parity"]["rows_checked"], d["parity"]["rows_total"], d["parity"]["oracle_seconds_per_rank"]))
dr = last(D + "/config3_1gpu_bench_under_rocprof.json")
line = [l for l in open(D + "/config3_1gpu_kernel_stats.csv") if "splice_aligned_kernel<true>" in l][0].split('",')[1].split(",")
print("  profiled run: %.0f Gbases/s, HIP events %.3f ms over %d launches, rocprofv3 %.3f ms over %s calls" % (dr["value"], dr["roofline"]["avg_launch_ms"], dr["roofline"]["launches"], float(line[2]) / 1e6, line[0]))
d5 = last(D + "/config5_1gpu_bench.json"); u, t = d5["unaligned"], d5["roofline_transpose"]
print("config 5: %.0f Gbases/s; splice %.3f ms = %.1f %%; transpose %.2f ms = %.1f %%, dense %.1f %% / %.1f %%" % (d5["value"], d5["roofline"]["avg_launch_ms"], 100 * d5["roofline"]["frac"], t["avg_launch_ms"], 100 * t["frac"], 100 * t["after_timing"]["dense_forward_frac"], 100 * t["after_timing"]["inverse_frac"]))
print("  unaligned %s = %.1f %%, aligned same rows %.3f ms -> %.3fx; end to end %.2f GB/s; CPU %.2f Gbases/s" % (u["kernels_ms"], 100 * u["roofline"]["frac"], u["aligned_kernel_same_rows_ms"], u["time_per_base_vs_aligned_kernel_same_rows"], d5["end_to_end"]["value"], d5["cpu_baseline"]["value"]))
dn = last(D + "/bench_n5_config3_rehearsal.json")
print("five ranks: %.0f Gbases/s, %d rows checked, bit-exact %s" % (dn["value"], dn["parity"]["rows_checked"], dn["parity"]["bit_exact"]))
print(" ".join(l.strip() for l in open(D + "/founder_kernels_counters.txt") if "average" in l))
print(open(D + "/e2e_config4_cli.txt").read().strip().splitlines()[-1]); print(open(D + "/e2e_config3_cli.txt").read().strip().splitlines()[-1])
