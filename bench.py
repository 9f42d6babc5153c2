#!/usr/bin/env python3
"""bench.py -- aligned A2M Gbases/s of the MI355X haplotype-splice path.

A "step" is one pass of the hot path over this rank's share of the synthetic workload:
transpose the rank's genotype bit matrix (transpose_matrix), then splice every owned row
(REF on rank 0 + the rank's chromosome copies) into aligned A2M row bodies in HBM, in batches that
reuse one device output buffer (the rows of config 3 total 501 GB).  All inputs are resident in HBM
before the timed region starts; nothing is copied to the host inside it.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Haplotypes shard across ranks in blocks of 8 chromosome copies (whole bytes of the bit-packed path matrix) with the
reference and graph replicated; there is no collective on the data path.  The total work is the
named config's and is fixed as N grows, hence "scaling": "strong".

Rank 0 prints ONE JSON line (see README / DESIGN.md for the fields).
"""

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
	if p not in sys.path:
		sys.path.insert(0, p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 achievable)


def log(*a):
	print(*a, file=sys.stderr, flush=True)


def _hip_runtime():
	import ctypes
	# the HIP runtime already loaded by torch (one runtime per process)
	return ctypes.CDLL("libamdhip64.so.7")


def _hip_memset(torch, ptr, nbytes):
	import ctypes
	rt = _hip_runtime()
	rt.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
	rc = rt.hipMemsetAsync(ptr, 45, nbytes, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
	assert rc == 0, "hipMemsetAsync failed"


def _device_bytes(torch, ptr, nbytes):
	import ctypes
	rt = _hip_runtime()
	rt.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
	buf = ctypes.create_string_buffer(nbytes)
	rc = rt.hipMemcpy(buf, ptr, nbytes, 2)   # hipMemcpyDeviceToHost
	assert rc == 0, "hipMemcpy failed"
	return buf.raw


def main():
	ap = argparse.ArgumentParser()
	ap.add_argument("--gpus", type=int, default=1)
	ap.add_argument("--steps", type=int, default=3)
	ap.add_argument("--warmup", type=int, default=1)
	ap.add_argument("--config", default="config3", help="synthetic workload (vcf2multialign_amd/synth.py CONFIGS)")
	ap.add_argument("--batch-rows", type=int, default=0, help="rows per splice launch (one device output buffer of this many rows is reused); 0 = as many as fit --batch-gb")
	ap.add_argument("--batch-gb", type=float, default=64.0, help="size of the reused device output buffer when --batch-rows is 0: launches that write a ~64-GB address range reach the full HBM write rate (DESIGN.md section 4)")
	ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for a one-GPU rehearsal of N > 1)")
	ap.add_argument("--force-device", type=int, default=None, help="rehearsal only: put every rank on this HIP device")
	ap.add_argument("--output-candidates", type=int, default=4, help="device buffers v2m_alloc_output may hold at once to choose the output buffer from (as many as fit are tried; 1 = plain allocation)")
	ap.add_argument("--cpu-baseline-rows", type=int, default=320, help="haplotypes (plus REF) the CPU oracle is timed on (320 rows of config 3 = 32 Gbases, about 11 s on one core); 0 disables")
	ap.add_argument("--verify-rows", type=int, default=3, help="rows of the last batch checked against the CPU oracle after timing; 0 disables")
	args = ap.parse_args()

	rank = int(os.environ.get("RANK", "0"))
	local_rank = int(os.environ.get("LOCAL_RANK", "0"))
	world = int(os.environ.get("WORLD_SIZE", "1"))
	if world != args.gpus:
		if world == 1 and args.gpus > 1:
			sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
		sys.exit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))

	import torch
	import torch.distributed as dist

	dev_index = local_rank if args.force_device is None else args.force_device
	torch.cuda.set_device(dev_index)
	dev = torch.device("cuda", dev_index)
	red_dev = dev if args.dist_backend == "nccl" else torch.device("cpu")   # where the timing / parity reductions live
	if world > 1:
		os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
		if args.dist_backend == "nccl":
			dist.init_process_group("nccl", device_id=dev)
		else:
			dist.init_process_group(args.dist_backend)

	from vcf2multialign_amd import build as _build
	if local_rank == 0:
		_build.build_native()   # no-op when the in-tree libraries are newer than their sources (hipcc cross-compiles gfx950)
	if world > 1:
		dist.barrier()          # nobody loads the libraries before the (possible) rebuild is over
	import vcf2multialign_amd as v2m
	from vcf2multialign_amd import _native as N
	from vcf2multialign_amd import synth
	from vcf2multialign_amd.sharding import max_over_ranks, shard_copies

	# ---- workload: generated on every rank (deterministic), resident in HBM before timing ----------
	t0 = time.time()
	ds = synth.dataset(args.config)
	g = ds.graph
	L, R, NN, E = g.aligned_length, len(ds.reference), g.node_count, g.edge_count
	H = ds.n_copies
	Ep = ds.path_rows
	if rank == 0:
		log("[bench] %s: R=%d variants/edges=%d nodes=%d L=%d copies=%d (generated in %.1fs)" % (args.config, R, E, NN, L, H, time.time() - t0))

	ctx = v2m.Context(dev_index)
	ctx.upload_graph(g, ds.reference)
	pitch = ctx.min_row_pitch

	c0, c1, hp_local = shard_copies(H, world, rank)
	# Matrix dimensions padded to multiples of 1024 bits: every column then starts on a 128-B line and the transpose moves
	# whole lines (0.31 ms instead of 0.43 ms on the config-3 matrix).  The reference pads to 64 (variant_graph.cc:277,449);
	# the ABI takes any multiple of 64, padding rows and columns are zero.
	pad1024 = lambda n: (n + 1023) // 1024 * 1024
	hp_local = pad1024(hp_local) if hp_local else 0
	Ep_alg = Ep            # 64 * ceil(E / 64): what the algorithmic-byte formula and the CPU oracle use
	Ep = pad1024(Ep)
	n_local_copies = c1 - c0
	rows = ([v2m.PLOIDY_MAX] if rank == 0 else []) + list(range(n_local_copies))   # local copy indices into this rank's matrix
	n_rows = len(rows)
	total_rows = H + 1

	# equal-sized batches of at most --batch-rows rows (5009 rows -> 10 x 501 rather than 9 x 512 + 401)
	max_batch_rows = args.batch_rows if args.batch_rows > 0 else max(1, int(args.batch_gb * 1e9) // pitch)
	n_batches = max(1, -(-n_rows // max(1, max_batch_rows)))
	batch_rows = max(1, -(-n_rows // n_batches))
	# Output buffer: placement matters on this hardware, so the library picks it by measurement.
	out_bytes = batch_rows * pitch
	out_ptr = ctx.alloc_output(out_bytes, candidates=args.output_candidates)
	thr = torch.from_numpy(ds.edge_thresholds.astype(np.int64)).to(torch.int32).to(dev) if E else torch.zeros(1, dtype=torch.int32, device=dev)
	words = Ep // 64 * hp_local
	paths_src = torch.empty(max(words, 1), dtype=torch.int64, device=dev)   # paths_by_edge_and_chrom_copy (this rank's copies x Ep)
	paths_dst = torch.empty(max(words, 1), dtype=torch.int64, device=dev)   # paths_by_chrom_copy_and_edge (Ep x this rank's copies)
	torch.cuda.synchronize()
	if hp_local:
		ds.fill_paths_device(ctx.stream, paths_src.data_ptr(), thr.data_ptr(), copy_base=c0, n_rows=hp_local, n_cols=Ep, copy_end=c1)
	ctx.synchronize()

	batches = [v2m.RowBatch(rows[i:i + batch_rows]) for i in range(0, n_rows, batch_rows)]

	def step():
		if hp_local:
			ctx.transpose_bits_device(paths_src.data_ptr(), hp_local, Ep, paths_dst.data_ptr())
			ctx.set_paths_device(paths_dst.data_ptr(), Ep, hp_local)
		for b in batches:
			ctx.splice_rows_device(b, out_ptr, pitch)

	def fence():
		ctx.synchronize()
		torch.cuda.synchronize()
		if world > 1:
			dist.barrier()

	for _ in range(args.warmup):
		step()
	fence()
	ctx.profile_enable(True)
	ctx.profile_reset()
	t_begin = time.perf_counter()
	for _ in range(args.steps):
		step()
	ctx.synchronize()
	torch.cuda.synchronize()
	elapsed = time.perf_counter() - t_begin
	if world > 1:
		dist.barrier()
		elapsed = max_over_ranks(elapsed, dist, red_dev)
	ctx.profile_enable(False)

	# ---- roofline of the dominant kernel, from HIP events on the kernel's own stream ---------------
	launches, splice_ms = ctx.profile_get(N.KERNEL_SPLICE_ALIGNED)
	per_launch = ctx.profile_launches(N.KERNEL_SPLICE_ALIGNED)
	if rank == 0 and os.environ.get("V2M_BENCH_LOG_LAUNCHES"):
		log("[bench] splice launches (ms): " + " ".join("%.2f" % x for x in per_launch))
	_, resolve_ms = ctx.profile_get(N.KERNEL_RESOLVE)
	_, transpose_ms = ctx.profile_get(N.KERNEL_TRANSPOSE)
	label_bytes = len(g.label_bytes)
	# algorithmic bytes of one launch over Hb rows (SURVEY.md 8d / BASELINE.md):
	#   Hb*L + Hb*Ep/8 + R + 24*N + 8*E + sum|label|
	alg_bytes_total = 0
	for b in batches:
		alg_bytes_total += b.n_rows * L + b.n_rows * Ep_alg // 8 + R + 24 * NN + 8 * E + label_bytes
	alg_bytes_per_launch = alg_bytes_total / max(1, len(batches))
	avg_launch_s = (splice_ms / 1e3) / max(1, launches)
	achieved = alg_bytes_per_launch / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0

	value = total_rows * L * args.steps / elapsed / 1e9

	# HBM traffic of the dominant kernel comes from separate rocprofv3 --pmc passes of this same command
	# (profiles/pmc_traffic.json); it is only quoted when the run matches the profiled configuration.
	traffic, traffic_source = None, None
	try:
		with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
			rec = json.load(f).get(args.config)
		if rec and rec["batch_rows"] == batch_rows and rec["n_gpus"] == world:
			traffic, traffic_source = rec["hbm_bytes_per_launch"], rec["source"]
	except (OSError, ValueError, KeyError):
		pass

	result = {
		"metric": "aligned A2M Gbases/sec",
		"value": round(value, 3),
		"unit": "Gbases/s",
		"n_gpus": world,
		"steps": args.steps,
		"warmup": args.warmup,
		"ms_per_step": round(1e3 * elapsed / args.steps, 3),
		"higher_is_better": True,
		"scaling": "strong",
		"vs_baseline": None,
		"dtype": "u8",
		"data": "synthetic",
		"config": {
			"workload": "%s: synthetic %d bp reference, %d variant records (%d ALT edges), %d diploid samples = %d haplotype rows + REF, --haplotypes aligned A2M, L=%d"
				% (args.config, R, ds.n_variants, E, ds.samples, H, L),
			"rows_total": total_rows, "aligned_length": L, "batch_rows": batch_rows,
			"sharding": "contiguous chromosome copies per rank (multiples of 8), graph + reference replicated, no collective",
			"tuning": ctx.info,
		},
		"roofline": {
			"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
			"frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_unit": "bytes per launch (PMC: WRITE_SIZE + 2*FETCH_SIZE)", "traffic_source": traffic_source,
			"kernel": "splice_aligned_kernel", "launches": launches, "avg_launch_ms": round(1e3 * avg_launch_s, 4),
			"algorithmic_bytes_per_launch": int(alg_bytes_per_launch),
			"other_kernels_ms_per_step": {"resolve_effective_edges_kernel": round(resolve_ms / args.steps, 3), "transpose_bits_kernel": round(transpose_ms / args.steps, 3)},
		},
	}

	# Context for the roofline number: what a plain device memset of the very same output buffer reaches in this
	# process (the achievable write rate varies by +-10 % between processes / boxes).
	if rank == 0:
		ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
		fills = []
		for _ in range(3):
			ev0.record()
			_hip_memset(torch, out_ptr, out_bytes)
			ev1.record()
			torch.cuda.synchronize()
			fills.append(ev0.elapsed_time(ev1))
		result["roofline"]["memset_same_buffer_GBs"] = round(out_bytes / (min(fills) * 1e-3) / 1e9, 1)

	# ---- CPU oracle: every rank checks sampled rows of its own shard; rank 0 at N=1 times the baseline ----
	def oracle_graph(copies):
		"""Oracle graph whose path matrix holds the CPU re-derivation (genotype hash) of the given global copies."""
		import oracle
		n_cols = 64 * ((len(copies) + 63) // 64)
		cols = [ds.copy_column(c) for c in copies] + [np.zeros(Ep_alg // 64, np.uint64)] * (n_cols - len(copies))
		return oracle.graph_from_arrays(g.reference_positions, g.aligned_positions, g.alt_edge_targets, g.alt_edge_count_csum,
			g.label_offsets, g.label_bytes, np.concatenate(cols) if Ep_alg else np.zeros(0, np.uint64), Ep_alg, n_cols,
			["S%d" % i for i in range(n_cols // ds.ploidy)], np.arange(n_cols // ds.ploidy + 1, dtype=np.uint32) * ds.ploidy)

	if args.verify_rows:
		k = min(args.verify_rows, batches[0].n_rows) if batches else 0
		ok = True
		if k:
			# re-run the first batch of this rank and compare device checksums + one full row with the oracle
			ctx.splice_rows_device(batches[0], out_ptr, pitch)
			sums = ctx.checksum_rows_device(out_ptr, pitch, k, length=L)
			local = rows[:k]
			copies = [c0 + r for r in local if r != v2m.PLOIDY_MAX]
			og = oracle_graph(copies)
			col_of = {c: i for i, c in enumerate(copies)}
			bodies = [og.output_sequence(ds.reference) if r == v2m.PLOIDY_MAX else og.output_sequence(ds.reference, copy_index=col_of[c0 + r]) for r in local]
			ok = bool(np.array_equal(sums, v2m.checksum_rows_host(bodies))) and _device_bytes(torch, out_ptr + (k - 1) * pitch, L) == bodies[k - 1]
		checked = k
		if world > 1:
			flags = torch.tensor([1 if ok else 0, checked], dtype=torch.int64, device=red_dev)
			lo = flags.clone()
			dist.all_reduce(lo, op=dist.ReduceOp.MIN)
			dist.all_reduce(flags, op=dist.ReduceOp.SUM)
			ok, checked = bool(lo[0].item()), int(flags[1].item())
		result["parity"] = {"rows_checked": checked, "bit_exact": ok, "method": "per rank: device row checksums + one full row of its first batch vs the CPU oracle"}
		if not ok:
			log("[bench] PARITY FAILURE against the CPU oracle")

	if rank == 0 and world == 1 and args.cpu_baseline_rows:
		nb = min(args.cpu_baseline_rows, H)
		og = oracle_graph(list(range(nb)))
		nbytes, secs = og.haplotype_output_a2m(ds.reference, None, first_copy=0, n_copies=nb)
		bases = (nb + 1) * L
		result["cpu_baseline"] = {
			"value": round(bases / secs / 1e9, 4), "unit": "Gbases/s", "cores": 1, "kind": "port",
			"sample": "REF + first %d haplotypes of %s (%.2f Gbases) through the oracle's output_sequence/haplotype_output_a2m into a discarding std::ostream, %.1f s; rows are independent, so the rate carries to the full %d rows; host has %d logical CPUs"
				% (nb, args.config, bases / 1e9, secs, total_rows, os.cpu_count()),
		}

	if rank == 0:
		print(json.dumps(result), flush=True)
	if world > 1:
		dist.barrier()
		dist.destroy_process_group()
	ctx.free_output(out_ptr)
	ctx.close()
	if result.get("parity", {}).get("bit_exact") is False:
		sys.exit(3)


if __name__ == "__main__":
	main()
