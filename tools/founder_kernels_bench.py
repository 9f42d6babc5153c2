#!/usr/bin/env python3
"""BASELINE config 4's founder searches with their chunk walks on the GPU (v2m_pbwt_cut_trials_streamed + v2m_pbwt_cut_records),
REPS times on one context: wall time per run, CRC-32 of cut positions / matchings / score (must not change when the kernels do).
Run it under `rocprofv3 --kernel-trace --stats` for the kernels' durations or `--pmc ...` for their counters
(tools/founder_pmc.sh).  Usage: python tools/founder_kernels_bench.py [config3] [reps]"""
import os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import vcf2multialign_amd as v2m
from vcf2multialign_amd import synth
from vcf2multialign_amd.host import HostGraph

cfg = sys.argv[1] if len(sys.argv) > 1 else "config3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
founders, min_dist = 25, 50
ds = synth.dataset(cfg); g = ds.graph
ctx = v2m.Context(0); ctx.upload_graph(g, ds.reference)
dev = torch.device("cuda", 0)
hp, ep = ds.path_cols, ds.path_rows
thr = torch.from_numpy(ds.edge_thresholds.astype(np.int64)).to(torch.int32).to(dev)
src = torch.empty(ep // 64 * hp, dtype=torch.int64, device=dev); dst = torch.empty_like(src)
torch.cuda.synchronize()
ds.fill_paths_device(ctx.stream, src.data_ptr(), thr.data_ptr(), 0, hp)
ctx.transpose_bits_device(src.data_ptr(), hp, ep, dst.data_ptr())
ctx.set_paths_device(dst.data_ptr(), ep, hp)
ctx.synchronize()
hg = HostGraph.from_arrays(g, src.cpu().numpy().view(np.uint64), hp, ep, ds.samples, ds.ploidy)
hg.set_transposed_paths(dst.cpu().numpy().view(np.uint64), ep, hp)
for rep in range(reps):
	t = time.time()
	if os.environ.get("FOUNDER_BENCH_SEARCH_ONLY"):      # the cut search alone (timing-only kernel variants give cut positions no matching could use)
		res = hg.find_cut_positions_gpu(ctx, min_dist, threads=16)
		print("run %d: cut search alone %.3f s; %s" % (rep, time.time() - t, "no solution" if res is None else "%d cuts, score %d" % (len(res[0]), res[1])), flush=True)
		continue
	cuts, assigned, score = hg.find_founders_gpu(ctx, founders, min_dist, keep_ref_edges=False, threads=16)
	dt = time.time() - t
	print("run %d: %.3f s; %d cuts, score %d, chunks (search GPU, host; matching GPU, host) = %s; crc32 cuts %08x matchings %08x" % (rep, dt, len(cuts), score, hg.gpu_chunks,
		zlib.crc32(np.asarray(cuts, dtype=np.uint64).tobytes()), zlib.crc32(np.asarray(assigned, dtype=np.uint32).tobytes())), flush=True)
