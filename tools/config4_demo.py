#!/usr/bin/env python3
"""BASELINE config 4: the config-3 input with --founder-sequences=25 --minimum-distance=50.
Genotype matrix generated and transposed on the GPU; cut positions + greedy matching on the host (sequential pBWT);
REF + 25 founder rows spliced on the GPU and checked against the CPU oracle's walk with the same cuts."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import vcf2multialign_amd as v2m
from vcf2multialign_amd import _native as N, synth
from vcf2multialign_amd.host import HostGraph
import oracle

cfg = sys.argv[1] if len(sys.argv) > 1 else "config3"
founders, min_dist = 25, 50
ds = synth.dataset(cfg); g = ds.graph
ctx = v2m.Context(0); ctx.upload_graph(g, ds.reference)
dev = torch.device("cuda", 0)
hp, ep = ds.path_cols, ds.path_rows
thr = torch.from_numpy(ds.edge_thresholds.astype(np.int64)).to(torch.int32).to(dev)
src = torch.empty(ep // 64 * hp, dtype=torch.int64, device=dev); dst = torch.empty_like(src)
torch.cuda.synchronize()
ds.fill_paths_device(ctx.stream, src.data_ptr(), thr.data_ptr(), 0, hp)
ctx.profile_enable(True)
ctx.transpose_bits_device(src.data_ptr(), hp, ep, dst.data_ptr())
ctx.set_paths_device(dst.data_ptr(), ep, hp)
ctx.synchronize()
print("%s: %d nodes, %d edges, %d copies; GPU transpose %.3f ms" % (cfg, g.node_count, g.edge_count, ds.n_copies, ctx.profile_get(N.KERNEL_TRANSPOSE)[1]), flush=True)

t = time.time()
hg = HostGraph.from_arrays(g, src.cpu().numpy().view(np.uint64), hp, ep, ds.samples, ds.ploidy)
hg.set_transposed_paths(dst.cpu().numpy().view(np.uint64), ep, hp)     # the GPU transpose's result: lets the search run on several threads
print("host graph (D2H of the two %d MB matrices + copy): %.1f s" % (src.numel() * 8 >> 20, time.time() - t), flush=True)
threads = int(os.environ.get("V2M_FOUNDER_THREADS", "0"))
t = time.time()
res = hg.find_founders(founders, min_dist, keep_ref_edges=False, threads=threads)
t_host = time.time() - t
assert res is not None
cuts, assigned, score = res
rows_per_col = len(cuts) - 1
import zlib
print("crc32 of the cut positions / assigned samples: %08x / %08x" % (zlib.crc32(np.asarray(cuts, dtype=np.uint64).tobytes()), zlib.crc32(np.asarray(assigned, dtype=np.uint32).tobytes())), flush=True)
print("find_cut_positions + find_matchings on the host (%s): %.1f s; %d cut positions, maximum segmentation height %d" % ("1 thread" if threads == 1 else "up to 16 threads" if threads == 0 else "%d threads" % threads, t_host, len(cuts), 1 + score), flush=True)

batch_rows = [v2m.PLOIDY_MAX] + [list(zip(cuts[:-1], assigned[f * rows_per_col:(f + 1) * rows_per_col])) for f in range(founders)]
pitch = ctx.min_row_pitch
out = ctx.alloc_output(len(batch_rows) * pitch, 1)
L = g.aligned_length
batch = v2m.RowBatch(batch_rows)
ctx.splice_rows_device(batch, out, pitch)   # warm-up
ctx.profile_reset()
ctx.splice_rows_device(batch, out, pitch)
sums = ctx.checksum_rows_device(out, pitch, len(batch_rows), length=L)
print("GPU: %d rows x %d bases: resolve %.3f ms + splice %.3f ms" % (len(batch_rows), L, ctx.profile_get(N.KERNEL_RESOLVE)[1], ctx.profile_get(N.KERNEL_SPLICE_ALIGNED)[1]), flush=True)

t = time.time()
og = oracle.graph_from_arrays(g.reference_positions, g.aligned_positions, g.alt_edge_targets, g.alt_edge_count_csum, g.label_offsets, g.label_bytes,
	dst.cpu().numpy().view(np.uint64), ep, hp)
check = [0, 1, founders // 2, founders]
exp = [og.output_sequence(ds.reference) if r == 0 else og.output_sequence(ds.reference, cuts=batch_rows[r]) for r in check]
ok = np.array_equal(sums[check], v2m.checksum_rows_host(exp))
print("oracle walk of rows %s: %.1f s; bit-exact: %s" % (check, time.time() - t, ok), flush=True)
sys.exit(0 if ok else 3)
