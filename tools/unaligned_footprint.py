#!/usr/bin/env python3
"""Is the unaligned kernel's extra cost on a 256-row launch the kernel's or the memory's?  
The first 256 config-3 rows through the aligned and the unaligned splice kernel on each of K separately allocated 64-GB buffers
(held at once, so they are different physical memory), three ways:
    packed   rows at the minimum pitch: a 26-GB address range (what bench.py's `first_rows_only` leg does)
    spread   the same rows at a pitch that spreads them over the buffer's whole 64 GB (what a 620-row launch covers)
and, beside them, a pure-write probe of the same buffer (the library's own probe, v2m_alloc_output's criterion, via V2M info).
If the unaligned / aligned ratio follows the buffer and the spread, it is the memory behind the address range; a kernel problem would
show the same ratio everywhere.  Usage: python tools/unaligned_footprint.py [K=4] [rows=256]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import vcf2multialign_amd as v2m
from vcf2multialign_amd import _native as N, synth

K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 256
ds = synth.dataset("config3"); g = ds.graph
ctx = v2m.Context(0); ctx.upload_graph(g, ds.reference)
dev = torch.device("cuda", 0)
hp = 64 * ((rows + 63) // 64)
thr = torch.from_numpy(ds.edge_thresholds.astype(np.int64)).to(torch.int32).to(dev)
src = torch.empty(ds.path_rows // 64 * hp, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
ds.fill_paths_device(ctx.stream, src.data_ptr(), thr.data_ptr(), 0, hp)
ctx.bind_path_matrix_device(src.data_ptr(), hp, ds.path_rows)
ctx.synchronize()
L = g.aligned_length
upitch = (ctx.max_unaligned_length + 255) // 256 * 256
buf_bytes = 627 * ctx.min_row_pitch                       # bench.py's output buffer
spread = buf_bytes // rows // 256 * 256
batch = v2m.RowBatch([v2m.PLOIDY_MAX] + list(range(rows - 1)))
os.environ["V2M_NT_STORES"] = "1"
os.environ["V2M_UNALIGNED_STORE"] = "plain"
bufs = [ctx.alloc_output(buf_bytes, 1) for _ in range(K)]
print("%d rows of config 3; %d buffers of %.1f GB; packed pitch %d (%.1f GB range), spread pitch %d (%.1f GB range)" % (rows, K, buf_bytes / 1e9, upitch, rows * upitch / 1e9, spread, rows * spread / 1e9), flush=True)
ctx.profile_enable(True)


def kernel_ms(out, pitch, unaligned, reps=3):
	ctx.splice_rows_device(batch, out, pitch, unaligned=unaligned)
	ctx.synchronize()
	ctx.profile_reset()
	for _ in range(reps):
		ctx.splice_rows_device(batch, out, pitch, unaligned=unaligned)
	n, ms = ctx.profile_get(N.KERNEL_SPLICE_UNALIGNED if unaligned else N.KERNEL_SPLICE_ALIGNED)
	return ms / n


for i, out in enumerate(bufs):
	ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
	res = {}
	for name, pitch in (("packed", upitch), ("spread", spread)):
		a, u = kernel_ms(out, pitch, False), kernel_ms(out, pitch, True)
		res[name] = (a, u)
	print("buffer %d: packed  aligned %.3f ms  unaligned %.3f ms  ratio %.3f | spread  aligned %.3f ms  unaligned %.3f ms  ratio %.3f" % (
		i, res["packed"][0], res["packed"][1], res["packed"][1] / res["packed"][0], res["spread"][0], res["spread"][1], res["spread"][1] / res["spread"][0]), flush=True)
print("(time per base: the unaligned rows are %.4f of the aligned rows' bytes; ratios above are per launch)" % ((ctx.max_unaligned_length) / L))
