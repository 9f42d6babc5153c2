#!/usr/bin/env python3
"""Splice kernel rate vs rows per launch (what an 8-GPU strong-scaling rank sees: 627 rows = 2 x 314), one process."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import vcf2multialign_amd as v2m
from vcf2multialign_amd import _native as N, synth
ds = synth.dataset("config3"); g = ds.graph
ctx = v2m.Context(0); ctx.upload_graph(g, ds.reference)
dev = torch.device("cuda", 0)
hp = 640
thr = torch.from_numpy(ds.edge_thresholds.astype(np.int64)).to(torch.int32).to(dev)
src = torch.empty(ds.path_rows // 64 * hp, dtype=torch.int64, device=dev); dst = torch.empty_like(src)
torch.cuda.synchronize()
ds.fill_paths_device(ctx.stream, src.data_ptr(), thr.data_ptr(), 0, hp)
ctx.transpose_bits_device(src.data_ptr(), hp, ds.path_rows, dst.data_ptr())
ctx.set_paths_device(dst.data_ptr(), ds.path_rows, hp)
pitch = ctx.min_row_pitch
BIG = int(os.environ.get('SWEEP_ROWS_CAP', '640'))
out = ctx.alloc_output(BIG * pitch, 1 if BIG > 700 else 3)
ctx.synchronize(); ctx.profile_enable(True)
L = g.aligned_length
for rows in (640, 512, 314, 256, 157, 128, 64, 640):
	b = v2m.RowBatch(list(range(rows)))
	ts = []
	for rep in range(4):
		ctx.profile_reset()
		ctx.splice_rows_device(b, out, pitch)
		ts.append((ctx.profile_get(N.KERNEL_SPLICE_ALIGNED)[1], ctx.profile_get(N.KERNEL_RESOLVE)[1]))
	t, r = min(x for x, _ in ts[1:]), min(y for _, y in ts[1:])
	print("%4d rows: splice %.3f ms = %.0f GB/s   resolve %.3f ms" % (rows, t, rows * L / t / 1e6, r))
# same row counts, rows spread over the whole 64-GB buffer (larger pitch): is it the address footprint?
for rows, mult in ((314, 2), (157, 4), (64, 10), (314, 1), (32, 20), (16, 40)) + (((640, 2), (640, 3), (320, 4), (320, 6), (640, 1), (960, 2), (1280, 1), (1920, 1)) if BIG >= 1920 else ()):
	b = v2m.RowBatch(list(range(rows)))
	ts = []
	for rep in range(4):
		ctx.profile_reset()
		ctx.splice_rows_device(b, out, pitch * mult)
		ts.append(ctx.profile_get(N.KERNEL_SPLICE_ALIGNED)[1])
	t = min(ts[1:])
	print("%4d rows at %2d x pitch: splice %.3f ms = %.0f GB/s" % (rows, mult, t, rows * L / t / 1e6))
print(ctx.info)
