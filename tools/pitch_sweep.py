#!/usr/bin/env python3
"""Is the rate of a 314-row launch a matter of address footprint or of the row pitch itself?"""
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
out = ctx.alloc_output(1300 * pitch, 1)
ctx.synchronize(); ctx.profile_enable(True)
L = g.aligned_length
def run(rows, p):
	b = v2m.RowBatch(list(range(rows)))
	ts = []
	for rep in range(4):
		ctx.profile_reset(); ctx.splice_rows_device(b, out, p); ts.append(ctx.profile_get(N.KERNEL_SPLICE_ALIGNED)[1])
	return min(ts[1:])
for rows in (314, 128):
	for label, p in (("min pitch", pitch), ("+256", pitch + 256), ("+4096", pitch + 4096), ("+64Ki", pitch + 65536), ("+1Mi", pitch + (1 << 20)), ("+3Mi+256", pitch + (3 << 20) + 256),
			("x1.25", int(pitch * 1.25) // 256 * 256), ("x1.5", int(pitch * 1.5) // 256 * 256), ("x2", pitch * 2), ("x3", pitch * 3), ("x4", pitch * 4), ("128 MiB", 128 << 20), ("96 MiB+256", (96 << 20) + 256)):
		if rows * p > 1300 * pitch or p < pitch: continue
		t = run(rows, p)
		print("%4d rows, pitch %-10s (%11d B, footprint %5.1f GB): %.3f ms = %.0f GB/s" % (rows, label, p, rows * p / 1e9, t, rows * L / t / 1e6))
