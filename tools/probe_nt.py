#!/usr/bin/env python3
"""Why are nontemporal stores bimodal across processes?  Time the splice launch into output buffers at different
base offsets / fresh allocations inside ONE process."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import vcf2multialign_amd as v2m
from vcf2multialign_amd import _native as N, synth

ds = synth.dataset("config3")
g = ds.graph
ctx = v2m.Context(0)
ctx.upload_graph(g, ds.reference)
dev = torch.device("cuda", 0)
rows = 512
hp = 512
thr = torch.from_numpy(ds.edge_thresholds.astype(np.int64)).to(torch.int32).to(dev)
src = torch.empty(ds.path_rows // 64 * hp, dtype=torch.int64, device=dev)
dst = torch.empty_like(src)
pitch = ctx.min_row_pitch
torch.cuda.synchronize()
ds.fill_paths_device(ctx.stream, src.data_ptr(), thr.data_ptr(), 0, hp)
ctx.transpose_bits_device(src.data_ptr(), hp, ds.path_rows, dst.data_ptr())
ctx.set_paths_device(dst.data_ptr(), ds.path_rows, hp)
batch = v2m.RowBatch(list(range(rows)))
ctx.synchronize()
ctx.profile_enable(True)

def t(ptr, p=pitch, nt="1", reps=3):
	os.environ["V2M_NT_STORES"] = nt
	out = []
	for _ in range(reps + 1):
		ctx.profile_reset()
		ctx.splice_rows_device(batch, ptr, p)
		out.append(ctx.profile_get(N.KERNEL_SPLICE_ALIGNED)[1])
	return min(out[1:])

big = torch.empty(rows * pitch + (64 << 20), dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
print("base ptr %x pitch %d" % (big.data_ptr(), pitch))
for off in (0, 256, 4096, 65536, 1 << 20, 2 << 20, 3 << 20, 32 << 20):
	print("offset %9d: nt %.3f ms   plain %.3f ms" % (off, t(big.data_ptr() + off), t(big.data_ptr() + off, nt="0")))
for p in (pitch, pitch + 256, pitch + 4096, pitch + 65536):
	if rows * p <= big.numel():
		print("pitch +%6d: nt %.3f ms   plain %.3f ms" % (p - pitch, t(big.data_ptr(), p), t(big.data_ptr(), p, nt="0")))
del big
torch.cuda.empty_cache()
for i in range(4):
	pad = torch.empty((i + 1) * (37 << 20), dtype=torch.uint8, device=dev)
	buf = torch.empty(rows * pitch, dtype=torch.uint8, device=dev)
	torch.cuda.synchronize()
	print("fresh alloc %d ptr %x: nt %.3f ms   plain %.3f ms" % (i, buf.data_ptr(), t(buf.data_ptr()), t(buf.data_ptr(), nt="0")))
	del buf, pad
	torch.cuda.empty_cache()
