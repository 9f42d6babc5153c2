#!/usr/bin/env python3
"""Does the nontemporal-store slow mode come from sustained back-to-back launches?  Per-launch times of N launches
queued without host synchronisation, NT vs plain, and with idle gaps."""
import os, sys, time
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
rows, hp = 512, 512
thr = torch.from_numpy(ds.edge_thresholds.astype(np.int64)).to(torch.int32).to(dev)
src = torch.empty(ds.path_rows // 64 * hp, dtype=torch.int64, device=dev)
dst = torch.empty_like(src)
pitch = ctx.min_row_pitch
out = torch.empty(rows * pitch, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
ds.fill_paths_device(ctx.stream, src.data_ptr(), thr.data_ptr(), 0, hp)
ctx.transpose_bits_device(src.data_ptr(), hp, ds.path_rows, dst.data_ptr())
ctx.set_paths_device(dst.data_ptr(), ds.path_rows, hp)
batch = v2m.RowBatch(list(range(rows)))
ctx.synchronize()
ctx.profile_enable(True)

def burst(n, nt, gap=0.0):
	os.environ["V2M_NT_STORES"] = nt
	ctx.profile_reset()
	for _ in range(n):
		ctx.splice_rows_device(batch, out.data_ptr(), pitch)
		if gap:
			ctx.synchronize(); time.sleep(gap)
	return ctx.profile_launches(N.KERNEL_SPLICE_ALIGNED)

for label, nt, gap in (("nt back-to-back", "1", 0), ("plain back-to-back", "0", 0), ("nt back-to-back", "1", 0), ("nt 5ms gaps", "1", 0.005), ("nt back-to-back", "1", 0), ("plain back-to-back", "0", 0)):
	t = burst(40, nt, gap)
	print("%-20s first5 %s | last5 %s | median %.2f" % (label, " ".join("%.2f" % x for x in t[:5]), " ".join("%.2f" % x for x in t[-5:]), np.median(t)))
