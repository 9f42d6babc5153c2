#!/usr/bin/env python3
"""The splice kernel runs in a fast (~7.4 ms) or slow (~9 ms / 51 GB) mode that is fixed per process and alternates
between consecutive processes, while a memset of the same output buffer does not change.  Which buffer's placement
decides?  Re-allocate the graph-side buffers (second context) and the output buffer inside one process."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import vcf2multialign_amd as v2m
from vcf2multialign_amd import _native as N, synth

os.environ["V2M_NT_STORES"] = "1"
ds = synth.dataset("config3")
g = ds.graph
dev = torch.device("cuda", 0)
rows, hp = 512, 512
thr = torch.from_numpy(ds.edge_thresholds.astype(np.int64)).to(torch.int32).to(dev)
src = torch.empty(ds.path_rows // 64 * hp, dtype=torch.int64, device=dev)
dst = torch.empty_like(src)
batch = v2m.RowBatch(list(range(rows)))

def make_ctx():
	c = v2m.Context(0)
	c.upload_graph(g, ds.reference)
	torch.cuda.synchronize()
	ds.fill_paths_device(c.stream, src.data_ptr(), thr.data_ptr(), 0, hp)
	c.transpose_bits_device(src.data_ptr(), hp, ds.path_rows, dst.data_ptr())
	c.set_paths_device(dst.data_ptr(), ds.path_rows, hp)
	c.synchronize()
	c.profile_enable(True)
	return c

def t(c, ptr, pitch, reps=3):
	out = []
	for _ in range(reps + 1):
		c.profile_reset()
		c.splice_rows_device(batch, ptr, pitch)
		out.append(c.profile_get(N.KERNEL_SPLICE_ALIGNED)[1])
	return min(out[1:])

c1 = make_ctx()
pitch = c1.min_row_pitch
bufs = [torch.empty(rows * pitch, dtype=torch.uint8, device=dev) for _ in range(4)]
torch.cuda.synchronize()
os.environ["V2M_TILE_RUN"] = "1"
base = [t(c1, b.data_ptr(), pitch) for b in bufs]
print("4 buffers, tile_run 1, nt:", " ".join("%.3f" % x for x in base), " ptrs", " ".join("%x" % b.data_ptr() for b in bufs))
fast, slow = bufs[int(np.argmin(base))], bufs[int(np.argmax(base))]
os.environ["V2M_TILE_RUN"] = "1"
small = v2m.RowBatch(list(range(64)))
def t64(ptr):
	out = []
	for _ in range(3):
		c1.profile_reset()
		c1.splice_rows_device(small, ptr, pitch)
		out.append(c1.profile_get(N.KERNEL_SPLICE_ALIGNED)[1])
	return min(out[1:])
for name, b in (("fast", fast), ("slow", slow)):
	print("%s buffer, 64-row launches (6.4 GB) at row offsets 0,64,..: " % name + " ".join("%.3f" % t64(b.data_ptr() + k * 64 * pitch) for k in range(8)))
# finer: 8-row launches over the slow buffer in steps of 32 rows
tiny = v2m.RowBatch(list(range(8)))
def t8(ptr):
	out = []
	for _ in range(3):
		c1.profile_reset()
		c1.splice_rows_device(tiny, ptr, pitch)
		out.append(c1.profile_get(N.KERNEL_SPLICE_ALIGNED)[1])
	return min(out[1:])
for name, b in (("fast", fast), ("slow", slow)):
	print("%s buffer, 8-row launches (0.8 GB) every 32 rows: " % name + " ".join("%.3f" % t8(b.data_ptr() + k * 32 * pitch) for k in range(16)))
