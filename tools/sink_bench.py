#!/usr/bin/env python3
"""End-to-end (PCIe-inclusive) rate of the sink path: v2m_splice_rows hands every row body to a C sink that writes
it to a file descriptor, the way the host's A2M writer does.  Usage: python tools/sink_bench.py [--config config2]
[--rows N] [--dst /dev/null|count|PATH] [--slot-mb 512]"""
import argparse, ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="config2")
ap.add_argument("--rows", type=int, default=0)
ap.add_argument("--dst", default="/dev/null")
ap.add_argument("--slot-mb", type=int, default=0)
ap.add_argument("--reps", type=int, default=3)
args = ap.parse_args()
if args.slot_mb:
	os.environ["V2M_RING_SLOT_BYTES"] = str(args.slot_mb << 20)

import torch
import vcf2multialign_amd as v2m
from vcf2multialign_amd import _native as N, synth, build

ds = synth.dataset(args.config)
g = ds.graph
ctx = v2m.Context(0)
ctx.upload_graph(g, ds.reference)
n_copies = ds.n_copies if not args.rows else min(args.rows, ds.n_copies)
hp = 64 * ((n_copies + 63) // 64)
dev = torch.device("cuda", 0)
thr = torch.from_numpy(ds.edge_thresholds.astype(np.int64)).to(torch.int32).to(dev)
src = torch.empty(ds.path_rows // 64 * hp, dtype=torch.int64, device=dev)
dst = torch.empty_like(src)
torch.cuda.synchronize()
ds.fill_paths_device(ctx.stream, src.data_ptr(), thr.data_ptr(), 0, hp)
ctx.transpose_bits_device(src.data_ptr(), hp, ds.path_rows, dst.data_ptr())
ctx.set_paths_device(dst.data_ptr(), ds.path_rows, hp)
ctx.synchronize()

lib = C.CDLL(build.SYNTH_LIB_PATH)
class State(C.Structure):
	_fields_ = [("fd", C.c_int), ("rows", C.c_uint64), ("bytes", C.c_uint64)]
sink = C.cast(lib.v2ms_fd_sink, N.SINK_FN)
batch = v2m.RowBatch([v2m.PLOIDY_MAX] + list(range(n_copies)))
fd = -1 if args.dst == "count" else os.open(args.dst, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
L = g.aligned_length
for rep in range(args.reps):
	st = State(fd, 0, 0)
	t0 = time.perf_counter()
	rc = ctx._lib.v2m_splice_rows(ctx._h, C.byref(batch.struct), 0, sink, C.byref(st))
	dt = time.perf_counter() - t0
	assert rc == 0, ctx._lib.v2m_last_error(ctx._h)
	assert st.rows == batch.n_rows and st.bytes == batch.n_rows * L
	print("rep %d: %d rows x %d bases -> %s in %.3f s = %.2f Gbases/s end to end (%.1f GB/s over PCIe)" % (rep, st.rows, L, args.dst, dt, st.bytes / dt / 1e9, st.bytes / dt / 1e9))
