#!/usr/bin/env python3
"""What bounds bench.py's end-to-end leg: the same 640 config-3 rows through v2m_splice_rows into (a) a sink that only counts
(the link's rate) and (b) the checksumming sink on 4 .. 64 host threads.  Usage: python tools/e2e_sink_sweep.py [--rows 640]"""
import argparse, ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="config3")
ap.add_argument("--rows", type=int, default=640)
ap.add_argument("--threads", default="4,8,16,32,64")
args = ap.parse_args()

import torch
import vcf2multialign_amd as v2m
from vcf2multialign_amd import _native as N, synth, build

ds = synth.dataset(args.config)
g = ds.graph
ctx = v2m.Context(0)
ctx.upload_graph(g, ds.reference)
n_copies = min(args.rows, ds.n_copies)
hp = 64 * ((n_copies + 63) // 64)
dev = torch.device("cuda", 0)
thr = torch.from_numpy(ds.edge_thresholds.astype(np.int64)).to(torch.int32).to(dev)
src = torch.empty(ds.path_rows // 64 * hp, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
ds.fill_paths_device(ctx.stream, src.data_ptr(), thr.data_ptr(), 0, hp)
ctx.bind_path_matrix_device(src.data_ptr(), hp, ds.path_rows)
ctx.synchronize()

sl = C.CDLL(build.SYNTH_LIB_PATH)
sl.v2ms_checksum_sink_create.restype = C.c_void_p
sl.v2ms_checksum_sink_create.argtypes = [C.c_uint64, C.c_uint32]
sl.v2ms_checksum_sink_destroy.argtypes = [C.c_void_p]
sl.v2ms_checksum_sink_bytes.restype = C.c_uint64
sl.v2ms_checksum_sink_bytes.argtypes = [C.c_void_p]


class State(C.Structure):
	_fields_ = [("fd", C.c_int), ("rows", C.c_uint64), ("bytes", C.c_uint64)]


batch = v2m.RowBatch([v2m.PLOIDY_MAX] + list(range(n_copies - 1)))
L = g.aligned_length
count_sink = C.cast(sl.v2ms_fd_sink, N.SINK_FN)
sum_sink = C.cast(sl.v2ms_checksum_sink_fn, N.SINK_FN)


def run(sink, state):
	t0 = time.perf_counter()
	rc = ctx._lib.v2m_splice_rows(ctx._h, C.byref(batch.struct), 0, sink, state)
	dt = time.perf_counter() - t0
	assert rc == 0, ctx._lib.v2m_last_error(ctx._h)
	return dt


st = State(-1, 0, 0)
run(count_sink, C.byref(st))
for rep in range(2):
	st = State(-1, 0, 0)
	dt = run(count_sink, C.byref(st))
	print("counting sink: %d rows, %.3f s = %.1f GB/s" % (st.rows, dt, st.bytes / dt / 1e9), flush=True)
for t in [int(x) for x in args.threads.split(",")]:
	for rep in range(2):
		s = sl.v2ms_checksum_sink_create(batch.n_rows, t)
		dt = run(sum_sink, s)
		print("checksum sink, %2d threads: %.3f s = %.1f GB/s" % (t, dt, sl.v2ms_checksum_sink_bytes(s) / dt / 1e9), flush=True)
		sl.v2ms_checksum_sink_destroy(s)
