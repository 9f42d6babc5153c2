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
ap.add_argument("--candidates", type=int, default=0, help="before the sink path's first use: v2m_alloc_output of a 63-GB buffer chosen among this many candidates (what bench.py does first)")
ap.add_argument("--churn", default="", help="before the sink path's first use: hipMalloc this many 63-GB buffers; 'N' frees them again at once, 'N:hold' keeps them, 'N:touch' writes them (hipMemset) before freeing")
ap.add_argument("--recovery", type=int, default=0, help="after the setup: this many extra passes of 96 rows (9.6 GB) through the counting sink, each with the time since the setup ended -- how long does a slow link stay slow?")
ap.add_argument("--slots-first", action="store_true", help="with --candidates: run the sink path once BEFORE the candidates are allocated, so that its device and pinned slots exist already")
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


if args.churn:
	from vcf2multialign_amd import _native
	rt = _native.hip_runtime()
	rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
	rt.hipFree.argtypes = [C.c_void_p]
	rt.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
	n, _, how = args.churn.partition(":")
	ptrs = []
	for i in range(int(n)):
		p = C.c_void_p()
		assert rt.hipMalloc(C.byref(p), 627 * ctx.min_row_pitch) == 0
		ptrs.append(p)
	if how == "touch":
		for p in ptrs:
			assert rt.hipMemset(p, 1, 627 * ctx.min_row_pitch) == 0
		torch.cuda.synchronize()
	if how != "hold":
		for p in ptrs:
			assert rt.hipFree(p) == 0
	print("churn: %s x 63 GB hipMalloc'ed%s" % (n, {"hold": ", kept", "touch": ", written, freed"}.get(how, ", freed")), flush=True)
if args.candidates:
	if args.slots_first:
		st = State(-1, 0, 0)
		run(count_sink, C.byref(st))
	out = ctx.alloc_output(627 * ctx.min_row_pitch, args.candidates)
	print("output buffer of %.1f GB among %d candidates%s: %s" % (627 * ctx.min_row_pitch / 1e9, args.candidates, " (sink slots set up first)" if args.slots_first else "", ctx.info), flush=True)
if args.recovery:
	t_setup = time.perf_counter()
	small = v2m.RowBatch([v2m.PLOIDY_MAX] + list(range(95)))
	for i in range(args.recovery):
		st = State(-1, 0, 0)
		t0 = time.perf_counter()
		rc = ctx._lib.v2m_splice_rows(ctx._h, C.byref(small.struct), 0, count_sink, C.byref(st))
		dt = time.perf_counter() - t0
		print("  %5.1f s after the setup: 96 rows through the counting sink at %.1f GB/s" % (t0 - t_setup, st.bytes / dt / 1e9), flush=True)
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
