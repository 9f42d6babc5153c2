#!/usr/bin/env python3
"""A/B of the splice kernel's launch knobs in ONE process, interleaved rounds (cdna_hip_programming.md rule 24).
Usage (GPU box): python tools/tune_splice.py [--config config3] [--rows 512] [--rounds 5]"""

import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def main():
	ap = argparse.ArgumentParser()
	ap.add_argument("--config", default="config3")
	ap.add_argument("--rows", type=int, default=512)
	ap.add_argument("--rounds", type=int, default=5)
	args = ap.parse_args()

	import torch
	import vcf2multialign_amd as v2m
	from vcf2multialign_amd import _native as N
	from vcf2multialign_amd import synth

	ds = synth.dataset(args.config)
	g = ds.graph
	ctx = v2m.Context(0)
	ctx.upload_graph(g, ds.reference)
	dev = torch.device("cuda", 0)
	rows = min(args.rows, ds.n_copies)
	hp = 64 * ((rows + 63) // 64)
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

	variants = [
		("nt rpg16", {"V2M_NT_STORES": "1"}),
		("plain rpg16", {"V2M_NT_STORES": "0"}),
		("nt rpg8", {"V2M_NT_STORES": "1", "V2M_ROWS_PER_GROUP": "8"}),
		("nt rpg12", {"V2M_NT_STORES": "1", "V2M_ROWS_PER_GROUP": "12"}),
		("nt rpg24", {"V2M_NT_STORES": "1", "V2M_ROWS_PER_GROUP": "24"}),
		("nt rpg32", {"V2M_NT_STORES": "1", "V2M_ROWS_PER_GROUP": "32"}),
	]
	times = {name: [] for name, _ in variants}
	nbytes = rows * g.aligned_length
	ctx.profile_enable(True)
	for r in range(args.rounds + 1):
		for name, env in variants:
			for k in ("V2M_NT_STORES", "V2M_ROWS_PER_GROUP"):
				os.environ.pop(k, None)
			os.environ.update(env)
			ctx.profile_reset()
			ctx.splice_rows_device(batch, out.data_ptr(), pitch)
			_, ms = ctx.profile_get(N.KERNEL_SPLICE_ALIGNED)
			if r:   # round 0 = warm-up
				times[name].append(ms)
	# memset ceiling for the same byte count
	torch.cuda.synchronize()
	ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
	fills = []
	for _ in range(args.rounds):
		ev0.record()
		out.fill_(45)
		ev1.record()
		torch.cuda.synchronize()
		fills.append(ev0.elapsed_time(ev1))
	print("rows %d x L %d = %.2f GB per launch" % (rows, g.aligned_length, nbytes / 1e9))
	for name, _ in variants:
		t = np.array(times[name])
		print("%-12s median %.3f ms  min %.3f ms  -> %.0f GB/s (median)" % (name, np.median(t), t.min(), nbytes / np.median(t) / 1e6))
	t = np.array(fills)
	print("%-12s median %.3f ms  min %.3f ms  -> %.0f GB/s (torch fill_ of the padded buffer, %.2f GB)" % ("fill_", np.median(t), t.min(), rows * pitch / np.median(t) / 1e6, rows * pitch / 1e9))


if __name__ == "__main__":
	main()
