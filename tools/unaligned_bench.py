#!/usr/bin/env python3
"""Throughput of the unaligned path (count + scan + compact) next to the aligned one, same rows, config 3."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import vcf2multialign_amd as v2m
from vcf2multialign_amd import _native as N, synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "config3"
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 256
ds = synth.dataset(cfg); g = ds.graph
ctx = v2m.Context(0); ctx.upload_graph(g, ds.reference)
dev = torch.device("cuda", 0)
hp = 64 * ((rows + 63) // 64)
thr = torch.from_numpy(ds.edge_thresholds.astype(np.int64)).to(torch.int32).to(dev)
src = torch.empty(ds.path_rows // 64 * hp, dtype=torch.int64, device=dev); dst = torch.empty_like(src)
torch.cuda.synchronize()
ds.fill_paths_device(ctx.stream, src.data_ptr(), thr.data_ptr(), 0, hp)
ctx.transpose_bits_device(src.data_ptr(), hp, ds.path_rows, dst.data_ptr())
ctx.set_paths_device(dst.data_ptr(), ds.path_rows, hp)
upitch = (ctx.max_unaligned_length + 255) // 256 * 256
out = ctx.alloc_output(rows * upitch, 3)
batch = v2m.RowBatch(list(range(rows)))
ctx.synchronize(); ctx.profile_enable(True)
for name, un, mode in (("aligned", False, ""), ("unaligned", True, ""), ("aligned", False, ""), ("unaligned", True, ""), ("unaligned", True, "plain")):
	os.environ["V2M_UNALIGNED_STORE"] = mode
	ctx.profile_reset()
	lengths = ctx.splice_rows_device(batch, out, upitch, unaligned=un, want_lengths=True)
	k = N.KERNEL_SPLICE_UNALIGNED if un else N.KERNEL_SPLICE_ALIGNED
	n, ms = ctx.profile_get(k)
	_, ms_count = ctx.profile_get(N.KERNEL_UNALIGNED_COUNT)
	sums = ctx.checksum_rows_device(out, upitch, rows, lengths=lengths)
	print("%-10s %-10s %d rows, %.2f Gbases out: splice %.3f ms + count/scan %.3f ms -> %.0f Gbases/s   (checksum of checksums %016x)" % (name, mode, rows, lengths.sum() / 1e9, ms, ms_count if un else 0.0, lengths.sum() / (ms + (ms_count if un else 0)) / 1e6, int(np.bitwise_xor.reduce(sums))))
