#!/usr/bin/env python3
"""A/B of the unaligned kernel's two pack forms in ONE process, on one buffer (V2M_UNALIGNED_PACK is read per call): per-slot (round 3) against
queued-per-wave (round 4), alternating.  The second form is NOT in the product: apply profiles/r04/unaligned_batched_pack_experiment.diff first
(without it both modes run the same kernel).  Usage: python tools/unaligned_pack_ab.py [config3] [rows=256] [reps=6]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import vcf2multialign_amd as v2m
from vcf2multialign_amd import _native as N, synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "config3"
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 256
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
ds = synth.dataset(cfg); g = ds.graph
ctx = v2m.Context(0); ctx.upload_graph(g, ds.reference)
dev = torch.device("cuda", 0)
hp = 64 * ((rows + 63) // 64)
thr = torch.from_numpy(ds.edge_thresholds.astype(np.int64)).to(torch.int32).to(dev)
src = torch.empty(ds.path_rows // 64 * hp, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
ds.fill_paths_device(ctx.stream, src.data_ptr(), thr.data_ptr(), 0, hp)
ctx.bind_path_matrix_device(src.data_ptr(), hp, ds.path_rows)
upitch = (ctx.max_unaligned_length + 255) // 256 * 256
out = ctx.alloc_output(rows * upitch, 1)
batch = v2m.RowBatch([v2m.PLOIDY_MAX] + list(range(rows - 1)))
os.environ["V2M_UNALIGNED_STORE"] = "plain"
os.environ["V2M_NT_STORES"] = "1"
ctx.synchronize(); ctx.profile_enable(True)
res = {"aligned": [], "perslot": [], "batched": []}
sums = {}
for rep in range(reps):
	for mode in ("aligned", "perslot", "batched"):
		os.environ["V2M_UNALIGNED_PACK"] = mode
		ctx.profile_reset()
		un = mode != "aligned"
		lengths = ctx.splice_rows_device(batch, out, upitch, unaligned=un, want_lengths=True)
		ctx.synchronize()
		n, ms = ctx.profile_get(N.KERNEL_SPLICE_UNALIGNED if un else N.KERNEL_SPLICE_ALIGNED)
		res[mode].append(ms / n)
		if un:
			s = int(np.bitwise_xor.reduce(ctx.checksum_rows_device(out, upitch, rows, lengths=lengths)))
			sums.setdefault(mode, s)
			assert sums[mode] == s
assert sums["perslot"] == sums["batched"], "the two forms disagree"
for mode, t in res.items():
	print("%s %-8s %d rows: %s ms (min %.3f, median %.3f)" % (cfg, mode, rows, " ".join("%.3f" % x for x in t), min(t), sorted(t)[len(t) // 2]))
print("batched / perslot (median): %.3f; per-slot / aligned %.3f, batched / aligned %.3f" % (sorted(res["batched"])[reps // 2] / sorted(res["perslot"])[reps // 2],
	sorted(res["perslot"])[reps // 2] / sorted(res["aligned"])[reps // 2], sorted(res["batched"])[reps // 2] / sorted(res["aligned"])[reps // 2]))
