#!/usr/bin/env python3
"""A/B of the transpose kernels in one process: a path matrix (default: config 3's 5056 x 1 000 000 bits, the reference's
own 64-bit padding) forward and inverse, every candidate checked word for word against the first one's result.

    python tools/tune_transpose.py [rows cols] [--set quick|full] [name ...]

Kernel names are those of V2M_TRANSPOSE_PANEL (vcf2multialign_amd/csrc/v2m_hip.hip: launch_transpose_named)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# the shapes beyond the product's three exist in the tuning build only (vcf2multialign_amd/build.py)
os.environ.setdefault("V2M_HIP_LIBRARY", os.path.join(ROOT, "vcf2multialign_amd", "libv2m_hip_tuning.so"))
import torch  # noqa: E402

import vcf2multialign_amd as v2m  # noqa: E402
from vcf2multialign_amd import _native as N  # noqa: E402

QUICK = ["8x8", "stream16", "stream16:old", "lines8", "lines8:4", "lines8:8", "lines8:16", "lines8:32", "lines8:0,4", "lines8:0,16", "ring:8,8,8,4,64", "ring:8,8,8,8,64", "ring:8,8,8,8,64,nt", "ring:8,8,8,8,128", "ring:8,8,8,8,128,nt", "ring:8,8,8,16,128", "ring:8,4,8,8,64", "ring:8,4,8,8,128,nt", "ring:16,16,8,8,64", "ring:16,16,8,8,128,nt"]
FULL = QUICK + ["8x8/rr", "stream16/rr", "4x16", "16x4", "ring:16,8,8,4,64", "ring:16,8,8,4,64,slow", "ring:16,8,4,4,64", "ring:16,8,16,4,64", "ring:8,4,8,4,64"]

args = sys.argv[1:]
names_set = QUICK
if "--set" in args:
	i = args.index("--set")
	names_set = FULL if args[i + 1] == "full" else QUICK
	del args[i:i + 2]
dims = [a for a in args if a.isdigit()]
names = [a for a in args if not a.isdigit()] or names_set
hp, ep = (int(dims[0]), int(dims[1])) if len(dims) >= 2 else (5056, 1000000)

ctx = v2m.Context(0)
n = hp // 64 * ep
print("matrix %d x %d bits, %.3f GB moved per transpose" % (hp, ep, 2 * n * 8 / 1e9), flush=True)
src = torch.randint(-2**62, 2**62, (n,), dtype=torch.int64, device="cuda")
dst = torch.empty_like(src)
back = torch.empty_like(src)
want_dst = None
torch.cuda.synchronize()
ctx.profile_enable(True)
for name in names:
	os.environ["V2M_TRANSPOSE_PANEL"] = name
	ts = []
	try:
		dst.zero_(); back.zero_()
		torch.cuda.synchronize()
		for rep in range(4):
			ctx.profile_reset()
			ctx.transpose_bits_device(src.data_ptr(), hp, ep, dst.data_ptr())
			t1 = ctx.profile_get(N.KERNEL_TRANSPOSE)[1]
			ctx.profile_reset()
			ctx.transpose_bits_device(dst.data_ptr(), ep, hp, back.data_ptr())
			t2 = ctx.profile_get(N.KERNEL_TRANSPOSE)[1]
			if rep:
				ts.append((t1, t2))
	except v2m.V2MError as e:
		print("%-26s %s" % (name, e), flush=True)
		continue
	if want_dst is None:
		want_dst = dst.clone()
	ok = torch.equal(back, src) and torch.equal(dst, want_dst)
	if not ok:
		for label, got, want in (("forward", dst, want_dst), ("inverse", back, src)):
			bad = (got != want).nonzero().flatten()
			if bad.numel():
				i = int(bad[0].item())
				print("    %s: %d of %d words differ; first at word %d (column %d, word %d of it): got %016x want %016x" % (label, bad.numel(), n, i,
					i // ((ep if label == "forward" else hp) // 64), i % ((ep if label == "forward" else hp) // 64), got[i].item() & (2**64 - 1), want[i].item() & (2**64 - 1)), flush=True)
	t1, t2 = min(a for a, _ in ts), min(b for _, b in ts)
	print("%-26s forward %.3f ms = %5.0f GB/s   inverse %.3f ms = %5.0f GB/s   %s" % (name, t1, 2 * n * 8 / t1 / 1e6, t2, 2 * n * 8 / t2 / 1e6, "ok" if ok else "WRONG RESULT"), flush=True)
