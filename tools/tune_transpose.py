#!/usr/bin/env python3
"""A/B of the transpose panel shape in one process: the config-3 path matrix (5056 x 1M bits) and its inverse."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import vcf2multialign_amd as v2m
from vcf2multialign_amd import _native as N
ctx = v2m.Context(0)
hp, ep = (int(x) for x in (sys.argv[1:3] if len(sys.argv) > 2 else (5056, 1000000)))
print("matrix %d x %d bits" % (hp, ep))
n = hp // 64 * ep
src = torch.randint(-2**62, 2**62, (n,), dtype=torch.int64, device="cuda")
dst = torch.empty_like(src); back = torch.empty_like(src)
torch.cuda.synchronize(); ctx.profile_enable(True)
for shape in ("8x8", "stream16", "4x16", "8x8", "stream16"):
	os.environ["V2M_TRANSPOSE_PANEL"] = shape
	ts = []
	for rep in range(4):
		ctx.profile_reset()
		ctx.transpose_bits_device(src.data_ptr(), hp, ep, dst.data_ptr())
		t1 = ctx.profile_get(N.KERNEL_TRANSPOSE)[1]
		ctx.profile_reset()
		ctx.transpose_bits_device(dst.data_ptr(), ep, hp, back.data_ptr())
		t2 = ctx.profile_get(N.KERNEL_TRANSPOSE)[1]
		if rep: ts.append((t1, t2))
	ok = torch.equal(back, src)
	t1, t2 = min(a for a, _ in ts), min(b for _, b in ts)
	print("%-5s forward (copies x edges -> edges x copies) %.3f ms = %.0f GB/s   inverse %.3f ms = %.0f GB/s   involution %s" % (shape, t1, 2 * n * 8 / t1 / 1e6, t2, 2 * n * 8 / t2 / 1e6, ok))
