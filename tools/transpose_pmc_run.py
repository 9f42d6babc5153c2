#!/usr/bin/env python3
"""The transposes under counters, in a fixed dispatch order (tools/transpose_pmc.sh parses by that order): for each shape -- config 3's dense
5056 x 1 000 000 bits, then both dimensions padded to 1024 bits (every column on a 128-byte line: what the product path's destination looks like) -- and
each kernel of KERNELS: forward, inverse, three times.  Prints the order and the HIP-event times."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import vcf2multialign_amd as v2m  # noqa: E402
from vcf2multialign_amd import _native as N  # noqa: E402

SHAPES = [(5056, 1000000), (5120, 1000448)]
KERNELS = ["stream16", "lines8"]
REPS = 3

ctx = v2m.Context(0)
ctx.profile_enable(True)
for hp, ep in SHAPES:
	n = hp // 64 * ep
	src = torch.randint(-2**62, 2**62, (n,), dtype=torch.int64, device="cuda")
	dst = torch.empty_like(src)
	back = torch.empty_like(src)
	torch.cuda.synchronize()
	for name in KERNELS:
		os.environ["V2M_TRANSPOSE_PANEL"] = name
		fw, inv = [], []
		for rep in range(REPS):
			ctx.profile_reset()
			ctx.transpose_bits_device(src.data_ptr(), hp, ep, dst.data_ptr())
			fw.append(ctx.profile_get(N.KERNEL_TRANSPOSE)[1])
			ctx.profile_reset()
			ctx.transpose_bits_device(dst.data_ptr(), ep, hp, back.data_ptr())
			inv.append(ctx.profile_get(N.KERNEL_TRANSPOSE)[1])
		ok = torch.equal(back, src)
		print("ORDER %dx%d %s: %d x (forward, inverse); forward %.3f ms, inverse %.3f ms (best of the last %d), %.3f GB moved each way, involution %s"
			% (hp, ep, name, REPS, min(fw[1:]), min(inv[1:]), REPS - 1, 2 * n * 8 / 1e9, "ok" if ok else "WRONG"), flush=True)
	del src, dst, back
