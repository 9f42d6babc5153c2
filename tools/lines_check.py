#!/usr/bin/env python3
"""Correctness of named transpose kernels on ragged shapes with guard words (tuning build), before they are timed.

    python tools/lines_check.py [--random N] name [name ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
os.environ.setdefault("V2M_HIP_LIBRARY", os.path.join(ROOT, "vcf2multialign_amd", "libv2m_hip_tuning.so"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import oracle  # noqa: E402
import vcf2multialign_amd as v2m  # noqa: E402

SHAPES = [(1, 1), (1, 2), (2, 1), (3, 5), (8, 8), (9, 7), (16, 17), (79, 33), (5, 130), (64, 64), (17, 15), (33, 31), (2, 200), (1, 79), (79, 1), (7, 300), (300, 7), (13, 257)]
names = sys.argv[1:]
if "--random" in names:          # --random N: N more shapes of up to 400 x 400 words (a soak run)
	i = names.index("--random")
	rng0 = np.random.default_rng(int(names[i + 1]))
	SHAPES = SHAPES + [(int(rng0.integers(1, 400)), int(rng0.integers(1, 400))) for _ in range(int(names[i + 1]))]
	del names[i:i + 2]
bad = 0
with v2m.Context(0) as ctx:
	for name in names:
		os.environ["V2M_TRANSPOSE_PANEL"] = name
		for h, w in SHAPES:
			rng = np.random.default_rng(77 * h + w)
			rows, cols = 64 * h, 64 * w
			n = rows * cols // 64
			src = rng.integers(0, 2 ** 63, size=n, dtype=np.uint64) | (rng.integers(0, 2, size=n, dtype=np.uint64) << np.uint64(63))
			guard = 4096
			d_src = torch.from_numpy(src.view(np.int64)).cuda()
			d_dst = torch.full((n + 2 * guard,), 0x5A5A5A5A5A5A5A5A, dtype=torch.int64, device="cuda")
			torch.cuda.synchronize()
			ctx.transpose_bits_device(d_src.data_ptr(), rows, cols, d_dst.data_ptr() + 8 * guard)
			ctx.synchronize()
			host = d_dst.cpu().numpy().view(np.uint64)
			want = oracle.transpose_matrix(src, rows, cols, naive=True)
			ok_guard = bool((host[:guard] == 0x5A5A5A5A5A5A5A5A).all() and (host[guard + n:] == 0x5A5A5A5A5A5A5A5A).all())
			ok = np.array_equal(host[guard:guard + n], want)
			if not (ok and ok_guard):
				bad += 1
				diff = np.nonzero(host[guard:guard + n] != want)[0]
				print("%s %dx%d: guard %s, %d of %d words differ%s" % (name, h, w, "ok" if ok_guard else "OVERWRITTEN", diff.size, n,
					"" if not diff.size else "; first at word %d (column %d word %d): got %016x want %016x" % (diff[0], diff[0] // w, diff[0] % w, host[guard + diff[0]], want[diff[0]])), flush=True)
		print("%s: checked %d shapes" % (name, len(SHAPES)), flush=True)
print("FAILED: %d" % bad if bad else "all ok")
sys.exit(1 if bad else 0)
