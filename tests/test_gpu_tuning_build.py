"""The tuning build (libv2m_hip_tuning.so = the product's sources with -DV2M_TUNING_BUILD): every transpose shape and flavour
that tools/tune_transpose.py can time must still be CORRECT, or its timings mean nothing.  The product library is what this
process has loaded, so the tuning build is exercised in a child process (V2M_HIP_LIBRARY)."""

import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys
sys.path[:0] = [%(root)r, os.path.join(%(root)r, "tests")]
import numpy as np
import oracle
import vcf2multialign_amd as v2m
from vcf2multialign_amd import _native
assert _native.library_path().endswith("libv2m_hip_tuning.so")
KERNELS = ["4x16", "16x8", "16x4", "4x8", "8x4", "8x16", "ring:16,8,8,4,16", "ring:16,8,8,4,64,slow", "ring:8,4,8,4,8", "ring:8,4,8,8,64",
	"ring:16,8,4,4,64", "ring:16,8,16,4,32", "ring:16,16,8,8,64/rr", "ring:8,8,8,4,64", "ring:8,8,8,8,128,nt", "ring:8,8,8,16,24/sf", "8x8", "stream16",
	"stream16:2,64", "stream16:4,32", "stream16:2,16", "stream16:4,32,8", "stream16:2,16,8", "stream8x32", "stream8x32:8", "stream32x8", "stream4x64", "stream16:old",
	"ring:8,8,8,8,64", "ring:8,8,8,8,128", "lines8", "lines8:2,4", "lines8:0,88", "lines8:3,816", "lines8:0,16", "lines8:2,168/sf", "lines8:400,1",
	"lines16", "lines16:2", "lines8:3,28", "lines8:400,281", "lines8:2,282/sf", "lines8:3,2816", "lines8:2,288/rr",
	"rot8", "rot8:1", "rot8:2/sf", "rot8:5/rr", "rot8:400", "rot8:0,88", "rot8:3,4", "rot8:0,16", "rot8:2,1616"]
SHAPES = [(1, 1), (1, 2), (3, 5), (9, 7), (16, 17), (79, 33), (5, 130), (17, 15), (2, 200)]
n = 0
with v2m.Context(0) as ctx:
	for kernel in KERNELS:
		os.environ["V2M_TRANSPOSE_PANEL"] = kernel
		for h, w in SHAPES:
			rng = np.random.default_rng(1000 * h + w)
			rows, cols = 64 * h, 64 * w
			src = rng.integers(0, 2 ** 63, size=rows * cols // 64, dtype=np.uint64) | (rng.integers(0, 2, size=rows * cols // 64, dtype=np.uint64) << np.uint64(63))
			got = ctx.transpose_matrix(src, rows, cols)
			assert np.array_equal(got, oracle.transpose_matrix(src, rows, cols, naive=True)), (kernel, h, w)
			assert np.array_equal(ctx.transpose_matrix(got, cols, rows), src), (kernel, h, w)
			n += 1
print("tuning build: %%d transposes ok" %% n)
"""


def test_every_tuning_kernel_transposes_correctly():
	from vcf2multialign_amd import build
	assert os.path.exists(build.TUNING_LIB_PATH), "build_native() builds it"
	env = dict(os.environ, V2M_HIP_LIBRARY=build.TUNING_LIB_PATH)
	env.pop("V2M_TRANSPOSE_PANEL", None)
	r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, env=env)
	assert r.returncode == 0, r.stderr.decode()[-3000:]
	assert b"transposes ok" in r.stdout
