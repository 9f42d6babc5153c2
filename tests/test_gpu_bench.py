"""bench.py's output contract, on a small workload: one JSON line with the fields the driver and the judge read."""

import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_has_the_contract_fields():
	r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "mini3", "--steps", "2", "--warmup", "1", "--output-candidates", "2", "--cpu-baseline-rows", "8"],
		stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, cwd=ROOT)
	assert r.returncode == 0, r.stderr.decode()[-2000:]
	lines = [l for l in r.stdout.decode().splitlines() if l.strip()]
	assert len(lines) == 1, "exactly one line on stdout"
	d = json.loads(lines[0])
	assert d["metric"] == "aligned A2M Gbases/sec" and d["unit"] == "Gbases/s"
	assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
	assert d["value"] > 0 and d["ms_per_step"] > 0 and d["higher_is_better"] is True
	assert d["scaling"] in ("weak", "strong") and d["vs_baseline"] is None
	assert d["dtype"] == "u8" and d["data"] == "synthetic"
	assert "workload" in d["config"] and "model" not in d["config"]
	# value = rows * L * steps / time
	assert abs(d["value"] - d["config"]["rows_total"] * d["config"]["aligned_length"] / (d["ms_per_step"] * 1e-3) / 1e9) <= 0.01 * d["value"]
	roof = d["roofline"]
	assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
	assert roof["achieved"] > 0 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
	assert "traffic" in roof and roof["launches"] >= 2 and roof["avg_launch_ms"] > 0
	assert abs(roof["achieved"] - roof["algorithmic_bytes_per_launch"] / (roof["avg_launch_ms"] * 1e-3) / 1e9) <= 0.01 * roof["achieved"]
	cpu = d["cpu_baseline"]
	assert cpu["value"] > 0 and cpu["unit"] == "Gbases/s" and cpu["cores"] == 1 and cpu["kind"] == "port" and cpu["sample"]
	assert d["parity"]["bit_exact"] is True and d["parity"]["rows_checked"] > 0
	assert d["parity"]["batches_covered"] >= 1
	tr = d["roofline_transpose"]                    # the transpose at the reference's own 64-bit padding
	assert tr["matrix_bits"][0] % 64 == 0 and tr["matrix_bits"][1] % 64 == 0 and tr["matrix_bits"][0] < 1024   # mini3: 200 copies -> 256, not 1024
	assert tr["algorithmic_bytes"] == 2 * tr["matrix_bits"][0] * tr["matrix_bits"][1] // 8 and tr["achieved"] > 0
	assert tr["after_timing"]["involution_bit_exact"] is True and tr["after_timing"]["dense_forward_ms"] > 0
	un = d["unaligned"]                             # the separately timed --unaligned leg
	assert un["value"] > 0 and un["parity"]["bit_exact"] is True and un["roofline"]["kernel"] == "splice_unaligned_kernel"
	assert set(un["kernels_ms"]) == {"resolve_effective_edges_kernel", "count_unaligned_kernel+scan_tile_counts_kernel", "splice_unaligned_kernel"}
