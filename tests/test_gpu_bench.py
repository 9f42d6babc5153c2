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
	many = cpu["rows_dealt_to_threads"]             # context for the end-to-end figure; labelled as not the reference's behaviour
	assert many["value"] > 0 and many["cores"] >= 1 and "not the reference's behaviour" in many["note"]
	assert d["parity"]["bit_exact"] is True and d["parity"]["rows_checked"] > 0
	assert d["parity"]["batches_covered"] >= 1
	tr = d["roofline_transpose"]                    # the transpose at the reference's own 64-bit padding
	assert tr["matrix_bits"][0] % 64 == 0 and tr["matrix_bits"][1] % 64 == 0 and tr["matrix_bits"][0] < 1024   # mini3: 200 copies -> 256, not 1024
	assert tr["algorithmic_bytes"] == 2 * tr["matrix_bits"][0] * tr["matrix_bits"][1] // 8 and tr["achieved"] > 0
	assert tr["after_timing"]["involution_bit_exact"] is True and tr["after_timing"]["dense_forward_ms"] > 0
	tc = tr["cpu_baseline"]                         # the CPU path timed beside the GPU transpose, on the same matrix
	assert tc["kind"] == "port" and tc["cores"] == 1 and tc["seconds"] > 0 and tc["value"] > 0 and tc["unit"] == "GB/s"
	assert tc["bit_exact_vs_gpu_dense_form"] is True and tc["sample"]
	un = d["unaligned"]                             # the separately timed --unaligned leg: on the aligned leg's footprint, and on the first rows only
	for leg in (un, un["first_rows_only"]):
		assert leg["value"] > 0 and leg["parity"]["bit_exact"] is True and leg["roofline"]["kernel"] == "splice_unaligned_kernel"
		assert set(leg["kernels_ms"]) == {"resolve_effective_edges_kernel", "count_unaligned_kernel+scan_tile_counts_kernel", "splice_unaligned_kernel"}
		assert leg["time_per_base_vs_aligned_kernel"] > 0 and leg["footprint_GB"] > 0
		assert leg["aligned_kernel_same_rows_ms"] > 0 and leg["time_per_base_vs_aligned_kernel_same_rows"] > 0   # like for like: same rows, buffer and pitch
	assert un["rows"] >= un["first_rows_only"]["rows"] and "tuning" in un
	assert d["config"]["per_rank"] == [{"rank": 0, "rows": d["config"]["rows_total"], "batches": d["parity"]["batches_covered"], "ms_per_step": d["ms_per_step"],
		"launches": roof["launches"], "avg_launch_ms": roof["avg_launch_ms"]}]
	assert d["residency"] == "hbm" and "end_to_end" in d["residency_note"]      # the line says which number `value` is
	assert d["parity"]["all_rows"] is True and d["parity"]["rows_checked"] == d["parity"]["rows_total"] == d["config"]["rows_total"]   # every row, not a sample
	_check_placement(d, 1)
	_check_end_to_end(d, n_ranks=1)


def test_bench_unaligned_leg_on_the_dense_graph():
	"""The dense variant mix of BASELINE config 5 (mini5: one ALT edge per ~40 bp, MNPs, multi-allelic sites), where the unaligned stream-out's
	queue of short chunks is busiest: the leg is bit-exact against the oracle and carries its like-for-like yardstick (the aligned kernel on the
	same rows, buffer and pitch), so the figure the documents quote for config 5 is one the driver's own suite exercises."""
	r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "mini5", "--steps", "1", "--warmup", "0", "--output-candidates", "1", "--cpu-baseline-rows", "4",
		"--e2e-gb", "0", "--cpu-transpose", "0", "--transpose-extras", "0"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, cwd=ROOT)
	assert r.returncode == 0, r.stderr.decode()[-2000:]
	d = json.loads([l for l in r.stdout.decode().splitlines() if l.strip()][-1])
	assert d["parity"]["bit_exact"] is True and d["parity"]["all_rows"] is True
	un = d["unaligned"]
	assert un["parity"]["bit_exact"] is True and un["roofline"]["kernel"] == "splice_unaligned_kernel"
	assert un["kernels_ms"]["splice_unaligned_kernel"] > 0 and un["aligned_kernel_same_rows_ms"] > 0
	assert abs(un["time_per_base_vs_aligned_kernel_same_rows"] * un["aligned_kernel_same_rows_ms"] * un["bases"]
		- un["kernels_ms"]["splice_unaligned_kernel"] * d["config"]["aligned_length"] * un["rows"]) <= 0.02 * un["kernels_ms"]["splice_unaligned_kernel"] * d["config"]["aligned_length"] * un["rows"]


def _check_placement(d, n_ranks, forced_device=0):
	"""Every rank says where it ran (the HIP device the runtime reports, its PCI bus id, the slot's NUMA node, the CPUs it may use) and how many
	host threads it was given out of the job's quota: a mis-bound rank or an oversubscribed host is visible in the record."""
	host = d["config"]["host"]
	assert host["cpu_quota_cores"] >= 1 and host["cpu_quota_source"] and host["local_world_size"] == n_ranks
	place = d["config"]["placement"]
	assert [p["rank"] for p in place] == list(range(n_ranks))
	for p in place:
		assert p["hip_device"] == forced_device and p["pci_bus_id"] and p["name"] and p["cpus_allowed"]
		assert 1 <= p["host_threads"] <= max(1, host["cpu_quota_cores"] // n_ranks)
		# at N > 1 the rank's host side stays on its GPU's NUMA node where the box says which one that is (never at N = 1: the default run is as it was)
		assert "numa_bound_to_cpus" in p and (n_ranks > 1 or p["numa_bound_to_cpus"] is None)
		if n_ranks > 1 and p["numa_node"] is not None and p["numa_node"] >= 0:
			assert p["numa_bound_to_cpus"]
	assert sum(p["host_threads"] for p in place) <= max(n_ranks, host["cpu_quota_cores"])


def _check_end_to_end(d, n_ranks):
	"""The PCIe-inclusive leg: every row through v2m_splice_rows into a host sink that checksums it, all rows against the oracle."""
	e = d["end_to_end"]
	assert e["unit"] == "Gbases/s" and e["value"] > 0 and abs(e["value"] - e["bytes"] / e["seconds"] / 1e9) <= 0.01 * e["value"]
	assert e["rows"] >= n_ranks and e["bytes"] == e["rows"] * d["config"]["aligned_length"]
	roof = e["roofline"]
	assert roof["bound"] == "pcie" and roof["peak"] == 63.0 and roof["unit"] == "GB/s" and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
	assert len(e["per_rank_GBs"]) == n_ranks and min(e["per_rank_GBs"]) > 0 and abs(roof["achieved"] - min(e["per_rank_GBs"])) < 0.02
	assert e["parity"]["bit_exact"] is True and e["parity"]["rows_checked"] == e["rows"]


def test_bench_gpus_2_as_typed_starts_its_own_ranks():
	"""`python bench.py --gpus 2` with no launcher around it: the parent starts both ranks as child processes (here both on device 0
	over gloo, the one-GPU rehearsal of the N > 1 path), relays exactly one line on stdout and the ranks' chatter on stderr."""
	env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
	r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--force-device", "0", "--config", "mini3",
		"--steps", "2", "--warmup", "1", "--output-candidates", "1", "--batch-rows", "40", "--cpu-baseline-rows", "8"],
		stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, cwd=ROOT, env=env)
	assert r.returncode == 0, r.stderr.decode()[-3000:]
	assert b"init_process_group" not in r.stderr and b"c10d" not in r.stderr       # this launch form has no torch.distributed group
	lines = [l for l in r.stdout.decode().splitlines() if l.strip()]
	assert len(lines) == 1, "exactly one line on stdout: " + repr(lines)[:500]
	_check_two_rank_line(json.loads(lines[0]))
	d = json.loads(lines[0])
	assert "no torch.distributed" in d["config"]["ranks_coordinated_by"]


def _check_two_rank_line(d):
	"""What every N > 1 line has to carry (the first thing a SCALE record is judged on)."""
	assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["value"] > 0
	assert d["parity"]["bit_exact"] is True and d["residency"] == "hbm"
	per_rank = d["config"]["per_rank"]
	assert [p["rank"] for p in per_rank] == [0, 1] and all(p["ms_per_step"] > 0 for p in per_rank)
	assert sum(p["rows"] for p in per_rank) == d["config"]["rows_total"]
	assert d["parity"]["batches_covered"] == sum(p["batches"] for p in per_rank) and all(p["batches"] >= 2 for p in per_rank)
	assert abs(d["ms_per_step"] - max(p["ms_per_step"] for p in per_rank)) <= 1e-3   # the line's time is the MAX over ranks
	# every rank's kernel average is in the line, and the roofline is the SLOWEST rank's
	assert all(p["launches"] == 2 * p["batches"] and p["avg_launch_ms"] > 0 for p in per_rank)
	roof = d["roofline"]
	assert roof["avg_launch_ms"] == max(p["avg_launch_ms"] for p in per_rank)
	assert roof["avg_launch_ms"] == per_rank[roof["rank"]]["avg_launch_ms"] and roof["launches"] == per_rank[roof["rank"]]["launches"]
	assert abs(roof["achieved"] - roof["algorithmic_bytes_per_launch"] / (roof["avg_launch_ms"] * 1e-3) / 1e9) <= 0.01 * roof["achieved"]
	# the CPU path timed beside the GPU one in the same run, at N > 1 too (rank 0, after the timed region) -- splice and transpose
	cpu = d["cpu_baseline"]
	assert cpu["value"] > 0 and cpu["unit"] == "Gbases/s" and cpu["cores"] == 1 and cpu["kind"] == "port" and "rank 0" in cpu["sample"]
	tc = d["roofline_transpose"]["cpu_baseline"]
	assert tc["kind"] == "port" and tc["cores"] == 1 and tc["value"] > 0 and tc["bit_exact_vs_gpu_dense_form"] is True
	assert d["parity"]["all_rows"] is True and d["parity"]["rows_checked"] == d["config"]["rows_total"]
	_check_placement(d, 2)
	_check_end_to_end(d, n_ranks=2)


def test_bench_two_ranks_under_torch_distributed_run():
	"""The driver's launch form: python -m torch.distributed.run ... bench.py --gpus 2 (both ranks on device 0 here).  The ranks' figures
	travel over a gloo group; the line carries the same fields as the self-launched form."""
	import socket
	with socket.socket() as sock:
		sock.bind(("127.0.0.1", 0))
		port = sock.getsockname()[1]
	env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
	r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
		os.path.join(ROOT, "bench.py"), "--gpus", "2", "--force-device", "0", "--config", "mini3",
		"--steps", "2", "--warmup", "1", "--output-candidates", "1", "--batch-rows", "40", "--cpu-baseline-rows", "8"],
		stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, cwd=ROOT, env=env)
	assert r.returncode == 0, r.stderr.decode()[-3000:]
	lines = [l for l in r.stdout.decode().splitlines() if l.strip().startswith("{")]
	assert len(lines) == 1, "exactly one result line on stdout: " + repr(lines)[:500]
	d = json.loads(lines[0])
	_check_two_rank_line(d)
	assert "torch.distributed (gloo)" in d["config"]["ranks_coordinated_by"]


def test_bench_four_ranks_as_typed_uneven_shards():
	"""Four ranks on one GPU (mini3: 200 copies -> blocks of 8 dealt 48 / 48 / 48 / 56, REF on rank 0): every rank binds only its own slice,
	passes the same barriers (timed region, three end-to-end passes) and contributes its figures; the line adds up."""
	env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
	r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--force-device", "0", "--config", "mini3",
		"--steps", "2", "--warmup", "1", "--output-candidates", "1", "--batch-rows", "20", "--cpu-baseline-rows", "8", "--e2e-threads", "4"],
		stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, cwd=ROOT, env=env)
	assert r.returncode == 0, r.stderr.decode()[-3000:]
	lines = [l for l in r.stdout.decode().splitlines() if l.strip()]
	assert len(lines) == 1
	d = json.loads(lines[0])
	per_rank = d["config"]["per_rank"]
	assert d["n_gpus"] == 4 and [p["rank"] for p in per_rank] == [0, 1, 2, 3]
	assert [p["rows"] for p in per_rank] == [49, 48, 48, 56] and sum(p["rows"] for p in per_rank) == d["config"]["rows_total"] == 201
	assert d["parity"]["bit_exact"] is True and d["parity"]["batches_covered"] == sum(p["batches"] for p in per_rank)
	assert abs(d["ms_per_step"] - max(p["ms_per_step"] for p in per_rank)) <= 1e-3
	e = d["end_to_end"]
	assert e["rows"] == 201 and len(e["per_rank_GBs"]) == 4 and e["parity"]["bit_exact"] is True and e["parity"]["rows_checked"] == 201
	assert d["cpu_baseline"]["value"] > 0 and d["roofline_transpose"]["cpu_baseline"]["bit_exact_vs_gpu_dense_form"] is True


# A GPU box allows at most 6 processes on its card at once, and the pytest process is one of them: five ranks is the most a test
# started from here may put on the device.  The 8-rank plumbing of both launch forms runs without GPU work in tests/test_host_cpu.py.
FIVE = ["--gpus", "5", "--force-device", "0", "--config", "mini3", "--steps", "2", "--warmup", "1", "--output-candidates", "1", "--batch-rows", "8", "--cpu-baseline-rows", "8"]


def _check_five_rank_line(d, rows, rank0_has_copies=True):
	per_rank = d["config"]["per_rank"]
	assert d["n_gpus"] == 5 and [p["rank"] for p in per_rank] == list(range(5)) and [p["rows"] for p in per_rank] == rows
	assert sum(rows) == d["config"]["rows_total"] and d["value"] > 0
	assert d["parity"]["bit_exact"] is True and d["parity"]["all_rows"] is True and d["parity"]["rows_checked"] == sum(rows)
	assert d["parity"]["batches_covered"] == sum(p["batches"] for p in per_rank)
	assert all(p["batches"] == -(-p["rows"] // 8) and p["launches"] == 2 * p["batches"] for p in per_rank)     # --batch-rows 8: several launches per rank and step
	assert abs(d["ms_per_step"] - max(p["ms_per_step"] for p in per_rank)) <= 1e-3
	roof = d["roofline"]
	assert roof["avg_launch_ms"] == max(p["avg_launch_ms"] for p in per_rank) and per_rank[roof["rank"]]["rows"] > 0
	_check_placement(d, 5)
	e = d["end_to_end"]
	with_rows = sum(1 for r in rows if r)
	assert e["rows"] == sum(rows) and len(e["per_rank_GBs"]) == with_rows and e["parity"]["bit_exact"] is True and e["parity"]["rows_checked"] == sum(rows)
	assert d["cpu_baseline"]["value"] > 0
	if rank0_has_copies:    # (the transpose legs are rank 0's; a rank 0 that carries REF alone has no matrix to transpose)
		assert d["roofline_transpose"]["cpu_baseline"]["bit_exact_vs_gpu_dense_form"] is True


def test_bench_five_ranks_as_typed():
	"""`python bench.py --gpus 5`, all on device 0: mini3's 200 copies in blocks of 8 -> 40 per rank, REF on rank 0, launches of 8 rows."""
	env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
	r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + FIVE, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, cwd=ROOT, env=env)
	assert r.returncode == 0, r.stderr.decode()[-3000:]
	lines = [l for l in r.stdout.decode().splitlines() if l.strip()]
	assert len(lines) == 1
	_check_five_rank_line(json.loads(lines[0]), [41, 40, 40, 40, 40])
	assert r.stderr.decode().count("HIP device 0") == 5        # every rank logged where it landed


def test_bench_five_ranks_under_torch_distributed_run():
	import socket
	with socket.socket() as sock:
		sock.bind(("127.0.0.1", 0))
		port = sock.getsockname()[1]
	env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
	r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "5", "--master-addr", "127.0.0.1", "--master-port", str(port),
		os.path.join(ROOT, "bench.py")] + FIVE, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, cwd=ROOT, env=env)
	assert r.returncode == 0, r.stderr.decode()[-3000:]
	lines = [l for l in r.stdout.decode().splitlines() if l.strip().startswith("{")]
	assert len(lines) == 1
	d = json.loads(lines[0])
	_check_five_rank_line(d, [41, 40, 40, 40, 40])
	assert "torch.distributed (gloo)" in d["config"]["ranks_coordinated_by"]


def test_bench_ranks_that_own_nothing():
	"""12 samples = 24 copies = 3 blocks of 8 over 5 ranks: rank 0 carries REF alone, rank 1 owns no row at all, ranks 2-4 eight copies each.
	The empty rank passes every barrier, contributes zeros, and the line still adds up."""
	env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
	r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + FIVE + ["--samples", "12"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, cwd=ROOT, env=env)
	assert r.returncode == 0, r.stderr.decode()[-3000:]
	lines = [l for l in r.stdout.decode().splitlines() if l.strip()]
	assert len(lines) == 1
	_check_five_rank_line(json.loads(lines[0]), [1, 0, 8, 8, 8], rank0_has_copies=False)
