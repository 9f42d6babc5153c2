#!/usr/bin/env python3
"""bench.py -- aligned A2M Gbases/s of the MI355X haplotype-splice path.

A "step" is one pass of the hot path over this rank's share of the synthetic workload:
transpose the rank's genotype bit matrix (transpose_matrix), then splice every owned row
(REF on rank 0 + the rank's chromosome copies) into aligned A2M row bodies in HBM, in batches that
reuse one device output buffer (the rows of config 3 total 501 GB).  All inputs are resident in HBM
before the timed region starts; nothing is copied to the host inside it.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Haplotypes shard across ranks in blocks of 8 chromosome copies (whole bytes of the bit-packed path matrix) with the
reference and graph replicated; there is no collective on the data path.  The total work is the
named config's and is fixed as N grows, hence "scaling": "strong".

What the ranks need from each other is a start barrier, the maximum of their times and the AND of their parity flags -- no
data.  Typed as `python bench.py --gpus N` the parent process is that hub itself: it starts the N ranks as children and
serves barriers and one gather over their pipes (no torch.distributed, no RCCL, nothing that could fail to initialise).
Under `torch.distributed.run` the same three operations go through a torch.distributed group (--dist-backend, default gloo:
CPU tensors only; `nccl` = RCCL is accepted, the data path never touches it either way).

Rank 0 prints ONE JSON line (see README / DESIGN.md for the fields).  `value` is the HBM-resident rate (`residency`: "hbm":
inputs in HBM, rows written to HBM, nothing crosses PCIe in the timed region); the rate a caller of output_a2m sees, with
every row crossing the link into a host sink, is the separate `end_to_end` leg.  Besides the contract's fields the line
carries `roofline` (the dominant kernel, splice_aligned_kernel; at N > 1 the SLOWEST rank's launches), `roofline_transpose`
(the transpose of a source at the reference's own 64-bit padding into the library's path matrix, inside the timed region; the
ABI's dense form, forward and inverse, and a 1024-bit-padded matrix measured after it), `unaligned` (a second, separately
timed leg: the same rows without '-' padding), `end_to_end`, `parity` and -- at every N, timed on rank 0 after the timed
region -- `cpu_baseline` (and `roofline_transpose.cpu_baseline`).
"""

import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
	if p not in sys.path:
		sys.path.insert(0, p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 achievable)
PCIE_PEAK_GBS = 63.0    # MI355X host link: PCIe Gen5 x16, 63 GB/s spec (MI355X_MICROARCH.md "Host link")


def _kernel_sources():
	"""Every file libv2m_hip.so is compiled from: v2m_hip.hip and what it reaches through #include "..." (the build's own
	dependency list, read off the sources -- vcf2multialign_amd/build.py:include_closure)."""
	import importlib.util
	spec = importlib.util.spec_from_file_location("_v2m_build", os.path.join(ROOT, "vcf2multialign_amd", "build.py"))
	b = importlib.util.module_from_spec(spec)
	spec.loader.exec_module(b)
	return tuple(sorted(os.path.relpath(p, ROOT) for p in b.HIP_DEPS))


KERNEL_SOURCES = _kernel_sources()


def log(*a):
	print(*a, file=sys.stderr, flush=True)


def git_blob_hash(path):
	"""What `git hash-object` prints for the file: the stamp that ties a PMC traffic figure to the kernel sources it was measured on."""
	with open(path, "rb") as f:
		data = f.read()
	return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def kernel_source_stamp():
	return {p: git_blob_hash(os.path.join(ROOT, p)) for p in KERNEL_SOURCES}


def _hip_runtime():
	# the HIP runtime already loaded by torch / the library (one runtime per process), whatever its soname
	from vcf2multialign_amd import _native
	return _native.hip_runtime()


def _hip_memset(torch, ptr, nbytes):
	import ctypes
	rt = _hip_runtime()
	rt.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
	rc = rt.hipMemsetAsync(ptr, 45, nbytes, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
	assert rc == 0, "hipMemsetAsync failed"


def _device_bytes(ptr, nbytes):
	import ctypes
	rt = _hip_runtime()
	rt.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
	buf = ctypes.create_string_buffer(nbytes)
	rc = rt.hipMemcpy(buf, ptr, nbytes, 2)   # hipMemcpyDeviceToHost
	assert rc == 0, "hipMemcpy failed"
	return buf.raw


def device_binding(dev_index):
	"""Where this rank landed: the HIP device the runtime reports as current (after torch.cuda.set_device), its name and PCI bus id,
	the NUMA node that slot hangs off (sysfs) and the CPUs this process may run on -- logged by every rank and carried in
	config.per_rank, so that a rank bound to the wrong GPU, or to a GPU across the socket, is visible in the record."""
	import ctypes
	out = {"hip_device": None, "name": None, "pci_bus_id": None, "numa_node": None, "cpus_allowed": None}
	try:
		rt = _hip_runtime()
		cur = ctypes.c_int(-1)
		if rt.hipGetDevice(ctypes.byref(cur)) == 0:
			out["hip_device"] = cur.value
		buf = ctypes.create_string_buffer(64)
		if rt.hipDeviceGetPCIBusId(buf, 64, ctypes.c_int(dev_index)) == 0:
			out["pci_bus_id"] = buf.value.decode().lower()
		import torch
		out["name"] = torch.cuda.get_device_name(dev_index)
	except Exception as e:   # a record, not a requirement
		out["error"] = repr(e)
	if out["pci_bus_id"]:
		try:
			with open("/sys/bus/pci/devices/%s/numa_node" % out["pci_bus_id"]) as f:
				out["numa_node"] = int(f.read().strip())
		except (OSError, ValueError):
			pass
	try:
		with open("/proc/self/status") as f:
			for line in f:
				if line.startswith("Cpus_allowed_list:"):
					out["cpus_allowed"] = line.split(":", 1)[1].strip()
	except OSError:
		pass
	return out


def bind_to_numa_node(node):
	"""At N > 1 every rank keeps its host side -- the pinned slots its D2H copies land in (first touch), the sink's and the checker's
	threads -- on the socket its GPU hangs off: eight links deliver ~450 GB/s into host memory and the sinks read it all back, which is
	the two sockets' DRAM bandwidth, not something to send across the socket interconnect as well.  The CPUs of the
	GPU's NUMA node that this process is allowed to run on; None (and nothing changed) when the node is unknown or shares no CPU with
	the allowed set.  Never fatal."""
	try:
		if node is None or node < 0:
			return None
		with open("/sys/devices/system/node/node%d/cpulist" % node) as f:
			text = f.read().strip()
		cpus = set()
		for part in text.split(","):
			if part:
				lo, _, hi = part.partition("-")
				cpus.update(range(int(lo), int(hi or lo) + 1))
		cpus &= os.sched_getaffinity(0)
		if not cpus:
			return None
		os.sched_setaffinity(0, cpus)
		return text
	except (OSError, ValueError, AttributeError):
		return None


def build_checker_once():
	"""The CPU oracle's library (test infrastructure: loaded and called only AFTER the timed region, by every rank) is brought up to date
	here, once -- by the parent of `python bench.py --gpus N` or by local rank 0 before the ranks' first barrier -- so that N ranks never
	find it stale and run make side by side.  Building the checker is not using it: nothing of it is imported or loaded here."""
	import subprocess
	src, lib = os.path.join(ROOT, "oracle", "v2m_oracle.cc"), os.path.join(ROOT, "oracle", "libv2m_oracle.so")
	if os.path.exists(lib) and os.path.getmtime(lib) >= os.path.getmtime(src):
		return
	if "rocprof" in os.environ.get("LD_PRELOAD", "").lower() or any(k.startswith(("ROCPROF", "ROCPROFILER_")) for k in os.environ):
		sys.exit("[bench] oracle/libv2m_oracle.so is missing or stale and this process runs under a profiler: run `make -C oracle` first, without the profiler")
	subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"], stdout=sys.stderr)


HUB_ENV = "V2M_BENCH_HUB"   # set for the children of `python bench.py --gpus N`: the parent serves barrier / gather over their pipes


def launch_ranks(n, argv):
	"""`python bench.py --gpus N` typed as is (no WORLD_SIZE in the environment): this parent -- which has parsed its arguments
	and nothing else, no torch import, no HIP call -- builds the libraries (in a child), starts the N ranks as fresh child
	processes (RANK / LOCAL_RANK / WORLD_SIZE as under torch.distributed.run, plus V2M_BENCH_HUB) and is their hub: each rank's
	real stdout is a pipe to the parent, its stdin a pipe from it.  A rank writes `#B` to enter a barrier and is released by a
	`go` line once all N have; `#G <json>` contributes to the one gather, whose N objects go to rank 0 as one line; the line
	rank 0 prints that starts with `{` is the result and is relayed on stdout.  Everything else any rank prints goes to stderr.
	There is no torch.distributed group and no RCCL on this path.  Never os.exec*: the children are children."""
	import queue
	import subprocess
	import threading
	if "--hub-selftest" not in argv:
		rc = subprocess.call([sys.executable, "-c", "import sys; sys.path.insert(0, %r); from vcf2multialign_amd import build; build.build_native()" % ROOT], stdout=sys.stderr)
		if rc != 0:
			sys.exit("[bench] building the native libraries failed (exit %d)" % rc)
		build_checker_once()
	procs = []
	for r in range(n):
		env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n))
		env[HUB_ENV] = "1"
		env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
		procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=sys.stderr))
	events = queue.Queue()

	def pump(r):
		for raw in procs[r].stdout:
			events.put((r, raw.decode(errors="replace").rstrip("\n")))
		events.put((r, None))

	for r in range(n):
		threading.Thread(target=pump, args=(r,), daemon=True).start()

	def send(r, text):
		try:
			procs[r].stdin.write((text + "\n").encode())
			procs[r].stdin.flush()
		except (BrokenPipeError, OSError):
			pass   # the rank is gone: its exit code is reported below

	line, at_barrier, gathered, open_pipes, failed = None, set(), {}, n, False
	while open_pipes:
		try:
			r, msg = events.get(timeout=0.5)
		except queue.Empty:
			# a rank that dies leaves the others waiting at a barrier: end them (by their own PIDs) instead of hanging until a timeout
			if any(p.poll() not in (None, 0) for p in procs):
				failed = True
				time.sleep(2.0)
				for p in procs:
					if p.poll() is None:
						p.terminate()
			continue
		if msg is None:
			open_pipes -= 1
		elif msg == "#B":
			at_barrier.add(r)
			if len(at_barrier) == n:
				at_barrier.clear()
				for k in range(n):
					send(k, "go")
		elif msg.startswith("#G "):
			gathered[r] = msg[3:]
			if len(gathered) == n:
				send(0, "[" + ",".join(gathered[k] for k in range(n)) + "]")
				gathered = {}
		elif r == 0 and msg.startswith("{") and line is None:
			line = msg
		elif msg.strip():
			log(msg)
	codes = [p.wait() for p in procs]
	bad = [(r, c) for r, c in enumerate(codes) if c != 0]
	if bad:
		log("[bench] rank(s) failed: " + ", ".join("rank %d -> exit %d" % rc for rc in bad))
	if line is not None and not (failed and bad):
		print(line, flush=True)
	if bad:
		sys.exit(bad[0][1] if 0 < bad[0][1] < 256 else 1)
	if line is None:
		sys.exit("[bench] rank 0 printed no result line")


class SoloHub:
	"""N = 1: nothing to wait for."""
	kind = "single process"

	def barrier(self):
		pass

	def gather(self, obj):
		return [obj]

	def close(self):
		pass


class PipeHub(SoloHub):
	"""A rank started by launch_ranks(): barrier and gather through the parent, over this process's own stdin / stdout pipes."""
	kind = "parent process of `python bench.py --gpus N` over pipes (no torch.distributed, no RCCL)"

	def __init__(self, rank, out):
		self.rank, self.out = rank, out

	def _say(self, text):
		self.out.write(text + "\n")
		self.out.flush()

	def _hear(self):
		line = sys.stdin.readline()
		if not line:
			sys.exit("[bench] rank %d: the parent went away" % self.rank)
		return line.rstrip("\n")

	def barrier(self):
		self._say("#B")
		if self._hear() != "go":
			sys.exit("[bench] rank %d: unexpected reply at a barrier" % self.rank)

	def gather(self, obj):
		self._say("#G " + json.dumps(obj))
		return json.loads(self._hear()) if self.rank == 0 else None


class TorchHub(SoloHub):
	"""Under torch.distributed.run: the same two operations through a torch.distributed group.  gloo (the default) moves CPU
	objects only; with nccl (= RCCL) the objects travel through tensors on this rank's device."""

	def __init__(self, backend, rank, dev):
		import torch.distributed as dist
		os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
		# one node by contract: the group's own sockets go over loopback, whatever the host's name resolves to (or does not)
		os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
		os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
		if backend == "nccl":
			dist.init_process_group("nccl", device_id=dev)
		else:
			dist.init_process_group(backend)
		self.dist, self.rank = dist, rank
		self.kind = "torch.distributed (%s) under torch.distributed.run: barrier + all_gather_object of the ranks' figures, no data" % backend

	def barrier(self):
		self.dist.barrier()

	def gather(self, obj):
		everyone = [None] * self.dist.get_world_size()
		self.dist.all_gather_object(everyone, obj)
		return everyone if self.rank == 0 else None

	def close(self):
		self.dist.barrier()
		self.dist.destroy_process_group()


def main():
	ap = argparse.ArgumentParser()
	ap.add_argument("--gpus", type=int, default=1)
	ap.add_argument("--steps", type=int, default=3)
	ap.add_argument("--warmup", type=int, default=1)
	ap.add_argument("--config", default="config3", help="synthetic workload (vcf2multialign_amd/synth.py CONFIGS)")
	ap.add_argument("--samples", type=int, default=0, help="override the config's number of diploid samples (tests: so few copies that some ranks own no row at all)")
	ap.add_argument("--batch-rows", type=int, default=0, help="rows per splice launch (one device output buffer of this many rows is reused); 0 = as many as fit --batch-gb")
	ap.add_argument("--batch-gb", type=float, default=64.0, help="size of the reused device output buffer when --batch-rows is 0: launches that write a ~64-GB address range reach the full HBM write rate (DESIGN.md section 4)")
	ap.add_argument("--dist-backend", default="gloo", help="under torch.distributed.run only: the torch.distributed backend of the barrier / gather of figures (gloo: CPU objects; nccl = RCCL).  The data path has no collective; `python bench.py --gpus N` as typed uses neither")
	ap.add_argument("--force-device", type=int, default=None, help="rehearsal only: put every rank on this HIP device")
	ap.add_argument("--output-candidates", type=int, default=4, help="device buffers v2m_alloc_output may hold at once to choose the output buffer from (as many as fit are tried; 1 = plain allocation)")
	ap.add_argument("--cpu-baseline-rows", type=int, default=320, help="haplotypes (plus REF) the CPU oracle is timed on, on rank 0 after the timed region at every N (320 rows of config 3 = 32 Gbases, about 11 s on one core); 0 disables")
	ap.add_argument("--verify-rows", type=int, default=-1, help="after timing, every rank re-runs every batch of its step and checks rows against the CPU oracle by device checksum: -1 (default) = EVERY row of every batch (config 3: all 5009 rows, ~25 s of oracle time on the job's 16 quota cores, dealt over the ranks at N > 1); k > 0 = k rows per batch plus REF and the last batch's ragged final group; 0 disables")
	ap.add_argument("--unaligned-rows", type=int, default=256, help="the separately timed --unaligned leg (rank 0, after the main timing) runs on as many rows as the output buffer holds and, beside it, on the first this-many rows; 0 disables")
	ap.add_argument("--cpu-transpose", type=int, default=1, help="after timing (rank 0, every N): time the CPU oracle's transpose_matrix on rank 0's matrix and compare it bit for bit with the GPU's dense-form result (config 3: ~5 s); 0 disables")
	ap.add_argument("--transpose-extras", type=int, default=1, help="after timing, also measure the inverse transpose and a 1024-bit-padded matrix (rank 0); 0 disables")
	ap.add_argument("--e2e-gb", type=float, default=64.0, help="the end-to-end leg (every rank, last: after the main timing and rank 0's other legs, by which time the driver has finished wiping the output-buffer candidates that v2m_alloc_output freed): this many GB of the rank's rows through v2m_splice_rows -- device slots, D2H copies on the copy stream, pinned slots -- into a C sink that checksums every row on the host; 0 disables")
	ap.add_argument("--e2e-threads", type=int, default=0, help="host threads of the end-to-end leg's checksumming sink, PER RANK; 0 = by the CPU and the job's quota: 4 where the sink has its AVX-512DQ loop (41 GB/s per thread on the boxes' Zen 5 cores, profiles/r04/cpu_quota_and_sink_rates.txt: twice what the link delivers), 12 with the scalar loop (7 GB/s per thread), and never more than this rank's share of the job's CPU quota (cgroup cpu.max / affinity mask, divided by LOCAL_WORLD_SIZE, one core left for the rank's main thread): what goes beyond the quota gets the whole job throttled")
	ap.add_argument("--host-threads", type=int, default=0, help="threads per rank for the CPU oracle's row checks after timing; 0 = this rank's share of the job's CPU quota (sharding.host_threads_per_rank), at most 16")
	ap.add_argument("--numa-bind", default="auto", choices=["auto", "on", "off"], help="keep this rank's threads (and with them the pinned slots they first touch) on its GPU's NUMA node: auto = at N > 1 only")
	ap.add_argument("--hub-selftest", action="store_true", help="no GPU work: the ranks only exercise the barrier / gather plumbing of their launch form and rank 0 prints what it gathered (CPU test suite)")
	args = ap.parse_args()

	rank = int(os.environ.get("RANK", "0"))
	local_rank = int(os.environ.get("LOCAL_RANK", "0"))
	world = int(os.environ.get("WORLD_SIZE", "1"))
	if "WORLD_SIZE" not in os.environ and args.gpus > 1:
		return launch_ranks(args.gpus, sys.argv[1:])
	if world != args.gpus:
		sys.exit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))
	# Only the result line may appear on stdout: file descriptor 1 is pointed at stderr for everything else in this process
	# (c10d / gloo / RCCL print connection chatter from C++), and the line goes out through a private copy of the real stdout.
	sys.stdout.flush()
	result_out = os.fdopen(os.dup(1), "w")
	os.dup2(2, 1)

	under_parent = os.environ.get(HUB_ENV) == "1"
	local_world = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))
	if args.hub_selftest:
		hub = SoloHub() if world == 1 else PipeHub(rank, result_out) if under_parent else TorchHub(args.dist_backend, rank, None)
		hub.barrier()
		t_begin = time.perf_counter()
		hub.barrier()
		everyone = hub.gather({"rank": rank, "pid": os.getpid(), "elapsed_s": time.perf_counter() - t_begin})
		if rank == 0:
			result_out.write(json.dumps({"hub_selftest": everyone, "ranks_coordinated_by": hub.kind}) + "\n")
			result_out.flush()
		hub.close()
		return

	import torch

	dev_index = local_rank if args.force_device is None else args.force_device
	torch.cuda.set_device(dev_index)
	dev = torch.device("cuda", dev_index)
	if world == 1:
		hub = SoloHub()
	elif under_parent:
		hub = PipeHub(rank, result_out)
	else:
		hub = TorchHub(args.dist_backend, rank, dev)

	from vcf2multialign_amd import build as _build
	if not under_parent:            # (the parent of `python bench.py --gpus N` has built before it started the ranks)
		if local_rank == 0:
			try:
				_build.build_native()   # no-op when the in-tree libraries are newer than their sources (hipcc cross-compiles gfx950)
			except _build.StaleUnderProfiler as e:   # never a compiler launcher under a profiler's preload: build first, then profile
				sys.exit("[bench] " + str(e))
			build_checker_once()
		hub.barrier()               # nobody loads the libraries before the (possible) rebuild is over
	import vcf2multialign_amd as v2m
	from vcf2multialign_amd import _native as N
	from vcf2multialign_amd import synth
	from vcf2multialign_amd.sharding import cpu_quota, host_threads_per_rank, shard_copies

	# ---- the host side of this rank: sized from the JOB's quota, divided among the node's ranks ------------------------------
	quota_cores, quota_source = cpu_quota()
	host_threads = args.host_threads if args.host_threads > 0 else host_threads_per_rank(local_world, cap=16)
	where = device_binding(dev_index)
	if where["hip_device"] is not None and where["hip_device"] != dev_index:   # (cannot happen after torch.cuda.set_device; a rank on the wrong GPU must not produce a figure)
		sys.exit("[bench] rank %d: the HIP runtime's current device is %d, expected %d" % (rank, where["hip_device"], dev_index))
	where["numa_bound_to_cpus"] = bind_to_numa_node(where["numa_node"]) if ("on" == args.numa_bind or ("auto" == args.numa_bind and world > 1)) else None
	log("[bench] rank %d/%d (local %d/%d, pid %d): HIP device %s = %s, PCI %s, NUMA node %s; CPUs allowed %s, bound to %s; job quota %d cores (%s) -> %d host threads for this rank"
		% (rank, world, local_rank, local_world, os.getpid(), where["hip_device"], where["name"], where["pci_bus_id"], where["numa_node"], where["cpus_allowed"],
		where["numa_bound_to_cpus"] or "nothing narrower", quota_cores, quota_source, host_threads))

	# ---- workload: generated on every rank (deterministic), resident in HBM before timing ----------
	t0 = time.time()
	ds = synth.dataset(args.config, **({"samples": args.samples} if args.samples > 0 else {}))
	g = ds.graph
	L, R, NN, E = g.aligned_length, len(ds.reference), g.node_count, g.edge_count
	H = ds.n_copies
	Ep = ds.path_rows          # 64 * ceil(E / 64): the reference's own padding (variant_graph.cc:449)
	if rank == 0:
		log("[bench] %s: R=%d variants/edges=%d nodes=%d L=%d copies=%d (generated in %.1fs)" % (args.config, R, E, NN, L, H, time.time() - t0))

	ctx = v2m.Context(dev_index)
	ctx.upload_graph(g, ds.reference)
	pitch = ctx.min_row_pitch

	# this rank's copies, zero-padded to a multiple of 64 rows of the transpose input (variant_graph.cc:277) -- nothing more
	c0, c1, hp_local = shard_copies(H, world, rank)
	n_local_copies = c1 - c0
	rows = ([v2m.PLOIDY_MAX] if rank == 0 else []) + list(range(n_local_copies))   # local copy indices into this rank's matrix
	n_rows = len(rows)
	total_rows = H + 1

	# equal-sized batches of at most --batch-rows rows (5009 rows -> 8 x 627 rather than 7 x 640 + 529)
	max_batch_rows = args.batch_rows if args.batch_rows > 0 else max(1, int(args.batch_gb * 1e9) // pitch)
	n_batches = max(1, -(-n_rows // max(1, max_batch_rows)))
	batch_rows = max(1, -(-n_rows // n_batches))
	out_bytes = batch_rows * pitch
	# Output buffer: which physical memory it lands on matters on this hardware, so the library picks it by measurement.
	out_ptr = ctx.alloc_output(out_bytes, candidates=args.output_candidates)

	def make_paths(n_rows_bits, n_cols_bits, copy_base, copy_end):
		"""paths_by_edge_and_chrom_copy of the given copies in HBM (+ an equally sized destination)."""
		words = n_cols_bits // 64 * n_rows_bits
		src = torch.empty(max(words, 1), dtype=torch.int64, device=dev)
		dst = torch.empty(max(words, 1), dtype=torch.int64, device=dev)
		torch.cuda.synchronize()
		if n_rows_bits:
			ds.fill_paths_device(ctx.stream, src.data_ptr(), thr.data_ptr(), copy_base=copy_base, n_rows=n_rows_bits, n_cols=n_cols_bits, copy_end=copy_end)
		ctx.synchronize()
		return src, dst

	thr = torch.from_numpy(ds.edge_thresholds.astype(np.int64)).to(torch.int32).to(dev) if E else torch.zeros(1, dtype=torch.int32, device=dev)
	paths_src, paths_dst = make_paths(hp_local, Ep, c0, c1)

	batches = [v2m.RowBatch(rows[i:i + batch_rows]) for i in range(0, n_rows, batch_rows)]

	def step():
		if hp_local:
			# transpose_matrix at its one call site (variant_graph.cc:453): the transpose input, in the reference's layout, into the
			# library's own copy of paths_by_chrom_copy_and_edge (which only the GPU reads; its columns are line-aligned)
			ctx.bind_path_matrix_device(paths_src.data_ptr(), hp_local, Ep)
		for b in batches:
			ctx.splice_rows_device(b, out_ptr, pitch)

	for _ in range(args.warmup):
		step()
	ctx.synchronize()
	torch.cuda.synchronize()
	hub.barrier()
	ctx.profile_enable(True)
	ctx.profile_reset()
	t_begin = time.perf_counter()
	for _ in range(args.steps):
		step()
	ctx.synchronize()
	torch.cuda.synchronize()
	elapsed = time.perf_counter() - t_begin
	hub.barrier()
	ctx.profile_enable(False)

	# ---- this rank's kernel figures, from HIP events on the kernels' own stream ----------------------
	launches, splice_ms = ctx.profile_get(N.KERNEL_SPLICE_ALIGNED)
	per_launch = ctx.profile_launches(N.KERNEL_SPLICE_ALIGNED)
	if rank == 0 and os.environ.get("V2M_BENCH_LOG_LAUNCHES"):
		log("[bench] splice launches (ms): " + " ".join("%.2f" % x for x in per_launch))
	_, resolve_ms = ctx.profile_get(N.KERNEL_RESOLVE)
	transpose_launches, transpose_ms = ctx.profile_get(N.KERNEL_TRANSPOSE)
	label_bytes = len(g.label_bytes)
	# algorithmic bytes of one launch over Hb rows (SURVEY.md 8d / BASELINE.md):
	#   Hb*L + Hb*Ep/8 + R + 24*N + 8*E + sum|label|
	shared_bytes = R + 24 * NN + 8 * E + label_bytes
	alg_bytes_total = sum(b.n_rows * L + b.n_rows * Ep // 8 + shared_bytes for b in batches)
	alg_bytes_per_launch = alg_bytes_total / max(1, len(batches))
	mine = {
		"rank": rank, "rows": n_rows, "batches": len(batches), "elapsed_s": elapsed,
		"launches": launches, "splice_ms": splice_ms, "algorithmic_bytes_per_launch": alg_bytes_per_launch,
		"resolve_ms": resolve_ms, "transpose_launches": transpose_launches, "transpose_ms": transpose_ms,
		"binding": where, "host_threads": host_threads,
	}

	# ---- CPU oracle: every rank checks rows of every batch of its own shard ----------------------------
	def oracle_graph(copies):
		"""Oracle graph whose path matrix holds the CPU re-derivation (genotype hash) of the given global copies."""
		import oracle
		n_cols = 64 * ((len(copies) + 63) // 64)
		cols = [ds.copy_column(c) for c in copies] + [np.zeros(Ep // 64, np.uint64)] * (n_cols - len(copies))
		return oracle.graph_from_arrays(g.reference_positions, g.aligned_positions, g.alt_edge_targets, g.alt_edge_count_csum,
			g.label_offsets, g.label_bytes, np.concatenate(cols) if (Ep and cols) else np.zeros(0, np.uint64), Ep, n_cols,
			["S%d" % i for i in range(n_cols // ds.ploidy)], np.arange(n_cols // ds.ploidy + 1, dtype=np.uint32) * ds.ploidy)

	mine["parity_ok"], mine["parity_checked"], mine["parity_all_rows"] = True, 0, args.verify_rows < 0   # (a rank that owns no row has checked all of them)
	oracle_sums = {}     # local row index -> the oracle's checksum of that (aligned) row, kept for the end-to-end leg's check
	local_of = lambda bi, r: rows[bi * batch_rows + r]

	def oracle_rows_of(local_rows_):
		"""(oracle graph over the copies these local rows name, the oracle's row arguments for them)."""
		import oracle
		copies = sorted({c0 + lr for lr in local_rows_ if lr != v2m.PLOIDY_MAX})
		col_of = {c: i for i, c in enumerate(copies)}
		return oracle_graph(copies), [oracle.PLOIDY_MAX if lr == v2m.PLOIDY_MAX else col_of[c0 + lr] for lr in local_rows_]

	if args.verify_rows < 0 and batches:
		# EVERY row of every batch of this rank's step: the batch is re-run, every row reduced to its 64-bit checksum on the device, and the
		# oracle walks the same rows on this rank's host threads (the columns of one batch resident at a time).  "bit-exact vs CPU" is total.
		import oracle
		t_o, ok, n_checked, full_compare = time.time(), True, 0, None
		for bi, b in enumerate(batches):
			ctx.splice_rows_device(b, out_ptr, pitch)
			got = ctx.checksum_rows_device(out_ptr, pitch, b.n_rows, length=L)
			part = [local_of(bi, r) for r in range(b.n_rows)]
			og, want_rows = oracle_rows_of(part)
			want_sums, want_len = og.row_checksums(ds.reference, want_rows, threads=min(host_threads, len(want_rows)))
			bad = np.nonzero((got != want_sums) | (want_len != L))[0]
			if len(bad):
				ok = False
				log("[bench] rank %d PARITY: batch %d, %d rows differ from the oracle's (first: row %d of the batch)" % (rank, bi, len(bad), int(bad[0])))
			for r in range(b.n_rows):
				oracle_sums[bi * batch_rows + r] = int(want_sums[r])
			n_checked += b.n_rows
			if bi == len(batches) - 1:            # one row byte for byte: the very last row of the rank's step
				r = b.n_rows - 1
				body = og.output_sequence(ds.reference, copy_index=want_rows[r]) if want_rows[r] != oracle.PLOIDY_MAX else og.output_sequence(ds.reference)
				full_compare = _device_bytes(out_ptr + r * pitch, L) == body
			del og
		ok = ok and full_compare is not False
		mine["parity_ok"], mine["parity_checked"], mine["parity_all_rows"] = ok, n_checked, True
		mine["parity_oracle_s"] = round(time.time() - t_o, 2)
		log("[bench] rank %d parity: ALL %d rows of %d batches against the oracle in %.1f s on %d threads: %s" % (rank, n_checked, len(batches), time.time() - t_o, host_threads, "bit-exact" if ok else "MISMATCH"))
	elif args.verify_rows and batches:
		import oracle
		# which rows: per batch `verify_rows` rows spread over the batch (the first batch's includes row 0: REF on rank 0), and
		# every row of the last batch's final, ragged 16-row group (the rows a wrong grid or tile bound would lose first)
		picks = []   # (batch index, row within batch)
		for bi, b in enumerate(batches):
			for k in range(args.verify_rows):
				picks.append((bi, (k * b.n_rows // args.verify_rows + 37 * bi) % b.n_rows if (bi or k) else 0))
		last = batches[-1]
		tail = last.n_rows % 16 or min(16, last.n_rows)
		picks += [(len(batches) - 1, r) for r in range(last.n_rows - min(tail, 4), last.n_rows)]
		picks = sorted(set(picks))
		og, want_rows = oracle_rows_of([local_of(bi, r) for bi, r in picks])
		t_o = time.time()
		want_sums, want_len = og.row_checksums(ds.reference, want_rows, threads=min(host_threads, len(want_rows)))
		ok = bool((want_len == L).all())
		got_sums = np.zeros(len(picks), dtype=np.uint64)
		full_compare = None
		for bi, b in enumerate(batches):          # re-run every batch of this rank's step and look at the picked rows on the device
			mine_i = [i for i, (pb, _) in enumerate(picks) if pb == bi]
			if not mine_i:
				continue
			ctx.splice_rows_device(b, out_ptr, pitch)
			for i in mine_i:
				got_sums[i] = ctx.checksum_rows_device(out_ptr + picks[i][1] * pitch, pitch, 1, length=L)[0]
			if bi == len(batches) - 1:            # one row of the last batch byte for byte: its very last row
				i = mine_i[-1]
				body = og.output_sequence(ds.reference, copy_index=want_rows[i]) if want_rows[i] != oracle.PLOIDY_MAX else og.output_sequence(ds.reference)
				full_compare = _device_bytes(out_ptr + picks[i][1] * pitch, L) == body
		ok = ok and bool(np.array_equal(got_sums, want_sums)) and full_compare is not False
		mine["parity_ok"], mine["parity_checked"], mine["parity_all_rows"] = ok, len(picks), False
		log("[bench] rank %d parity: %d rows of %d batches against the oracle in %.1f s: %s" % (rank, len(picks), len(batches), time.time() - t_o, "bit-exact" if ok else "MISMATCH"))
		del og

	# ================= rank 0, before the end-to-end leg: the legs that only it runs =========================================
	# (They come first on purpose: v2m_alloc_output has just written four 63-GB candidates and freed three, the driver wipes freed VRAM that was
	# written in the background for about nine seconds, and while it does every D2H copy of the process runs at 40 instead of 56 GB/s
	# (profiles/r04/e2e_slow_after_alloc_output.txt).  That wipe is an artefact of choosing the output buffer by measurement, not something a
	# caller of output_a2m pays -- the command-line driver never calls v2m_alloc_output -- so the end-to-end leg must not be timed inside it.
	# The other ranks wait at the leg's first barrier meanwhile.)
	extras = {}
	if rank == 0:
		# ---- the transpose, at the reference's padding: algorithmic 2 * Hp * Ep / 8 bytes per call -----------------------
		tr_bytes = 2 * hp_local * Ep // 8
		if transpose_launches:
			t_ms = transpose_ms / transpose_launches
			extras["roofline_transpose"] = {
				"bound": "hbm", "kernel": "transpose_bits (v2m_bind_path_matrix_device: source in the reference's layout and padding, destination the library's own line-aligned copy; the kernel is chosen per matrix shape by measurement, see config.tuning)",
				"rank": 0, "matrix_bits": [hp_local, Ep], "algorithmic_bytes": tr_bytes, "launches": transpose_launches,
				"avg_launch_ms": round(t_ms, 4), "achieved": round(tr_bytes / t_ms / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
				"frac": round(tr_bytes / t_ms / 1e6 / HBM_PEAK_GBS, 4),
			}

		def time_transpose(src_ptr, n_r, n_c, dst_ptr, reps=5):
			ctx.transpose_bits_device(src_ptr, n_r, n_c, dst_ptr)   # first call of a shape: the library times its candidates
			ctx.synchronize()
			ctx.profile_enable(True)
			ctx.profile_reset()
			for _ in range(reps):
				ctx.transpose_bits_device(src_ptr, n_r, n_c, dst_ptr)
			n, ms = ctx.profile_get(N.KERNEL_TRANSPOSE)
			ctx.profile_enable(False)
			return ms / max(1, n)

		if args.transpose_extras and hp_local:
			# the ABI's dense form (v2m_transpose_bits_device: caller-visible destination, so both sides at the reference's padding)
			fwd_ms = time_transpose(paths_src.data_ptr(), hp_local, Ep, paths_dst.data_ptr())
			back = torch.empty_like(paths_src)
			inv_ms = time_transpose(paths_dst.data_ptr(), Ep, hp_local, back.data_ptr())
			involution = bool(torch.equal(back, paths_src))
			del back
			pad1024 = lambda n: (n + 1023) // 1024 * 1024
			hp_p, ep_p = pad1024(hp_local), pad1024(Ep)
			p_src, p_dst = make_paths(hp_p, ep_p, c0, c1)
			fwd_p = time_transpose(p_src.data_ptr(), hp_p, ep_p, p_dst.data_ptr())
			inv_p = time_transpose(p_dst.data_ptr(), ep_p, hp_p, p_src.data_ptr())
			del p_src, p_dst
			gbs = lambda nbytes, ms: round(nbytes / ms / 1e6, 1)
			extras.setdefault("roofline_transpose", {})["after_timing"] = {
				"dense_forward_ms": round(fwd_ms, 4), "dense_forward_GBs": gbs(tr_bytes, fwd_ms), "dense_forward_frac": round(tr_bytes / fwd_ms / 1e6 / HBM_PEAK_GBS, 4),
				"inverse_ms": round(inv_ms, 4), "inverse_GBs": gbs(tr_bytes, inv_ms), "inverse_frac": round(tr_bytes / inv_ms / 1e6 / HBM_PEAK_GBS, 4), "involution_bit_exact": involution,
				"note": "dense_forward / inverse: v2m_transpose_bits_device with a caller-visible destination (both sides at the reference's 64-bit padding); padded_1024: the same with both dimensions padded to 1024 bits",
				"padded_1024": {"matrix_bits": [hp_p, ep_p], "forward_ms": round(fwd_p, 4), "forward_GBs": gbs(2 * hp_p * ep_p // 8, fwd_p), "inverse_ms": round(inv_p, 4), "inverse_GBs": gbs(2 * hp_p * ep_p // 8, inv_p)},
				"kernels": "; ".join(note for note in ctx.info.split("; ") if note.startswith("transpose ")),   # which kernel the library measured fastest for each of these shapes
			}
			if not involution:
				log("[bench] PARITY FAILURE: transpose(transpose(m)) != m")
				extras["parity_failed"] = True
			ctx.bind_path_matrix_device(paths_src.data_ptr(), hp_local, Ep)

		# Context for the roofline number: what a plain device memset of the very same output buffer reaches in this
		# process (the achievable write rate varies by +-10 % between processes / boxes).
		ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
		fills = []
		for _ in range(3):
			ev0.record()
			_hip_memset(torch, out_ptr, out_bytes)
			ev1.record()
			torch.cuda.synchronize()
			fills.append(ev0.elapsed_time(ev1))
		extras["memset_same_buffer_GBs"] = round(out_bytes / (min(fills) * 1e-3) / 1e9, 1)
		extras["memset_same_buffer_rank"] = 0

		# ---- second leg, timed on its own: --unaligned (sequence_writer.cc:80: no '-' padding) -----------------------------
		# Measured on the SAME address footprint as the aligned leg (as many rows as the ~63-GB output buffer holds at the unaligned
		# pitch: the store pattern only reaches its full rate when a launch spans several tens of GB), and, for
		# continuity with rounds 1-2, on the first --unaligned-rows (256) rows as well.
		if args.unaligned_rows and n_rows > 1:
			import oracle
			upitch = (ctx.max_unaligned_length + 255) // 256 * 256
			n_tiles = -(-L // 16384)
			aligned_ms_per_base = (splice_ms / max(1, launches)) / max(1, batch_rows * L)

			def unaligned_leg(n_u, reps=3):
				ub = v2m.RowBatch(rows[:n_u])
				# like for like: the ALIGNED kernel on the very same rows, buffer and pitch (what a launch reaches depends on the address range it
				# covers and on the buffer's backing -- DESIGN.md section 4, profiles/r04/unaligned_footprint_4_buffers.txt -- so the full-footprint
				# aligned launches of the timed region are the wrong yardstick for a 256-row launch)
				a_same = None
				if upitch >= (L + 15) // 16 * 16:
					ctx.splice_rows_device(ub, out_ptr, upitch)
					ctx.synchronize()
					ctx.profile_enable(True)
					ctx.profile_reset()
					for _ in range(reps):
						ctx.splice_rows_device(ub, out_ptr, upitch)
					n_a, ms_a = ctx.profile_get(N.KERNEL_SPLICE_ALIGNED)
					ctx.profile_enable(False)
					a_same = ms_a / max(1, n_a)
				ctx.splice_rows_device(ub, out_ptr, upitch, unaligned=True)      # warm-up (builds the second template; first >= 1-GiB launch: calibrates the store flavour)
				ctx.synchronize()
				ctx.profile_enable(True)
				ctx.profile_reset()
				t_u = time.perf_counter()
				for _ in range(reps):
					lengths = ctx.splice_rows_device(ub, out_ptr, upitch, unaligned=True, want_lengths=True)
				ctx.synchronize()
				wall_u = (time.perf_counter() - t_u) / reps
				_, u_count = ctx.profile_get(N.KERNEL_UNALIGNED_COUNT)
				_, u_splice = ctx.profile_get(N.KERNEL_SPLICE_UNALIGNED)
				_, u_resolve = ctx.profile_get(N.KERNEL_RESOLVE)
				ctx.profile_enable(False)
				u_count, u_splice, u_resolve = u_count / reps, u_splice / reps, u_resolve / reps
				bases_u = int(lengths.sum())
				# pass 2 (splice_unaligned_kernel): row bytes out + the rows' bit columns + the shared inputs + one tile offset per (row, tile);
				# pass 1 (count + scan): the shared inputs + bit columns in, the tile counts out and in and out again (scan in place)
				alg_u = bases_u + n_u * Ep // 8 + shared_bytes + 4 * n_u * n_tiles
				alg_c = n_u * Ep // 8 + shared_bytes + 3 * 4 * n_u * n_tiles
				usample = sorted({0, 1, n_u // 2, n_u - 1})
				ucopies = [c0 + rows[i] for i in usample if rows[i] != v2m.PLOIDY_MAX]
				uog = oracle_graph(ucopies)
				ucol = {c: i for i, c in enumerate(ucopies)}
				uwant, ulen = uog.row_checksums(ds.reference, [oracle.PLOIDY_MAX if rows[i] == v2m.PLOIDY_MAX else ucol[c0 + rows[i]] for i in usample], unaligned=True, threads=len(usample))
				ugot = np.array([ctx.checksum_rows_device(out_ptr + i * upitch, upitch, 1, length=int(lengths[i]))[0] for i in usample], dtype=np.uint64)
				u_ok = bool(np.array_equal(ulen, lengths[usample])) and bool(np.array_equal(ugot, uwant))
				ms_per_base = u_splice / max(1, bases_u)
				return {
					"rows": n_u, "bases": bases_u, "footprint_GB": round(n_u * upitch / 1e9, 2),
					"value": round(bases_u / (u_resolve + u_count + u_splice) / 1e6, 1), "unit": "Gbases/s", "wall_ms_per_batch": round(1e3 * wall_u, 3),
					"kernels_ms": {"resolve_effective_edges_kernel": round(u_resolve, 3), "count_unaligned_kernel+scan_tile_counts_kernel": round(u_count, 3), "splice_unaligned_kernel": round(u_splice, 3)},
					"roofline": {"bound": "hbm", "kernel": "splice_unaligned_kernel", "algorithmic_bytes_per_launch": int(alg_u), "achieved": round(alg_u / u_splice / 1e6, 1), "peak": HBM_PEAK_GBS,
						"unit": "GB/s", "frac": round(alg_u / u_splice / 1e6 / HBM_PEAK_GBS, 4)},
					"time_per_base_vs_aligned_kernel": round(ms_per_base / aligned_ms_per_base, 3) if aligned_ms_per_base > 0 else None,
					"aligned_kernel_same_rows_ms": round(a_same, 3) if a_same else None,
					"time_per_base_vs_aligned_kernel_same_rows": round(ms_per_base / (a_same / (n_u * L)), 3) if a_same else None,
					"yardsticks": "time_per_base_vs_aligned_kernel: against the timed region's aligned launches (the output buffer's whole footprint); ..._same_rows: against the aligned kernel on these very rows, this buffer and this pitch",
					"roofline_count_pass": {"bound": "hbm", "kernel": "count_unaligned_kernel+scan_tile_counts_kernel", "algorithmic_bytes_per_launch": int(alg_c), "achieved": round(alg_c / u_count / 1e6, 1),
						"unit": "GB/s", "note": "reads the shared inputs and the rows' effective-edge bits, writes 4 bytes per (row, 16-KiB tile); builds no row"},
					"parity": {"rows_checked": len(usample), "bit_exact": u_ok, "method": "row lengths and device checksums against the CPU oracle's unaligned rows"},
				}

			n_full = max(1, min(out_bytes // upitch, n_rows))
			n_small = max(1, min(args.unaligned_rows, n_full))
			small = unaligned_leg(n_small)
			full = unaligned_leg(n_full) if n_full != n_small else small
			extras["unaligned"] = dict(full, metric="unaligned (--unaligned) Gbases/sec, one batch on the aligned leg's output buffer (same ~%.0f-GB address footprint), kernels only, rank 0" % (out_bytes / 1e9),
				first_rows_only=small, tuning=ctx.info)
			for leg in (small, full):
				if not leg["parity"]["bit_exact"]:
					log("[bench] PARITY FAILURE (unaligned leg, %d rows) against the CPU oracle" % leg["rows"])
					extras["parity_failed"] = True

		# ---- the CPU path beside the GPU transpose ("time the whole matrix on CPU, one call per run") ---------
		if args.cpu_transpose and hp_local and "roofline_transpose" in extras:
			import oracle
			host_src = np.frombuffer(_device_bytes(paths_src.data_ptr(), tr_bytes // 2), dtype=np.uint64)      # copied to the host once, after all GPU timing
			ctx.transpose_bits_device(paths_src.data_ptr(), hp_local, Ep, paths_dst.data_ptr())                # the ABI's dense form: caller-visible destination
			ctx.synchronize()
			gpu_dense = np.frombuffer(_device_bytes(paths_dst.data_ptr(), tr_bytes // 2), dtype=np.uint64)
			t_c = time.perf_counter()
			cpu_dst = oracle.transpose_matrix(host_src, hp_local, Ep)          # transpose_matrix.cc:41-109's traversal, one thread, -O2
			secs_c = time.perf_counter() - t_c
			same = bool(np.array_equal(cpu_dst, gpu_dense))
			gpu_ms = extras["roofline_transpose"]["avg_launch_ms"]
			extras["roofline_transpose"]["cpu_baseline"] = {
				"seconds": round(secs_c, 6), "value": round(tr_bytes / secs_c / 1e9, 4), "unit": "GB/s", "cores": 1, "kind": "port",
				"sample": "the whole %d x %d-bit matrix of rank 0 in this run (%.3f GB read + written), one call of the oracle's v2mo_transpose_matrix (the 8x8-block traversal of transpose_matrix.cc:41-109), including its zero-fill of the destination; host has %d logical CPUs"
					% (hp_local, Ep, tr_bytes / 1e9, os.cpu_count()),
				"bit_exact_vs_gpu_dense_form": same, "gpu_over_cpu": round(secs_c * 1e3 / gpu_ms, 1) if gpu_ms > 0 else None,
			}
			if not same:
				log("[bench] PARITY FAILURE: the GPU's dense-form transpose differs from the CPU oracle's")
				extras["parity_failed"] = True
			del host_src, gpu_dense, cpu_dst

		# ---- the CPU path beside the GPU splice: same run, same host, at every N (the other ranks have finished or wait) -------
		if args.cpu_baseline_rows:
			nb = min(args.cpu_baseline_rows, H)
			og = oracle_graph(list(range(nb)))
			nbytes, secs = og.haplotype_output_a2m(ds.reference, None, first_copy=0, n_copies=nb)
			bases = (nb + 1) * L
			extras["cpu_baseline"] = {
				"value": round(bases / secs / 1e9, 4), "unit": "Gbases/s", "cores": 1, "kind": "port",
				"sample": "REF + first %d haplotypes of %s (%.2f Gbases) through the oracle's output_sequence/haplotype_output_a2m (sequence_writer.cc:22-85 driven by haplotype_output.cc:38-82) into a discarding std::ostream, %.1f s, on rank 0's host cores after the timed region; rows are independent, so the rate carries to the full %d rows; host has %d logical CPUs"
					% (nb, args.config, bases / 1e9, secs, total_rows, os.cpu_count()),
			}
			# Context only, and NOT the reference's behaviour (it writes every row through one std::ostream on one thread):
			# the same rows dealt to as many threads as the job's CPU quota allows (16 on a GPU box), each walking its rows into its own
			# discarding stream.  It is what the end-to-end figure, not `value`, should be read against.
			from concurrent.futures import ThreadPoolExecutor
			quota = max(1, min(16, quota_cores))   # the whole job's quota: the other ranks wait at a barrier meanwhile
			share = -(-nb // quota)
			parts = [(c, min(share, nb - c)) for c in range(0, nb, share)]
			t_all = time.perf_counter()
			with ThreadPoolExecutor(max_workers=len(parts)) as ex:
				list(ex.map(lambda part: og.haplotype_output_a2m(ds.reference, None, output_reference=False, first_copy=part[0], n_copies=part[1]), parts))
			secs_all = time.perf_counter() - t_all
			extras["cpu_baseline"]["rows_dealt_to_threads"] = {
				"value": round(nb * L / secs_all / 1e9, 4), "unit": "Gbases/s", "cores": len(parts), "kind": "port",
				"note": "not the reference's behaviour (one thread, one stream): the same %d haplotypes dealt to %d threads (the job's CPU quota), each into its own discarding stream, %.2f s" % (nb, len(parts), secs_all),
			}


	# ---- end to end: what a caller of output::output_a2m gets (output.cc:47-76) -- every row crosses the link ------------
	# v2m_splice_rows (the sink form the command-line driver uses): the batch is spliced slice by slice into two device slots,
	# each slice's D2H copy runs on the copy stream under the next slice's kernels, and every finished row is handed to the sink
	# from the library's pinned slot.  The sink here is C (libv2m_synth.so: v2ms_checksum_sink_fn): it reads every byte of every
	# row -- the checksum of v2m_checksum_rows_device, on --e2e-threads host threads -- and keeps nothing.  All ranks run the leg
	# together (one PCIe link each), between barriers.
	e2e_rows = min(n_rows, max(1, int(args.e2e_gb * 1e9) // max(1, L))) if args.e2e_gb > 0 and n_rows else 0
	if args.e2e_gb > 0:
		import ctypes as C
		sl = C.CDLL(_build.SYNTH_LIB_PATH)
		sl.v2ms_checksum_sink_create.restype = C.c_void_p
		sl.v2ms_checksum_sink_create.argtypes = [C.c_uint64, C.c_uint32]
		sl.v2ms_checksum_sink_destroy.argtypes = [C.c_void_p]
		for name in ("rows", "bytes"):
			getattr(sl, "v2ms_checksum_sink_" + name).restype = C.c_uint64
			getattr(sl, "v2ms_checksum_sink_" + name).argtypes = [C.c_void_p]
		for name in ("checksums", "lengths"):
			getattr(sl, "v2ms_checksum_sink_" + name).restype = C.POINTER(C.c_uint64)
			getattr(sl, "v2ms_checksum_sink_" + name).argtypes = [C.c_void_p]
		sl.v2ms_checksum_sink_flavour.restype = C.c_char_p
		sl.v2ms_checksum_sink_flavour.argtypes = [C.c_void_p]
		sink_fn = C.cast(sl.v2ms_checksum_sink_fn, N.SINK_FN)
		sink_flavour = [None]
		if args.e2e_threads <= 0:
			probe = sl.v2ms_checksum_sink_create(1, 1)
			by_cpu = 4 if sl.v2ms_checksum_sink_flavour(probe) == b"avx512dq" else 12
			sl.v2ms_checksum_sink_destroy(probe)
			# never more than this rank's share of the job's quota, with one core per rank left for the rank's own thread (it drives the
			# copies and hands the rows over).  A 16-core quota gives 4 sink threads at N = 1 and 1 at N = 8 (one AVX-512DQ thread reads
			# 41 GB/s, below a link's 57: the line then shows it in end_to_end.sink); a quota of 16 cores per GPU gives 4 at every N.
			# DESIGN.md section 7 has the budget of the leg at 8 links.
			args.e2e_threads = max(1, min(by_cpu, host_threads_per_rank(local_world, cap=by_cpu, reserve=local_world)))

		def through_the_sink(batch):
			state = sl.v2ms_checksum_sink_create(max(1, batch.n_rows), max(1, args.e2e_threads))
			sink_flavour[0] = sl.v2ms_checksum_sink_flavour(state).decode()
			try:
				t_s = time.perf_counter()
				rc = ctx._lib.v2m_splice_rows(ctx._h, C.byref(batch.struct), 0, sink_fn, state)
				secs = time.perf_counter() - t_s
				ctx._check(rc)
				n = int(sl.v2ms_checksum_sink_rows(state))
				sums = np.ctypeslib.as_array(sl.v2ms_checksum_sink_checksums(state), shape=(max(1, batch.n_rows),))[:n].copy()
				lens = np.ctypeslib.as_array(sl.v2ms_checksum_sink_lengths(state), shape=(max(1, batch.n_rows),))[:n].copy()
				return secs, n, int(sl.v2ms_checksum_sink_bytes(state)), sums, lens
			finally:
				sl.v2ms_checksum_sink_destroy(state)

		e2e_batch = v2m.RowBatch(rows[:e2e_rows]) if e2e_rows else None
		if e2e_rows:
			# first use of the sink path: its device and pinned slots are set up here (as many rows as make the library choose the slot
			# size of the timed passes: 512-MB slots from 8 GiB of rows on)
			through_the_sink(v2m.RowBatch(rows[:min(e2e_rows, max(8, (9 << 30) // max(1, L)))]))
		passes = []
		for _ in range(3 if e2e_rows else 0):
			hub.barrier()
			e_secs, e_n, e_bytes, e_sums, e_lens = through_the_sink(e2e_batch)
			passes.append(e_secs)
		if not e2e_rows:
			for _ in range(3):
				hub.barrier()   # (a rank without rows keeps the others' barriers company)
		hub.barrier()
		if e2e_rows:
			import oracle
			t_o = time.time()
			if all(i in oracle_sums for i in range(e2e_rows)):
				# the parity leg above has walked these very rows on the CPU already (every row of the rank): the same oracle checksums
				ewant, ewant_len, how = np.array([oracle_sums[i] for i in range(e2e_rows)], dtype=np.uint64), np.full(e2e_rows, L, dtype=np.uint64), "the parity leg's oracle checksums of the same rows"
			else:
				eog, ewant_rows = oracle_rows_of(rows[:e2e_rows])
				ewant, ewant_len = eog.row_checksums(ds.reference, ewant_rows, threads=host_threads)
				how = "%.1f s on %d threads" % (time.time() - t_o, host_threads)
				del eog
			# (the checksums kept are the last pass's: every pass delivers the same rows)
			e_ok = e_n == e2e_rows and e_bytes == e2e_rows * L and bool(np.array_equal(e_lens, ewant_len)) and bool(np.array_equal(e_sums, ewant))
			e_secs = sorted(passes)[len(passes) // 2]
			log("[bench] rank %d end to end: %d rows = %.1f GB through the sink in %s s (median %.1f GB/s); every row against the oracle (%s): %s"
				% (rank, e_n, e_bytes / 1e9, " / ".join("%.3f" % p for p in passes), e_bytes / e_secs / 1e9, how, "bit-exact" if e_ok else "MISMATCH"))
			mine["e2e"] = {"rows": e_n, "bytes": e_bytes, "seconds": e_secs, "passes_s": [round(p, 4) for p in passes], "bit_exact": e_ok, "sink": "%d threads, %s loop" % (args.e2e_threads, sink_flavour[0])}

	everyone = hub.gather(mine)

	if rank != 0:
		hub.close()
		ctx.free_output(out_ptr)
		ctx.close()
		return

	# ================= rank 0: the line ================================================================================
	slowest = max(everyone, key=lambda f: f["elapsed_s"])
	elapsed_max = slowest["elapsed_s"]
	value = total_rows * L * args.steps / elapsed_max / 1e9
	avg_ms = lambda f: f["splice_ms"] / max(1, f["launches"])
	per_rank = [{"rank": f["rank"], "rows": f["rows"], "batches": f["batches"], "ms_per_step": round(1e3 * f["elapsed_s"] / args.steps, 3),
		"launches": f["launches"], "avg_launch_ms": round(avg_ms(f), 4)} for f in everyone]
	# where every rank ran (HIP device as the runtime reports it, PCI bus id, the slot's NUMA node, allowed CPUs) and what its host side was given
	placement = [dict(f["binding"], rank=f["rank"], host_threads=f["host_threads"], sink=(f.get("e2e") or {}).get("sink")) for f in everyone]
	# the roofline is the SLOWEST rank's kernel average over that rank's own launches (its rows per launch may differ by a few)
	roof_rank = max(everyone, key=avg_ms)
	avg_launch_s = avg_ms(roof_rank) / 1e3
	roof_alg = roof_rank["algorithmic_bytes_per_launch"]
	achieved = roof_alg / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0

	# HBM traffic of the dominant kernel comes from separate rocprofv3 --pmc passes of this same command
	# (profiles/pmc_traffic.json).  It is quoted only for the run it was measured on: same config, rows per launch and GPU
	# count, and the kernel sources (git blob hashes) unchanged since.
	traffic, traffic_source = None, None
	try:
		with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
			rec = json.load(f).get(args.config)
		if rec and rec["batch_rows"] == batch_rows and rec["n_gpus"] == world:
			if rec.get("kernel_sources") == kernel_source_stamp():
				traffic, traffic_source = rec["hbm_bytes_per_launch"], rec["source"]
			else:
				traffic_source = "stale: %s was measured on other kernel sources" % rec["source"]
	except (OSError, ValueError, KeyError):
		pass

	result = {
		"metric": "aligned A2M Gbases/sec",
		"value": round(value, 3),
		"unit": "Gbases/s",
		"residency": "hbm",
		"residency_note": "value is the HBM-resident rate: inputs in HBM, rows written to HBM, nothing crosses PCIe in the timed region; the rate with every row delivered to the host is end_to_end.value",
		"n_gpus": world,
		"steps": args.steps,
		"warmup": args.warmup,
		"ms_per_step": round(1e3 * elapsed_max / args.steps, 3),
		"higher_is_better": True,
		"scaling": "strong",
		"vs_baseline": None,
		"dtype": "u8",
		"data": "synthetic",
		"config": {
			"workload": "%s: synthetic %d bp reference, %d variant records (%d ALT edges), %d diploid samples = %d haplotype rows + REF, --haplotypes aligned A2M, L=%d"
				% (args.config, R, ds.n_variants, E, ds.samples, H, L),
			"rows_total": total_rows, "aligned_length": L, "batch_rows": batch_rows,
			"path_matrix": "%d x %d bits on rank 0: a rank's copies x ALT edges, both padded to multiples of 64 as in the reference (variant_graph.cc:277,449), nothing more" % (hp_local, Ep),
			"sharding": "contiguous chromosome copies per rank (multiples of 8), graph + reference replicated, no collective",
			"ranks_coordinated_by": hub.kind,
			"per_rank": per_rank,
			"placement": placement,
			"host": {"cpu_quota_cores": quota_cores, "cpu_quota_source": quota_source, "logical_cpus": os.cpu_count(), "local_world_size": local_world,
				"note": "host threads per rank (oracle checks, sink) are the job's quota divided by the node's ranks, not os.cpu_count()"},
			"tuning": ctx.info,
		},
		"roofline": {
			"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
			"frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_unit": "bytes per launch (PMC: WRITE_SIZE + 2*FETCH_SIZE)", "traffic_source": traffic_source,
			"kernel": "splice_aligned_kernel", "rank": roof_rank["rank"], "of_ranks": "the slowest rank's launches (largest avg_launch_ms of config.per_rank)",
			"launches": roof_rank["launches"], "avg_launch_ms": round(1e3 * avg_launch_s, 4),
			"algorithmic_bytes_per_launch": int(roof_alg),
			"other_kernels_ms_per_step": {"resolve_effective_edges_kernel": round(roof_rank["resolve_ms"] / args.steps, 3), "transpose_bits_kernel": round(roof_rank["transpose_ms"] / args.steps, 3)},
		},
	}

	checked = sum(f["parity_checked"] for f in everyone)
	if checked:
		ok = all(f["parity_ok"] for f in everyone)
		all_rows = all(f.get("parity_all_rows") for f in everyone)
		result["parity"] = {"rows_checked": checked, "rows_total": total_rows, "all_rows": all_rows and checked == total_rows, "batches_covered": sum(f["batches"] for f in everyone), "bit_exact": ok,
			"oracle_seconds_per_rank": [f.get("parity_oracle_s") for f in everyone],
			"method": ("per rank, after timing: every batch of the step re-run and EVERY row's device checksum (v2m_checksum_rows_device) compared with the CPU oracle's checksum of its own walk of that row (sequence_writer.cc:22-85), plus the last row of each rank byte for byte" if all_rows else
				"per rank, after timing: every batch of the step re-run; device checksums (v2m_checksum_rows_device) of %d row(s) per batch, REF and the last batch's final ragged group against the CPU oracle's rows, plus the last row of the last batch byte for byte" % args.verify_rows)}
		if not ok:
			log("[bench] PARITY FAILURE against the CPU oracle")

	result["roofline"].update({k: extras.pop(k) for k in ("memset_same_buffer_GBs", "memset_same_buffer_rank") if k in extras})
	if extras.pop("parity_failed", False):
		result.setdefault("parity", {})["bit_exact"] = False
	legs = [f["e2e"] for f in everyone if f.get("e2e")]
	if legs:
		e_bytes, e_secs = sum(l["bytes"] for l in legs), max(l["seconds"] for l in legs)
		e_ok = all(l["bit_exact"] for l in legs)
		slow = min(legs, key=lambda l: l["bytes"] / l["seconds"])
		result["end_to_end"] = {
			"metric": "aligned A2M Gbases/sec delivered to a host sink (PCIe-inclusive)", "value": round(e_bytes / e_secs / 1e9, 3), "unit": "Gbases/s", "GBs": round(e_bytes / e_secs / 1e9, 3),
			"rows": sum(l["rows"] for l in legs), "bytes": e_bytes, "seconds": round(e_secs, 4),
			"path": "v2m_splice_rows: slices of the batch spliced into two device slots, D2H on the copy stream under the next slice's kernels, rows handed to a C sink from the library's pinned slots (what output::output_a2m does, output.cc:47-76); the sink reads every byte (checksum on %s) and keeps nothing; all ranks at once, one link each" % legs[0]["sink"],
			"roofline": {"bound": "pcie", "achieved": round(slow["bytes"] / slow["seconds"] / 1e9, 2), "peak": PCIE_PEAK_GBS, "unit": "GB/s", "frac": round(slow["bytes"] / slow["seconds"] / 1e9 / PCIE_PEAK_GBS, 4),
				"note": "per GPU (the slowest rank's link): PCIe Gen5 x16, 63 GB/s spec per direction; a row byte crosses the link exactly once"},
			"per_rank_GBs": [round(l["bytes"] / l["seconds"] / 1e9, 2) for l in legs],
			"passes_s": [l["passes_s"] for l in legs], "passes_note": "three passes per rank, each between barriers; seconds / value are each rank's MEDIAN pass",
			"parity": {"rows_checked": sum(l["rows"] for l in legs), "bit_exact": e_ok, "method": "length and checksum of EVERY delivered row, computed on the host inside the sink, against the CPU oracle's rows"},
		}
		if not e_ok:
			log("[bench] PARITY FAILURE (end-to-end leg) against the CPU oracle")
			result.setdefault("parity", {})["bit_exact"] = False

	result.update(extras)          # roofline_transpose, unaligned, cpu_baseline: measured before the end-to-end leg (above)

	result_out.write(json.dumps(result) + "\n")
	result_out.flush()
	hub.close()
	ctx.free_output(out_ptr)
	ctx.close()
	if result.get("parity", {}).get("bit_exact") is False:
		sys.exit(3)


if __name__ == "__main__":
	main()
