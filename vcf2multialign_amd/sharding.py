"""Multi-GPU partition of the splice path: chromosome copies shard across ranks in contiguous blocks of whole
bytes of the bit-packed path matrix (8 copies: a rank's slice of paths_by_edge_and_chrom_copy is then a strided 2-D
byte copy, bytes [c0/8, c1/8) of every column, zero-padded to a multiple of 64 rows for the transpose), graph and
reference are replicated, rows stay in file order when the ranks' outputs are concatenated.  No collective is needed on the data path
(rows are independent, haplotype_output.cc:62-81).  What ranks exchange is figures, never rows: bench.py's hub (its parent
process over pipes, or a gloo group under torch.distributed.run) serves a barrier and one gather, and rank 0 takes the
maximum of the gathered times.

cpu_quota() / host_threads_per_rank(): the host side of a rank (sink threads, checker threads) is sized from what the job
may use -- the cgroup's cpu.max and the affinity mask, not os.cpu_count() -- divided among the ranks of the node."""

import os

PLOIDY_MAX = 0xFFFFFFFF


COPY_GRANULE = 8


def shard_copies(n_copies, world, rank):
	"""Returns (first copy, end copy, padded local copy count) of `rank`.  Blocks that do not divide evenly go to the
	last ranks: rank 0 also carries the REF row."""
	n_blocks = (n_copies + COPY_GRANULE - 1) // COPY_GRANULE
	base, extra = divmod(n_blocks, world)
	first_heavy = world - extra
	b0 = rank * base + max(0, rank - first_heavy)
	b1 = b0 + base + (1 if rank >= first_heavy else 0)
	c0, c1 = min(n_copies, COPY_GRANULE * b0), min(n_copies, COPY_GRANULE * b1)
	return c0, c1, 64 * ((c1 - c0 + 63) // 64)


def local_rows(n_copies, world, rank, include_reference=True):
	"""Row specs of `rank` as (global output row index, local copy index or PLOIDY_MAX).
	Output row 0 is REF (rank 0), row 1 + c is chromosome copy c (haplotype_output.cc:48-81)."""
	c0, c1, _ = shard_copies(n_copies, world, rank)
	rows = []
	if include_reference and rank == 0:
		rows.append((0, PLOIDY_MAX))
	off = 1 if include_reference else 0
	rows.extend((off + c, c - c0) for c in range(c0, c1))
	return rows


def cpu_quota():
	"""Cores' worth of CPU time this process may use: the smaller of its affinity mask and its cgroup's cpu.max quota
	(v2: /sys/fs/cgroup/cpu.max "limit period"; v1: cpu.cfs_quota_us / cpu.cfs_period_us).  A GPU box of the pool gives a
	1-GPU job 16 cores' worth on a 256-thread host, and a process that runs more busy threads than that is throttled as a
	whole (profiles/r04/cpu_quota_and_sink_rates.txt).  Returns (cores, where the figure comes from)."""
	try:
		cores, source = len(os.sched_getaffinity(0)), "affinity mask"
	except (AttributeError, OSError):
		cores, source = os.cpu_count() or 1, "os.cpu_count()"
	for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()),
			("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: (t.strip(), open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip()))):
		try:
			with open(path) as f:
				limit, period = parse(f.read())
			if limit not in ("max", "-1") and int(period) > 0:
				q = max(1, int(limit) // int(period))
				if q < cores:
					cores, source = q, path
			break
		except (OSError, ValueError):
			continue
	return max(1, cores), source


def host_threads_per_rank(local_world=1, cap=16, reserve=0):
	"""Busy host threads one rank may run when `local_world` ranks share the node's quota: (quota - reserve) / local_world,
	at least 1 and at most `cap`."""
	cores, _ = cpu_quota()
	return max(1, min(cap, (cores - reserve) // max(1, local_world)))
