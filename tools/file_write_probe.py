#!/usr/bin/env python3
"""What one output file allows: N threads pwrite()-ing 100-MiB rows into ONE file at disjoint offsets, against N threads writing a file each
(buffered writes, tmpfs or disk) -- and, with --direct, the same single file opened O_DIRECT after posix_fallocate(): page-aligned buffers,
offsets and lengths, no page cache and (on file systems that allow it) no exclusive inode lock for overwrites of allocated extents.
Usage: python tools/file_write_probe.py DIR [total GB = 16] [--direct]"""
import mmap, os, sys, threading, time
args = [a for a in sys.argv[1:] if not a.startswith("--")]
direct = "--direct" in sys.argv
d = args[0]
total = int(float(args[1]) * 1e9) if len(args) > 1 else 16_000_000_000
ROW = 100 << 20                                              # 100 MiB: a multiple of every block size
buf = mmap.mmap(-1, ROW)                                     # page-aligned, as O_DIRECT wants (and as the library's pinned slots are)
buf.write(bytes(bytearray(os.urandom(1 << 20)) * 100))
row = memoryview(buf)
n_rows = total // ROW


def where(path):
	best = ("", "?", "?")
	for line in open("/proc/mounts"):
		dev, mnt, fs = line.split()[:3]
		if os.path.realpath(path).startswith(mnt.rstrip("/") + "/") or os.path.realpath(path) == mnt:
			if len(mnt) >= len(best[0]):
				best = (mnt, dev, fs)
	return "%s on %s (%s)" % best


def run(n_threads, one_file, o_direct=False):
	paths = [os.path.join(d, "probe_%d.bin" % (0 if one_file else t)) for t in range(n_threads)]
	flags = os.O_WRONLY | os.O_CREAT | os.O_TRUNC | (os.O_DIRECT if o_direct else 0)
	fds = [os.open(p, flags, 0o644) for p in (paths[:1] if one_file else paths)]
	if o_direct:
		for fd in fds:
			os.posix_fallocate(fd, 0, n_rows * ROW if one_file else -(-n_rows // n_threads) * ROW)   # extents allocated up front: the writes are overwrites
	def work(t):
		fd = fds[0 if one_file else t]
		for i in range(t, n_rows, n_threads):
			off = i * ROW if one_file else (i // n_threads) * ROW
			done = 0
			while done < ROW:
				done += os.pwrite(fd, row[done:], off + done)
	ts = [threading.Thread(target=work, args=(t,)) for t in range(n_threads)]
	t0 = time.perf_counter()
	for t in ts: t.start()
	for t in ts: t.join()
	dt = time.perf_counter() - t0
	for fd in fds: os.close(fd)
	for p in set(paths): os.remove(p)
	return n_rows * ROW / dt / 1e9


print("%s: %s; %.0f GB per run in 100-MiB pwrite()s" % (d, where(d), n_rows * ROW / 1e9), flush=True)
for n in (1, 2, 4, 8):
	if direct:
		try:
			print("%d thread(s), O_DIRECT after posix_fallocate: one file %.1f GB/s, a file each %.1f GB/s" % (n, run(n, True, True), run(n, False, True)), flush=True)
		except OSError as e:
			print("O_DIRECT: %s" % e, flush=True)
			break
	else:
		print("%d thread(s): one file %.1f GB/s, a file each %.1f GB/s" % (n, run(n, True), run(n, False)), flush=True)
