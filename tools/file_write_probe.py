#!/usr/bin/env python3
"""What one output file allows: N threads pwrite()-ing 100-MB rows into ONE file at disjoint offsets, against N threads writing a file each
(buffered writes, tmpfs or disk).  Usage: python tools/file_write_probe.py DIR [total GB = 16]"""
import os, sys, threading, time
d = sys.argv[1]
total = int(float(sys.argv[2]) * 1e9) if len(sys.argv) > 2 else 16_000_000_000
row = bytes(bytearray(os.urandom(1 << 20)) * 100)          # 100 MiB
n_rows = total // len(row)


def run(n_threads, one_file):
	paths = [os.path.join(d, "probe_%d.bin" % (0 if one_file else t)) for t in range(n_threads)]
	fds = [os.open(p, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644) for p in (paths[:1] if one_file else paths)]
	def work(t):
		fd = fds[0 if one_file else t]
		for i in range(t, n_rows, n_threads):
			off = i * len(row) if one_file else (i // n_threads) * len(row)
			done = 0
			while done < len(row):
				done += os.pwrite(fd, memoryview(row)[done:], off + done)
	ts = [threading.Thread(target=work, args=(t,)) for t in range(n_threads)]
	t0 = time.perf_counter()
	for t in ts: t.start()
	for t in ts: t.join()
	dt = time.perf_counter() - t0
	for fd in fds: os.close(fd)
	for p in set(paths): os.remove(p)
	return n_rows * len(row) / dt / 1e9


for n in (1, 2, 4, 8):
	print("%d thread(s): one file %.1f GB/s, a file each %.1f GB/s" % (n, run(n, True), run(n, False)), flush=True)
