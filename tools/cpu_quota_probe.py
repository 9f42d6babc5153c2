#!/usr/bin/env python3
"""What a GPU box gives a job's host side: logical CPUs, the cgroup's CPU quota and throttling counters, and the rate of bench.py's checksumming
sink on plain host memory per thread count and loop flavour (scalar / AVX-512DQ).  Usage: python tools/cpu_quota_probe.py"""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from vcf2multialign_amd import build


def stat():
	try:
		d = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat").read().strip().splitlines())
		return "nr_throttled %s, throttled %.2f s" % (d.get("nr_throttled"), int(d.get("throttled_usec", 0)) / 1e6)
	except OSError:
		return "-"


print("logical CPUs %d, affinity %d, cgroup cpu.max: %s" % (os.cpu_count(), len(os.sched_getaffinity(0)), open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "-"))
print("loadavg", open("/proc/loadavg").read().strip())
sl = C.CDLL(build.SYNTH_LIB_PATH)
sl.v2ms_checksum_sink_create.restype = C.c_void_p
sl.v2ms_checksum_sink_create.argtypes = [C.c_uint64, C.c_uint32]
sl.v2ms_checksum_sink_fn.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
sl.v2ms_checksum_sink_destroy.argtypes = [C.c_void_p]
sl.v2ms_checksum_sink_flavour.restype = C.c_char_p
sl.v2ms_checksum_sink_flavour.argtypes = [C.c_void_p]
sl.v2ms_checksum_sink_force_scalar.argtypes = [C.c_void_p]
bufs = [np.full(100_000_000, 65 + i, dtype=np.uint8) for i in range(8)]
for scalar in (True, False):
	for t in (1, 2, 4, 8, 12, 16, 24):
		s = sl.v2ms_checksum_sink_create(4, t)
		if scalar:
			sl.v2ms_checksum_sink_force_scalar(s)
		flavour = sl.v2ms_checksum_sink_flavour(s).decode()
		t0 = time.perf_counter()
		for rep in range(4):
			for b in bufs:
				sl.v2ms_checksum_sink_fn(s, 0, b.ctypes.data, b.size)
		dt = time.perf_counter() - t0
		print("%-8s %2d threads: %6.1f GB/s   (cgroup: %s)" % (flavour, t, 3.2 / dt, stat()), flush=True)
		sl.v2ms_checksum_sink_destroy(s)
	if flavour == "scalar" and not scalar:
		break
