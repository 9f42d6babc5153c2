#!/usr/bin/env python3
"""Does the end-to-end rate depend on which NUMA node the process (and with it the library's pinned slots) lives on?
The 640 config-3 rows of tools/e2e_sink_sweep.py into the counting sink and the 16-thread checksumming sink, in a child process
per placement: unpinned, pinned to the GPU's NUMA node, pinned to the other one.  Usage: python tools/e2e_numa_probe.py"""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def cpus_of(node):
	out = set()
	for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
		a, _, b = part.partition("-")
		out.update(range(int(a), int(b or a) + 1))
	return out


if len(sys.argv) > 1 and sys.argv[1] == "child":
	cpus = sys.argv[2]
	if cpus != "any":
		os.sched_setaffinity(0, {int(c) for c in cpus.split(",")})
	sys.argv = [sys.argv[0], "--threads", "16"]
	exec(open(os.path.join(ROOT, "tools", "e2e_sink_sweep.py")).read())
	sys.exit(0)

nodes = sorted(int(p.rsplit("node", 1)[1]) for p in glob.glob("/sys/devices/system/node/node[0-9]*"))
gpu_nodes = {}
for p in glob.glob("/sys/class/drm/card*/device/numa_node"):
	try:
		gpu_nodes[p.split("/")[4]] = int(open(p).read())
	except (OSError, ValueError):
		pass
print("NUMA nodes %s; GPUs' nodes (sysfs): %s" % (nodes, gpu_nodes), flush=True)
for name, cpus in [("unpinned", "any")] + [("node %d" % n, ",".join(str(c) for c in sorted(cpus_of(n)))) for n in nodes]:
	r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", cpus], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
	lines = [l for l in r.stdout.decode().splitlines() if "sink" in l]
	print("== %s (exit %d)" % (name, r.returncode))
	print("\n".join(lines[-4:]), flush=True)
