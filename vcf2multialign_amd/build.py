"""Builds the native pieces in-tree (the .so files travel to the GPU box with the snapshot).

  libv2m_hip.so          -- the product: HIP kernels + C ABI (include/v2m_hip.h), gfx950 only.
  libv2m_hip_tuning.so   -- the same sources with -DV2M_TUNING_BUILD: every transpose shape / flavour that was measured on
                            the way to the three the product ships.  Loaded only by tools/tune_transpose.py and
                            tests/test_gpu_tuning_build.py (V2M_HIP_LIBRARY); nothing in the product path uses it.

hipcc cross-compiles for gfx950 without a GPU present.
"""

import contextlib
import fcntl
import os
import re
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libv2m_hip.so")

_QUOTED_INCLUDE = re.compile(r'^[ \t]*#[ \t]*include[ \t]*"([^"]+)"', re.M)


class MissingInclude(str):
	"""A quoted #include that resolves to no file: kept in the dependency list as a path that does not exist, which makes the
	target stale (the compiler then says what is wrong) without taking the other targets down with it."""


def include_closure(sources, strict=False):
	"""The sources plus every file they reach through `#include "..."`, transitively: what a target has to be rebuilt for.
	Read off the files themselves so that a new header cannot be forgotten (round 3 shipped a library older than
	founder_kernels.hpp because a hand-kept list did not name it).  An include that resolves to no file (one that relies on -I, or
	sits inside `#if 0`) is returned as a MissingInclude entry -- or raised with strict=True, which the CPU test suite uses."""
	seen, todo = [], [os.path.normpath(s) for s in sources]
	while todo:
		path = todo.pop()
		if path in seen:
			continue
		seen.append(path)
		if isinstance(path, MissingInclude):
			continue
		with open(path, errors="replace") as f:
			text = f.read()
		for inc in _QUOTED_INCLUDE.findall(text):
			dep = os.path.normpath(os.path.join(os.path.dirname(path), inc))
			if not os.path.exists(dep):
				if strict:
					raise RuntimeError('%s includes "%s", which does not exist' % (path, inc))
				dep = MissingInclude(dep)
			todo.append(dep)
	return sorted(seen)


HIP_SOURCES = [os.path.join(CSRC, "v2m_hip.hip")]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wextra"]
TUNING_LIB_PATH = os.path.join(PKG_DIR, "libv2m_hip_tuning.so")

# synthetic-input generator (bench + scale tests): host generator + the HIP kernel filling genotype bits
SYNTH_LIB_PATH = os.path.join(PKG_DIR, "libv2m_synth.so")
SYNTH_SOURCES = [os.path.join(CSRC, "synth", "synth_capi.hip"), os.path.join(CSRC, "synth", "synth.cc"), os.path.join(CSRC, "synth", "sink.cc"), os.path.join(CSRC, "host", "graph_builder.cc")]

# C++ host: graph builder, readers, output classes (libv2m_host.so) and the command-line driver
HOST_LIB_PATH = os.path.join(PKG_DIR, "libv2m_host.so")
HOST_DIR = os.path.join(CSRC, "host")
HOST_SOURCES = [os.path.join(HOST_DIR, f) for f in ("graph_builder.cc", "readers.cc", "gpu_path.cc", "output.cc", "founder.cc", "graph_file.cc", "host_capi.cc")]
CLI_PATH = os.path.join(PKG_DIR, "bin", "vcf2multialign")
CLI_SOURCES = [os.path.join(HOST_DIR, "main.cc")]
CXX_FLAGS = ["-O2", "-g", "-std=c++20", "-Wall", "-Wextra"]

# HIP_DEPS, SYNTH_DEPS, HOST_DEPS, CLI_DEPS: the include closures, computed on first use (module __getattr__ below) -- importing this
# module reads no source file, and a broken include in one target does not stop the others from building.
_SOURCES_OF = {"HIP_DEPS": HIP_SOURCES, "SYNTH_DEPS": SYNTH_SOURCES, "HOST_DEPS": HOST_SOURCES, "CLI_DEPS": CLI_SOURCES}
_closures = {}


def deps(name):
	if name not in _closures:
		_closures[name] = include_closure(_SOURCES_OF[name])
	return _closures[name]


def __getattr__(name):
	if name in _SOURCES_OF:
		return deps(name)
	raise AttributeError(name)


def under_profiler():
	"""True when a profiler's preload is in this process's environment (rocprofv3 sets LD_PRELOAD to its tool library and ROCPROF* /
	ROCPROFILER_* variables).  With --pmc that library has initialised the GPU before main(), and a compiler launcher started from
	here (hipcc, g++, make -> sh -> g++) would exec the real compiler with the preload inherited: the exec-after-GPU-init hop that
	takes a node of this pool down.  Nothing is ever compiled in that state."""
	if "rocprof" in os.environ.get("LD_PRELOAD", "").lower():
		return True
	return any(k.startswith(("ROCPROF", "ROCPROFILER_")) for k in os.environ)


class StaleUnderProfiler(RuntimeError):
	pass


def refuse_to_compile_under_profiler(target):
	if under_profiler():
		raise StaleUnderProfiler("%s is missing or older than its sources and this process runs under a profiler (LD_PRELOAD / ROCPROF* set): "
			"build first WITHOUT the profiler -- python3 -c 'import __graft_entry__ as g; g.build()' -- then profile" % os.path.relpath(target, ROOT))


def find_hipcc():
	for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
		if cand and os.path.exists(cand):
			return cand
	return None


def _stale(target, dep_list):
	if not os.path.exists(target):
		return True
	t = os.path.getmtime(target)
	return any(isinstance(d, MissingInclude) or (os.path.exists(d) and os.path.getmtime(d) > t) for d in dep_list)


def _link(cmd_prefix, target, cmd_suffix, verbose):
	"""Runs a compile-and-link command into a temporary name and renames it over the target, so that a process that
	loads the library while another one rebuilds it never sees a half-written file."""
	tmp = "%s.tmp%d" % (target, os.getpid())
	cmd = cmd_prefix + ["-o", tmp] + cmd_suffix
	if verbose:
		print(" ".join(cmd))
	try:
		subprocess.check_call(cmd, cwd=ROOT)
		os.replace(tmp, target)
	finally:
		with contextlib.suppress(OSError):
			os.unlink(tmp)


@contextlib.contextmanager
def _build_lock():
	"""One builder at a time per checkout (several ranks or pytest workers may start together)."""
	with open(os.path.join(PKG_DIR, ".build.lock"), "w") as f:
		fcntl.flock(f, fcntl.LOCK_EX)
		try:
			yield
		finally:
			fcntl.flock(f, fcntl.LOCK_UN)


def _build_host(force, verbose):
	cxx = os.environ.get("CXX") or shutil.which("g++")
	if cxx is None:
		raise RuntimeError("g++ not found")
	if force or _stale(HOST_LIB_PATH, deps("HOST_DEPS") + [LIB_PATH]):
		refuse_to_compile_under_profiler(HOST_LIB_PATH)
		_link([cxx] + CXX_FLAGS + ["-fPIC", "-shared"], HOST_LIB_PATH, HOST_SOURCES + ["-L" + PKG_DIR, "-lv2m_hip", "-Wl,-rpath,$ORIGIN"], verbose)
	if force or _stale(CLI_PATH, deps("CLI_DEPS") + deps("HOST_DEPS") + [HOST_LIB_PATH]):
		refuse_to_compile_under_profiler(CLI_PATH)
		os.makedirs(os.path.dirname(CLI_PATH), exist_ok=True)
		_link([cxx] + CXX_FLAGS, CLI_PATH, CLI_SOURCES + ["-L" + PKG_DIR, "-lv2m_host", "-lv2m_hip", "-Wl,-rpath,$ORIGIN/.."], verbose)


def _build(target, sources, dep_list, force, verbose, extra_flags=()):
	if not force and not _stale(target, dep_list):
		return target
	refuse_to_compile_under_profiler(target)
	hipcc = find_hipcc()
	if hipcc is None:
		raise RuntimeError("hipcc not found: cannot build " + target)
	_link([hipcc] + HIPCC_FLAGS + list(extra_flags), target, sources, verbose)
	return target


def build_native(force=False, verbose=False):
	"""Compiles libv2m_hip.so (and the synthetic-input helper) for gfx950 if missing or older than the sources.  Raises
	StaleUnderProfiler instead of compiling when a profiler's preload is in the environment (see under_profiler)."""
	with _build_lock():
		_build(SYNTH_LIB_PATH, SYNTH_SOURCES, deps("SYNTH_DEPS"), force, verbose)
		_build(LIB_PATH, HIP_SOURCES, deps("HIP_DEPS"), force, verbose)
		_build(TUNING_LIB_PATH, HIP_SOURCES, deps("HIP_DEPS"), force, verbose, ["-DV2M_TUNING_BUILD"])
		_build_host(force, verbose)
	return LIB_PATH
