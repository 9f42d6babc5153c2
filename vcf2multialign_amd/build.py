"""Builds the native pieces in-tree (the .so files travel to the GPU box with the snapshot).

  libv2m_hip.so   -- the product: HIP kernels + C ABI (include/v2m_hip.h), gfx950 only.

hipcc cross-compiles for gfx950 without a GPU present.
"""

import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libv2m_hip.so")

HIP_SOURCES = [os.path.join(CSRC, "v2m_hip.hip")]
HIP_DEPS = HIP_SOURCES + [os.path.join(CSRC, "kernels.hpp"), os.path.join(ROOT, "include", "v2m_hip.h")]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wextra"]

# synthetic-input generator (bench + scale tests): host generator + the HIP kernel filling genotype bits
SYNTH_LIB_PATH = os.path.join(PKG_DIR, "libv2m_synth.so")
SYNTH_SOURCES = [os.path.join(CSRC, "synth", "synth_capi.hip"), os.path.join(CSRC, "synth", "synth.cc"), os.path.join(CSRC, "host", "graph_builder.cc")]
SYNTH_DEPS = SYNTH_SOURCES + [os.path.join(CSRC, "synth", "synth.hh"), os.path.join(CSRC, "host", "graph_builder.hh"), os.path.join(CSRC, "host", "variant_graph.hh")]


def find_hipcc():
	for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
		if cand and os.path.exists(cand):
			return cand
	return None


def _stale(target, deps):
	if not os.path.exists(target):
		return True
	t = os.path.getmtime(target)
	return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


# C++ host: graph builder, readers, output classes (libv2m_host.so) and the command-line driver
HOST_LIB_PATH = os.path.join(PKG_DIR, "libv2m_host.so")
HOST_DIR = os.path.join(CSRC, "host")
HOST_SOURCES = [os.path.join(HOST_DIR, f) for f in ("graph_builder.cc", "readers.cc", "gpu_path.cc", "output.cc", "founder.cc", "graph_file.cc", "host_capi.cc")]
HOST_DEPS = HOST_SOURCES + [os.path.join(HOST_DIR, f) for f in ("graph_builder.hh", "readers.hh", "gpu_path.hh", "output.hh", "founder.hh", "graph_file.hh", "variant_graph.hh")] + [os.path.join(ROOT, "include", "v2m_hip.h")]
CLI_PATH = os.path.join(PKG_DIR, "bin", "vcf2multialign")
CLI_SOURCES = [os.path.join(HOST_DIR, "main.cc")]
CXX_FLAGS = ["-O2", "-g", "-std=c++20", "-Wall", "-Wextra"]


def _build_host(force, verbose):
	cxx = os.environ.get("CXX") or shutil.which("g++")
	if cxx is None:
		raise RuntimeError("g++ not found")
	if force or _stale(HOST_LIB_PATH, HOST_DEPS + [LIB_PATH]):
		cmd = [cxx] + CXX_FLAGS + ["-fPIC", "-shared", "-o", HOST_LIB_PATH] + HOST_SOURCES + ["-L" + PKG_DIR, "-lv2m_hip", "-Wl,-rpath,$ORIGIN"]
		if verbose:
			print(" ".join(cmd))
		subprocess.check_call(cmd, cwd=ROOT)
	if force or _stale(CLI_PATH, CLI_SOURCES + HOST_DEPS + [HOST_LIB_PATH]):
		os.makedirs(os.path.dirname(CLI_PATH), exist_ok=True)
		cmd = [cxx] + CXX_FLAGS + ["-o", CLI_PATH] + CLI_SOURCES + ["-L" + PKG_DIR, "-lv2m_host", "-lv2m_hip", "-Wl,-rpath,$ORIGIN/.."]
		if verbose:
			print(" ".join(cmd))
		subprocess.check_call(cmd, cwd=ROOT)


def _build(target, sources, deps, force, verbose):
	if not force and not _stale(target, deps):
		return target
	hipcc = find_hipcc()
	if hipcc is None:
		raise RuntimeError("hipcc not found: cannot build " + target)
	cmd = [hipcc] + HIPCC_FLAGS + ["-o", target] + sources
	if verbose:
		print(" ".join(cmd))
	subprocess.check_call(cmd, cwd=ROOT)
	return target


def build_native(force=False, verbose=False):
	"""Compiles libv2m_hip.so (and the synthetic-input helper) for gfx950 if missing or older than the sources."""
	_build(SYNTH_LIB_PATH, SYNTH_SOURCES, SYNTH_DEPS, force, verbose)
	_build(LIB_PATH, HIP_SOURCES, HIP_DEPS, force, verbose)
	_build_host(force, verbose)
	return LIB_PATH
