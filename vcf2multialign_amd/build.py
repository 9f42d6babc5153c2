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
	return _build(LIB_PATH, HIP_SOURCES, HIP_DEPS, force, verbose)
