"""Synthetic inputs of the BASELINE.json shapes (bench.py and the scale tests): binding of libv2m_synth.so.

The reference + variant records are generated on the host and pushed through the product's graph_builder;
the genotype bit matrix is generated in HBM by fill_paths_kernel.  See csrc/synth/synth.hh for the recipe."""

import ctypes as C
import os

import numpy as np

from . import build as _build
from .variant_graph import VariantGraph

_u64p = C.POINTER(C.c_uint64)


class _Config(C.Structure):
	_fields_ = [("seed", C.c_uint64), ("ref_length", C.c_uint64), ("n_variants", C.c_uint64),
		("frac_mnp", C.c_double), ("frac_insertion", C.c_double), ("frac_deletion", C.c_double), ("frac_multiallelic", C.c_double),
		("max_indel", C.c_uint32)]


# BASELINE.md "Synthetic inputs": seeds and mixes per config
CONFIGS = {
	"config2": dict(seed=2, ref_length=10_000_000, n_variants=100_000, samples=1000),
	"config3": dict(seed=3, ref_length=100_000_000, n_variants=1_000_000, samples=2504, frac_insertion=0.10, frac_deletion=0.10),
	"config5": dict(seed=5, ref_length=250_000_000, n_variants=6_000_000, samples=10000, frac_mnp=0.05, frac_insertion=0.08, frac_deletion=0.08, frac_multiallelic=0.04),
	# a few-second stand-in with config 3's mix, for tests
	"mini3": dict(seed=33, ref_length=2_000_000, n_variants=20_000, samples=100, frac_insertion=0.10, frac_deletion=0.10),
	"mini5": dict(seed=55, ref_length=1_000_000, n_variants=24_000, samples=80, frac_mnp=0.05, frac_insertion=0.08, frac_deletion=0.08, frac_multiallelic=0.04),
}

_lib = None


def _load():
	global _lib
	if _lib is None:
		path = _build.SYNTH_LIB_PATH
		if not os.path.exists(path):
			raise ImportError(path + " is missing: run __graft_entry__.build()")
		try:
			import torch  # noqa: F401  (one HIP runtime per process, see _native.load)
		except ImportError:
			pass
		L = C.CDLL(path)
		L.v2ms_generate.restype = C.c_void_p
		L.v2ms_generate.argtypes = [C.POINTER(_Config)]
		L.v2ms_free.argtypes = [C.c_void_p]
		for n in ("node_count", "edge_count", "ref_length"):
			f = getattr(L, "v2ms_" + n)
			f.restype = C.c_uint64
			f.argtypes = [C.c_void_p]
		for n in ("reference_positions", "aligned_positions", "alt_edge_targets", "alt_edge_count_csum", "label_offsets"):
			f = getattr(L, "v2ms_" + n)
			f.restype = _u64p
			f.argtypes = [C.c_void_p]
		for n in ("reference", "label_bytes", "edge_thresholds"):
			f = getattr(L, "v2ms_" + n)
			f.restype = C.c_void_p
			f.argtypes = [C.c_void_p]
		L.v2ms_fill_paths_device.restype = C.c_int
		L.v2ms_fill_paths_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64]
		L.v2ms_write_fasta_and_vcf.restype = C.c_int
		L.v2ms_write_fasta_and_vcf.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_char_p, C.c_char_p, C.c_char_p]
		L.v2ms_copy_column.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64]
		_lib = L
	return _lib


def _arr(ptr, n, dtype):
	if n == 0:
		return np.zeros(0, dtype=dtype)
	return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)


class Dataset:
	"""reference bytes + VariantGraph (no path matrix) + per-edge thresholds, and the genotype hash."""

	def __init__(self, seed, ref_length, n_variants, samples, ploidy=2, frac_mnp=0.0, frac_insertion=0.0, frac_deletion=0.0, frac_multiallelic=0.0, max_indel=32):
		L = _load()
		cfg = _Config(seed, ref_length, n_variants, frac_mnp, frac_insertion, frac_deletion, frac_multiallelic, max_indel)
		self._h = L.v2ms_generate(C.byref(cfg))
		self.seed = seed
		self.n_variants = n_variants
		self.samples = samples
		self.ploidy = ploidy
		N, E = L.v2ms_node_count(self._h), L.v2ms_edge_count(self._h)
		self.reference = C.string_at(L.v2ms_reference(self._h), L.v2ms_ref_length(self._h))
		offs = _arr(L.v2ms_label_offsets(self._h), E + 1, np.uint64)
		self.n_copies = samples * ploidy                           # H
		self.path_cols = 64 * ((self.n_copies + 63) // 64)         # Hp
		self.path_rows = 64 * ((E + 63) // 64)                     # Ep
		self.graph = VariantGraph(
			_arr(L.v2ms_reference_positions(self._h), N, np.uint64), _arr(L.v2ms_aligned_positions(self._h), N, np.uint64),
			_arr(L.v2ms_alt_edge_targets(self._h), E, np.uint64), _arr(L.v2ms_alt_edge_count_csum(self._h), N + 1, np.uint64),
			offs, C.string_at(L.v2ms_label_bytes(self._h), int(offs[-1])) if E else b"",
			None, self.path_rows, self.path_cols,
			["S%d" % i for i in range(samples)], np.arange(samples + 1, dtype=np.uint32) * ploidy)
		self.edge_thresholds = np.ctypeslib.as_array(C.cast(L.v2ms_edge_thresholds(self._h), C.POINTER(C.c_uint32)), shape=(E,)).copy() if E else np.zeros(0, np.uint32)

	def __del__(self):
		try:
			if self._h:
				_load().v2ms_free(self._h)
				self._h = None
		except Exception:
			pass

	def fill_paths_device(self, stream, d_words, d_thresholds, copy_base=0, n_rows=None, n_cols=None, copy_end=None):
		"""Fills paths_by_edge_and_chrom_copy in HBM: rows = the chromosome copies [copy_base, copy_base + n_rows)
		(n_rows a multiple of 64, default all Hp), cols = n_cols edges (default Ep; columns past the last edge are zero).
		Rows of copies >= copy_end (default: the number of copies) are zero: a rank's padding."""
		n_rows = self.path_cols if n_rows is None else n_rows
		n_cols = self.path_rows if n_cols is None else n_cols
		copy_end = self.n_copies if copy_end is None else min(copy_end, self.n_copies)
		rc = _load().v2ms_fill_paths_device(stream, d_words, n_rows, n_cols, copy_base, copy_end, self.graph.edge_count, d_thresholds, self.seed)
		if rc != 0:
			raise RuntimeError("fill_paths_kernel launch failed (%d)" % rc)

	def write_fasta_and_vcf(self, fasta_path, vcf_path, chromosome="1"):
		"""The same dataset as FASTA + VCF text (genotypes from the same hash), for the text pipeline."""
		rc = _load().v2ms_write_fasta_and_vcf(self._h, self.seed, self.samples, self.ploidy, chromosome.encode(), str(fasta_path).encode(), str(vcf_path).encode())
		if rc != 0:
			raise OSError("could not write " + str(vcf_path))

	def copy_column(self, copy):
		"""CPU re-derivation of chromosome copy `copy`'s column of paths_by_chrom_copy_and_edge (Ep/64 words)."""
		out = np.zeros(self.path_rows // 64, dtype=np.uint64)
		_load().v2ms_copy_column(self._h, self.seed, copy, out.ctypes.data, out.size)
		return out


def dataset(name, **overrides):
	kw = dict(CONFIGS[name])
	kw.update(overrides)
	return Dataset(**kw)
