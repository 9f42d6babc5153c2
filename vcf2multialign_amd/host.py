"""Binding of the C++ host library's graph builder (libv2m_host.so: csrc/host/readers.cc + graph_builder.cc),
used by the tests to compare it with the oracle's builder.  The transpose is not part of it (GPU)."""

import ctypes as C
import os

import numpy as np

from . import build as _build

_u64p = C.POINTER(C.c_uint64)
_lib = None


def _load():
	global _lib
	if _lib is None:
		if not os.path.exists(_build.HOST_LIB_PATH):
			raise ImportError(_build.HOST_LIB_PATH + " is missing: run __graft_entry__.build()")
		try:
			import torch  # noqa: F401  (one HIP runtime per process, see _native.load)
		except ImportError:
			pass
		L = C.CDLL(_build.HOST_LIB_PATH)
		L.v2mh_build_variant_graph.restype = C.c_void_p
		L.v2mh_build_variant_graph.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_uint, C.c_char_p, C.c_size_t]
		L.v2mh_free.argtypes = [C.c_void_p]
		L.v2mh_graph_from_arrays.restype = C.c_void_p
		L.v2mh_graph_from_arrays.argtypes = [C.c_uint64, C.c_uint64] + [C.c_void_p] * 5 + [C.c_char_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32]
		L.v2mh_write_graph.restype = C.c_int
		L.v2mh_write_graph.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_size_t]
		L.v2mh_read_graph.restype = C.c_void_p
		L.v2mh_read_graph.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
		for n in ("node_count", "edge_count", "sample_count", "ref_length", "sample_blob_size", "ploidy_csum_size", "handled_variants", "chr_id_mismatches", "overlap_count"):
			f = getattr(L, "v2mh_" + n)
			f.restype = C.c_uint64
			f.argtypes = [C.c_void_p]
		for n in ("reference_positions", "aligned_positions", "alt_edge_targets", "alt_edge_count_csum", "label_offsets"):
			f = getattr(L, "v2mh_" + n)
			f.restype = _u64p
			f.argtypes = [C.c_void_p]
		for n in ("reference", "label_bytes", "sample_blob", "ploidy_csum"):
			f = getattr(L, "v2mh_" + n)
			f.restype = C.c_void_p
			f.argtypes = [C.c_void_p]
		L.v2mh_paths_by_edge_and_chrom_copy.restype = _u64p
		L.v2mh_paths_by_edge_and_chrom_copy.argtypes = [C.c_void_p, _u64p, _u64p]
		L.v2mh_find_founders.restype = C.c_uint64
		L.v2mh_find_founders.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint32)]
		L.v2mh_find_founders_mt.restype = C.c_uint64
		L.v2mh_find_founders_mt.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint32), C.c_uint]
		L.v2mh_set_paths_by_chrom_copy_and_edge.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64]
		L.v2mh_overlap_get.argtypes = [C.c_void_p, C.c_uint64, _u64p, _u64p, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
		L.v2mh_shard_copies.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, _u64p, _u64p]
		L.v2mh_write_cut_positions.restype = C.c_int
		L.v2mh_write_cut_positions.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_char_p, C.c_size_t]
		L.v2mh_read_cut_positions.restype = C.c_int
		L.v2mh_read_cut_positions.argtypes = [C.c_char_p, C.c_void_p, _u64p, _u64p, C.POINTER(C.c_uint32), C.c_char_p, C.c_size_t]
		_lib = L
	return _lib


def shard_copies(n_copies, world, rank):
	"""The C++ driver's copy range of GPU `rank` of `world` (csrc/host/gpu_path.cc: shard_copies)."""
	a, b = C.c_uint64(), C.c_uint64()
	_load().v2mh_shard_copies(n_copies, world, rank, C.byref(a), C.byref(b))
	return a.value, b.value


def write_cut_positions(path, cut_positions, min_distance, score):
	"""--output-cut-positions file (layout: vcf2multialign_amd/csrc/host/founder.hh)."""
	cuts = np.ascontiguousarray(cut_positions, dtype=np.uint64)
	err = C.create_string_buffer(512)
	if 0 != _load().v2mh_write_cut_positions(str(path).encode(), cuts.ctypes.data, len(cuts), min_distance, score, err, len(err)):
		raise OSError(err.value.decode())


def read_cut_positions(path):
	"""Returns (cut_positions, min_distance, score)."""
	L = _load()
	err = C.create_string_buffer(512)
	n, md, sc = C.c_uint64(0), C.c_uint64(), C.c_uint32()
	if 0 != L.v2mh_read_cut_positions(str(path).encode(), None, C.byref(n), C.byref(md), C.byref(sc), err, len(err)):
		raise ValueError(err.value.decode())
	cuts = np.zeros(max(1, n.value), dtype=np.uint64)
	if 0 != L.v2mh_read_cut_positions(str(path).encode(), cuts.ctypes.data, C.byref(n), C.byref(md), C.byref(sc), err, len(err)):
		raise ValueError(err.value.decode())
	return cuts[:n.value].tolist(), md.value, sc.value


def _arr(ptr, n, dtype):
	if n == 0 or not ptr:
		return np.zeros(0, dtype=dtype)
	return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)


class HostGraph:
	"""Result of the host's build_variant_graph (without the final transpose)."""

	def __init__(self, fasta_path, vcf_path, chr_id, seq_id=None, exclude_sample=None, exclude_copy=-1, threads=0, _graph_file=None, _handle=None):
		L = _load()
		err = C.create_string_buffer(512)
		if _handle is not None:
			h = _handle
		elif _graph_file is not None:
			h = L.v2mh_read_graph(str(_graph_file).encode(), err, len(err))
		else:
			h = L.v2mh_build_variant_graph(str(fasta_path).encode(), seq_id.encode() if seq_id else None, str(vcf_path).encode(), chr_id.encode(),
				exclude_sample.encode() if exclude_sample else None, exclude_copy, threads, err, len(err))
		if not h:
			raise ValueError(err.value.decode())
		self._h = h
		if True:
			N, E, S = L.v2mh_node_count(h), L.v2mh_edge_count(h), L.v2mh_sample_count(h)
			self.ref = C.string_at(L.v2mh_reference(h), L.v2mh_ref_length(h))
			self.reference_positions = _arr(L.v2mh_reference_positions(h), N, np.uint64)
			self.aligned_positions = _arr(L.v2mh_aligned_positions(h), N, np.uint64)
			self.alt_edge_targets = _arr(L.v2mh_alt_edge_targets(h), E, np.uint64)
			self.alt_edge_count_csum = _arr(L.v2mh_alt_edge_count_csum(h), N + 1, np.uint64)
			self.label_offsets = _arr(L.v2mh_label_offsets(h), E + 1, np.uint64)
			self.label_bytes = C.string_at(L.v2mh_label_bytes(h), int(self.label_offsets[-1])) if E else b""
			blob = C.string_at(L.v2mh_sample_blob(h), L.v2mh_sample_blob_size(h))
			self.sample_names = [s.decode() for s in blob.split(b"\0")[:-1]] if S else []
			n_pc = L.v2mh_ploidy_csum_size(h)
			self.ploidy_csum = np.ctypeslib.as_array(C.cast(L.v2mh_ploidy_csum(h), C.POINTER(C.c_uint32)), shape=(n_pc,)).copy() if n_pc else np.zeros(1, np.uint32)
			r, c = C.c_uint64(), C.c_uint64()
			p = L.v2mh_paths_by_edge_and_chrom_copy(h, C.byref(r), C.byref(c))
			self.paths_by_edge_and_chrom_copy_dims = (r.value, c.value)
			self.paths_by_edge_and_chrom_copy = _arr(p, r.value * c.value // 64, np.uint64)
			self.handled_variants = L.v2mh_handled_variants(h)
			self.chr_id_mismatches = L.v2mh_chr_id_mismatches(h)
			self.overlaps = []
			for i in range(L.v2mh_overlap_count(h)):
				ln, rp, vid, smp, ci, gt = C.c_uint64(), C.c_uint64(), C.c_char_p(), C.c_char_p(), C.c_uint32(), C.c_uint32()
				L.v2mh_overlap_get(h, i, C.byref(ln), C.byref(rp), C.byref(vid), C.byref(smp), C.byref(ci), C.byref(gt))
				self.overlaps.append({"lineno": ln.value, "ref_pos": rp.value, "var_id": vid.value.decode(), "sample": smp.value.decode(), "chrom_copy_idx": ci.value, "gt": gt.value})

	def __del__(self):
		try:
			if getattr(self, "_h", None):
				_load().v2mh_free(self._h)
				self._h = None
		except Exception:
			pass

	@classmethod
	def from_arrays(cls, graph, paths_by_edge_and_chrom_copy, path_rows, path_cols, n_samples, ploidy):
		"""A host graph from a VariantGraph-like object plus the (copies x edges) path matrix words."""
		u64 = lambda x: np.ascontiguousarray(x, dtype=np.uint64)
		a = [u64(graph.reference_positions), u64(graph.aligned_positions), u64(graph.alt_edge_targets), u64(graph.alt_edge_count_csum), u64(graph.label_offsets)]
		pw = u64(paths_by_edge_and_chrom_copy)
		h = _load().v2mh_graph_from_arrays(len(a[0]), len(a[2]), a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data, a[3].ctypes.data, a[4].ctypes.data,
			bytes(graph.label_bytes) + b"\0", pw.ctypes.data if pw.size else None, path_rows, path_cols, n_samples, ploidy)
		return cls(None, None, None, _handle=h)

	@classmethod
	def read(cls, path):
		"""read_graph(): a graph from this build's flat V2MGRAF1 file (no reference sequence)."""
		return cls(None, None, None, _graph_file=path)

	def write(self, path):
		err = C.create_string_buffer(512)
		if 0 != _load().v2mh_write_graph(self._h, str(path).encode(), err, len(err)):
			raise OSError(err.value.decode())

	def set_transposed_paths(self, words, rows, cols):
		"""paths_by_chrom_copy_and_edge (rows = edges, cols = copies) as produced by the transpose: needed by the
		multi-threaded founder search when the graph was built without a GPU context."""
		w = np.ascontiguousarray(words, dtype=np.uint64)
		assert w.size == rows // 64 * cols
		_load().v2mh_set_paths_by_chrom_copy_and_edge(self._h, w.ctypes.data, rows, cols)

	def find_cut_positions_gpu(self, ctx, min_distance=0, threads=0):
		"""find_cut_positions with the chunk walks on the GPU (v2m_pbwt_cut_trials): `ctx` is a vcf2multialign_amd.Context that holds
		this graph WITH its transposed path matrix.  Returns (cut_positions, score) or None."""
		L = _load()
		L.v2mh_find_cut_positions_gpu.restype = C.c_uint64
		L.v2mh_find_cut_positions_gpu.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint, C.c_void_p, C.POINTER(C.c_uint32), C.c_void_p, C.c_char_p, C.c_size_t]
		n = len(self.reference_positions)
		cuts = np.zeros(n, dtype=np.uint64)
		score = C.c_uint32()
		err = C.create_string_buffer(512)
		chunks = (C.c_uint64 * 2)()
		k = L.v2mh_find_cut_positions_gpu(self._h, ctx._h, min_distance, threads, cuts.ctypes.data, C.byref(score), chunks, err, len(err))
		if err.value:
			raise RuntimeError(err.value.decode())
		self.gpu_chunks_walked, self.gpu_chunks_left = int(chunks[0]), int(chunks[1])   # by the GPU / by the host after all
		if k == 0:
			return None
		return cuts[:k].tolist(), score.value

	def find_founders_gpu(self, ctx, founder_count, min_distance=0, keep_ref_edges=False, threads=0, cut_positions=None):
		"""find_cut_positions + find_matchings with the chunk walks of both on the GPU (v2m_pbwt_cut_trials, v2m_pbwt_cut_records);
		cut_positions: skip the search and match over these.  Returns (cut_positions, assigned_samples column-major, score) or None;
		self.gpu_chunks = (search: on the GPU, on the host; matching: on the GPU, on the host)."""
		L = _load()
		L.v2mh_find_founders_gpu.restype = C.c_uint64
		L.v2mh_find_founders_gpu.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_uint, C.c_int, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64,
			C.POINTER(C.c_uint32), C.c_void_p, C.c_char_p, C.c_size_t]
		n = len(self.reference_positions)
		cuts = np.zeros(n, dtype=np.uint64)
		n_in = 0
		if cut_positions is not None:
			n_in = len(cut_positions)
			cuts[:n_in] = cut_positions
		assigned = np.zeros(max(1, n * founder_count), dtype=np.uint32)
		score = C.c_uint32()
		chunks = (C.c_uint64 * 4)()
		err = C.create_string_buffer(512)
		k = L.v2mh_find_founders_gpu(self._h, ctx._h, min_distance, founder_count, int(keep_ref_edges), threads, int(cut_positions is None), n_in,
			cuts.ctypes.data, assigned.ctypes.data, assigned.size, C.byref(score), chunks, err, len(err))
		if err.value:
			raise RuntimeError(err.value.decode())
		self.gpu_chunks = tuple(int(x) for x in chunks)
		if k == 0:
			return None
		return cuts[:k].tolist(), assigned[:(k - 1) * founder_count].tolist(), score.value

	def find_founders_walked_on_host(self, founder_count, min_distance=0, keep_ref_edges=False, threads=2, max_copies=2 ** 63):
		"""The walked searches (chunks handed to a founder_walker) with the host's own edge-by-edge walker: what the GPU path runs around
		its kernels, testable without a GPU.  self.gpu_chunks = chunks (search: walked, handed back; matching: walked, handed back)."""
		L = _load()
		L.v2mh_find_founders_walked_on_host.restype = C.c_uint64
		L.v2mh_find_founders_walked_on_host.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_uint, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint32), C.c_void_p, C.c_char_p, C.c_size_t]
		n = len(self.reference_positions)
		cuts = np.zeros(n, dtype=np.uint64)
		assigned = np.zeros(max(1, n * founder_count), dtype=np.uint32)
		score = C.c_uint32()
		chunks = (C.c_uint64 * 4)()
		err = C.create_string_buffer(512)
		k = L.v2mh_find_founders_walked_on_host(self._h, min_distance, founder_count, int(keep_ref_edges), threads, max_copies, cuts.ctypes.data, assigned.ctypes.data, assigned.size, C.byref(score), chunks, err, len(err))
		self.gpu_chunks = tuple(int(x) for x in chunks)
		if k == 0:
			if err.value:
				raise RuntimeError(err.value.decode())
			return None
		return cuts[:k].tolist(), assigned[:(k - 1) * founder_count].tolist(), score.value

	def find_founders(self, founder_count, min_distance=0, keep_ref_edges=False, threads=1):
		"""find_cut_positions + find_matchings (host algorithms).  Returns (cut_positions, assigned_samples column-major, score)
		or None when there is no solution.  threads > 1 (0 = automatic) spreads the matching's pBWT over threads."""
		n = len(self.reference_positions)
		cuts = np.zeros(n, dtype=np.uint64)
		assigned = np.zeros(max(1, n * founder_count), dtype=np.uint32)
		score = C.c_uint32()
		k = _load().v2mh_find_founders_mt(self._h, min_distance, founder_count, int(keep_ref_edges), cuts.ctypes.data, assigned.ctypes.data, assigned.size, C.byref(score), threads)
		if k == 0:
			return None
		return cuts[:k].tolist(), assigned[:(k - 1) * founder_count].tolist(), score.value
