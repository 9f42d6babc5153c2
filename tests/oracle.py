"""ctypes binding of the CPU oracle (oracle/libv2m_oracle.so).  TEST INFRASTRUCTURE ONLY.

The oracle is the checker: tests compare the HIP path with it; the product never
loads it (see oracle/v2m_oracle.cc header).
"""

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB_PATH = os.path.join(ORACLE_DIR, "libv2m_oracle.so")

PLOIDY_MAX = 0xFFFFFFFF

_u64p = C.POINTER(C.c_uint64)
_u32p = C.POINTER(C.c_uint32)


def build_oracle(force=False):
	src = os.path.join(ORACLE_DIR, "v2m_oracle.cc")
	if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
		# never start a compiler from a process that runs under a profiler's preload (vcf2multialign_amd/build.py:under_profiler)
		if "rocprof" in os.environ.get("LD_PRELOAD", "").lower() or any(k.startswith(("ROCPROF", "ROCPROFILER_")) for k in os.environ):
			raise RuntimeError("oracle/libv2m_oracle.so is missing or stale and this process runs under a profiler: run `make -C oracle` first, without the profiler")
		subprocess.check_call(["make", "-C", ORACLE_DIR, "-B" if force else "-s"])
	return _LIB_PATH


_lib = None


def lib():
	global _lib
	if _lib is not None:
		return _lib
	L = C.CDLL(build_oracle())
	L.v2mo_build_variant_graph.restype = C.c_void_p
	L.v2mo_build_variant_graph.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p), _u64p, C.c_char_p, C.c_size_t]
	L.v2mo_build_variant_graph_ex.restype = C.c_void_p
	L.v2mo_build_variant_graph_ex.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_void_p), _u64p, C.c_char_p, C.c_size_t]
	L.v2mo_free.argtypes = [C.c_void_p]
	L.v2mo_graph_from_arrays.restype = C.c_void_p
	L.v2mo_graph_from_arrays.argtypes = [C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p,
		C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_char_p, C.c_void_p]
	L.v2mo_graph_free.argtypes = [C.c_void_p]
	for name in ("node_count", "edge_count", "sample_count", "sample_name_blob_size", "handled_variants", "chr_id_mismatches", "overlap_count"):
		f = getattr(L, "v2mo_" + name)
		f.restype = C.c_uint64
		f.argtypes = [C.c_void_p]
	for name in ("reference_positions", "aligned_positions", "alt_edge_targets", "alt_edge_count_csum", "label_offsets"):
		f = getattr(L, "v2mo_" + name)
		f.restype = _u64p
		f.argtypes = [C.c_void_p]
	L.v2mo_label_bytes.restype = C.c_void_p
	L.v2mo_label_bytes.argtypes = [C.c_void_p]
	L.v2mo_sample_name_blob.restype = C.c_void_p
	L.v2mo_sample_name_blob.argtypes = [C.c_void_p]
	L.v2mo_ploidy_csum.restype = _u32p
	L.v2mo_ploidy_csum.argtypes = [C.c_void_p]
	L.v2mo_path_words.restype = _u64p
	L.v2mo_path_words.argtypes = [C.c_void_p, C.c_int, _u64p, _u64p]
	L.v2mo_overlap_get.argtypes = [C.c_void_p, C.c_uint64, _u64p, _u64p, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), _u32p, _u32p]
	for name in ("v2mo_transpose_matrix", "v2mo_transpose_matrix_naive"):
		f = getattr(L, name)
		f.restype = C.c_int
		f.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
	L.v2mo_output_sequence.restype = C.c_int64
	L.v2mo_output_sequence.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
	L.v2mo_row_checksum.restype = C.c_uint64
	L.v2mo_row_checksum.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64, _u64p]
	L.v2mo_haplotype_output_a2m.restype = C.c_int64
	L.v2mo_haplotype_output_a2m.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_char_p, C.POINTER(C.c_double)]
	L.v2mo_founder_output_a2m.restype = C.c_int64
	L.v2mo_founder_output_a2m.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
	L.v2mo_find_founders.restype = C.c_uint64
	L.v2mo_find_founders.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, _u32p]
	_lib = L
	return L


def _np_from(ptr, n, dtype):
	if n == 0:
		return np.zeros(0, dtype=dtype)
	return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)


class OracleGraph:
	"""An oracle-side variant graph (variant_graph.hh:57-66) with numpy copies of its arrays."""

	def __init__(self, handle, ref=None):
		L = lib()
		self._h = handle
		self.ref = ref  # bytes or None
		N = L.v2mo_node_count(handle)
		E = L.v2mo_edge_count(handle)
		S = L.v2mo_sample_count(handle)
		self.reference_positions = _np_from(L.v2mo_reference_positions(handle), N, np.uint64)
		self.aligned_positions = _np_from(L.v2mo_aligned_positions(handle), N, np.uint64)
		self.alt_edge_targets = _np_from(L.v2mo_alt_edge_targets(handle), E, np.uint64)
		self.alt_edge_count_csum = _np_from(L.v2mo_alt_edge_count_csum(handle), N + 1, np.uint64)
		self.label_offsets = _np_from(L.v2mo_label_offsets(handle), E + 1, np.uint64)
		nbytes = int(self.label_offsets[-1]) if E else 0
		self.label_bytes = C.string_at(L.v2mo_label_bytes(handle), nbytes) if nbytes else b""
		blob = C.string_at(L.v2mo_sample_name_blob(handle), L.v2mo_sample_name_blob_size(handle))
		self.sample_names = [s.decode() for s in blob.split(b"\0")[:-1]] if S else []
		# no record on the requested chromosome leaves ploidy_csum empty in the reference too (variant_graph.cc:215-221 never runs)
		pc = L.v2mo_ploidy_csum(handle)
		self.ploidy_csum = _np_from(pc, S + 1, np.uint32) if (S and pc) else np.zeros(1, np.uint32)
		r, c = C.c_uint64(), C.c_uint64()
		p = L.v2mo_path_words(handle, 0, C.byref(r), C.byref(c))
		self.path_rows, self.path_cols = r.value, c.value  # rows = edges (Ep), cols = copies (Hp)
		self.paths_by_chrom_copy_and_edge = _np_from(p, r.value * c.value // 64, np.uint64)
		p = L.v2mo_path_words(handle, 1, C.byref(r), C.byref(c))
		self.paths_by_edge_and_chrom_copy_dims = (r.value, c.value)  # rows = copies (Hp), cols = edges (Ep)
		self.paths_by_edge_and_chrom_copy = _np_from(p, r.value * c.value // 64, np.uint64)

	def __del__(self):
		try:
			if self._h:
				lib().v2mo_graph_free(self._h)
				self._h = None
		except Exception:
			pass

	@property
	def node_count(self):
		return len(self.reference_positions)

	@property
	def edge_count(self):
		return len(self.alt_edge_targets)

	@property
	def aligned_length(self):
		return int(self.aligned_positions[-1])

	@property
	def total_chromosome_copies(self):
		return int(self.ploidy_csum[-1])

	def labels(self):
		o = self.label_offsets
		return [self.label_bytes[int(o[i]):int(o[i + 1])].decode() for i in range(self.edge_count)]

	def overlaps(self):
		L = lib()
		out = []
		for i in range(L.v2mo_overlap_count(self._h)):
			ln, rp, vid, smp, ci, gt = C.c_uint64(), C.c_uint64(), C.c_char_p(), C.c_char_p(), C.c_uint32(), C.c_uint32()
			L.v2mo_overlap_get(self._h, i, C.byref(ln), C.byref(rp), C.byref(vid), C.byref(smp), C.byref(ci), C.byref(gt))
			out.append({"lineno": ln.value, "ref_pos": rp.value, "var_id": vid.value.decode(), "sample": smp.value.decode(), "chrom_copy_idx": ci.value, "gt": gt.value})
		return out

	# -- output_sequence (sequence_writer.cc:22-85) ---------------------------------------------
	def output_sequence(self, ref, copy_index=PLOIDY_MAX, cuts=None, fasta_id=None, unaligned=False):
		"""One row.  cuts = list of (cut_node, copy_index) pairs selects the founder delegate."""
		L = lib()
		ref = bytes(ref)
		cap = 2 * (self.aligned_length + len(ref)) + 4096 + (len(fasta_id) if fasta_id else 0)
		buf = C.create_string_buffer(cap)
		if cuts:
			cn = np.ascontiguousarray([c[0] for c in cuts], dtype=np.uint64)
			cc = np.ascontiguousarray([c[1] for c in cuts], dtype=np.uint32)
			n = L.v2mo_output_sequence(self._h, ref, fasta_id.encode() if fasta_id else None, int(unaligned), 0, cn.ctypes.data, cc.ctypes.data, len(cuts), buf, cap)
		else:
			n = L.v2mo_output_sequence(self._h, ref, fasta_id.encode() if fasta_id else None, int(unaligned), copy_index, None, None, 0, buf, cap)
		assert 0 <= n <= cap
		return buf.raw[:n]

	def row_checksum(self, ref, copy_index=PLOIDY_MAX, cuts=None, unaligned=False):
		"""(checksum, length) of the row output_sequence() writes: v2m_checksum_rows_device's formula computed inside the
		oracle, so whole 100-250 MB rows can be compared without coming through Python.  `ref` must be bytes (not copied)."""
		L = lib()
		n = C.c_uint64()
		if cuts:
			cn = np.ascontiguousarray([c[0] for c in cuts], dtype=np.uint64)
			cc = np.ascontiguousarray([c[1] for c in cuts], dtype=np.uint32)
			s = L.v2mo_row_checksum(self._h, ref, int(unaligned), 0, cn.ctypes.data, cc.ctypes.data, len(cuts), C.byref(n))
		else:
			s = L.v2mo_row_checksum(self._h, ref, int(unaligned), copy_index, None, None, 0, C.byref(n))
		return int(s), int(n.value)

	def row_checksums(self, ref, rows, unaligned=False, threads=8):
		"""row_checksum() of many rows (each an int copy index, PLOIDY_MAX, or a list of (cut node, copy) pairs) on a few
		threads (the C call releases the GIL and only reads the graph).  Returns (checksums u64 array, lengths u64 array)."""
		from concurrent.futures import ThreadPoolExecutor
		ref = bytes(ref)
		def one(r):
			return self.row_checksum(ref, cuts=r, unaligned=unaligned) if isinstance(r, (list, tuple)) else self.row_checksum(ref, copy_index=int(r), unaligned=unaligned)
		with ThreadPoolExecutor(max_workers=max(1, min(threads, len(rows) or 1))) as ex:
			res = list(ex.map(one, rows))
		return np.array([s for s, _ in res], dtype=np.uint64), np.array([n for _, n in res], dtype=np.uint64)

	# -- haplotype_output::output_a2m (haplotype_output.cc:38-82) ---------------------------------
	def haplotype_output_a2m(self, ref, path=None, chromosome_id=None, output_reference=True, unaligned=False, first_copy=0, n_copies=None):
		"""Writes to `path`, or discards (timed baseline) when path is None.  Returns (bytes, seconds)."""
		L = lib()
		if n_copies is None:
			n_copies = self.total_chromosome_copies
		secs = C.c_double()
		n = L.v2mo_haplotype_output_a2m(self._h, bytes(ref), chromosome_id.encode() if chromosome_id else None, int(output_reference), int(unaligned),
			first_copy, n_copies, path.encode() if path else None, C.byref(secs))
		if n < 0:
			raise OSError("oracle could not write " + str(path))
		return n, secs.value

	# -- founder_sequence_greedy_output::output_a2m (founder_sequence_greedy_output.cc:515-550) ---
	def founder_output_a2m(self, ref, cut_positions, assigned_samples_column_major, n_founders, chromosome_id=None, output_reference=True, unaligned=False):
		L = lib()
		ref = bytes(ref)
		cuts = np.ascontiguousarray(cut_positions, dtype=np.uint64)
		asg = np.ascontiguousarray(assigned_samples_column_major, dtype=np.uint32)
		cap = (n_founders + 1) * (2 * (self.aligned_length + len(ref)) + 4096)
		buf = C.create_string_buffer(cap)
		n = L.v2mo_founder_output_a2m(self._h, ref, chromosome_id.encode() if chromosome_id else None, int(output_reference), int(unaligned),
			cuts.ctypes.data, len(cuts), asg.ctypes.data, n_founders, buf, cap)
		assert 0 <= n <= cap
		return buf.raw[:n]

	# -- find_cut_positions + find_matchings (find_cut_positions.cc:93-211, founder_sequence_greedy_output.cc:154-512) ---
	def find_founders(self, founder_count, min_distance=0, keep_ref_edges=False):
		"""Literal restatement (every pBWT divergence-count update made).  Returns (cut_positions,
		assigned_samples column-major, score) or None when there is no solution."""
		n = self.node_count
		cuts = np.zeros(max(1, n), dtype=np.uint64)
		assigned = np.zeros(max(1, n * founder_count), dtype=np.uint32)
		score = C.c_uint32()
		k = lib().v2mo_find_founders(self._h, min_distance, founder_count, int(keep_ref_edges), cuts.ctypes.data, assigned.ctypes.data, assigned.size, C.byref(score))
		if k == 0:
			return None
		return cuts[:k].tolist(), assigned[:(k - 1) * founder_count].tolist(), score.value


def build_variant_graph(fasta_path, vcf_path, chr_id, seq_id=None, exclude_sample=None, exclude_copy=-1):
	"""build_variant_graph (variant_graph.cc:108-454) on FASTA + VCF files.  Returns OracleGraph with .ref set.
	exclude_sample / exclude_copy restate the delegate's should_include() (all copies of the sample when exclude_copy < 0)."""
	L = lib()
	err = C.create_string_buffer(512)
	refp, reflen = C.c_void_p(), C.c_uint64()
	h = L.v2mo_build_variant_graph_ex(str(fasta_path).encode(), seq_id.encode() if seq_id else None, str(vcf_path).encode(), chr_id.encode(),
		exclude_sample.encode() if exclude_sample else None, exclude_copy, C.byref(refp), C.byref(reflen), err, len(err))
	if not h:
		raise ValueError(err.value.decode())
	ref = C.string_at(refp, reflen.value)
	L.v2mo_free(refp)
	return OracleGraph(h, ref)


def graph_from_arrays(reference_positions, aligned_positions, alt_edge_targets, alt_edge_count_csum, label_offsets, label_bytes,
		path_words, path_rows, path_cols, sample_names=(), ploidy_csum=None):
	L = lib()
	a = [np.ascontiguousarray(x, dtype=np.uint64) for x in (reference_positions, aligned_positions, alt_edge_targets, alt_edge_count_csum, label_offsets)]
	pw = np.ascontiguousarray(path_words, dtype=np.uint64)
	blob = b"".join(s.encode() + b"\0" for s in sample_names)
	pc = np.ascontiguousarray(ploidy_csum, dtype=np.uint32) if ploidy_csum is not None else None
	lb = bytes(label_bytes) + b"\0"
	h = L.v2mo_graph_from_arrays(len(a[0]), len(a[2]), a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data, a[3].ctypes.data, a[4].ctypes.data, lb,
		pw.ctypes.data if pw.size else None, path_rows, path_cols, len(sample_names), blob + b"\0", pc.ctypes.data if pc is not None else None)
	return OracleGraph(h)


def transpose_matrix(words, rows, cols, naive=False):
	"""transpose_matrix (transpose_matrix.cc:41-109): column-major u64 words, dims % 64 == 0."""
	L = lib()
	src = np.ascontiguousarray(words, dtype=np.uint64)
	assert src.size == rows * cols // 64
	dst = np.zeros(src.size, dtype=np.uint64)
	fn = L.v2mo_transpose_matrix_naive if naive else L.v2mo_transpose_matrix
	rc = fn(src.ctypes.data if src.size else None, rows, cols, dst.ctypes.data if dst.size else None)
	if rc != 0:
		raise ValueError("matrix dimensions must be multiples of 64")
	return dst
