"""One GPU context = one v2m_ctx (include/v2m_hip.h): graph resident in HBM, kernels on its stream."""

import ctypes as C

import numpy as np

from . import _native as N


class V2MError(RuntimeError):
	def __init__(self, code, message):
		super().__init__("%s: %s" % (N.ERROR_NAMES.get(code, "V2M_ERR_%d" % code), message))
		self.code = code


class RowBatch:
	"""v2m_row_batch: per row either a constant chromosome copy (PLOIDY_MAX = REF row) or a list of
	(cut node, copy) pairs (the founder delegate, founder_sequence_greedy_output.cc:78-115)."""

	def __init__(self, rows):
		copy_index, cut_offsets, cut_nodes, cut_copies = [], [0], [], []
		any_cuts = False
		for r in rows:
			if isinstance(r, (int, np.integer)):
				copy_index.append(int(r))
			else:
				any_cuts = True
				copy_index.append(N.V2M_PLOIDY_MAX)
				for node, copy in r:
					cut_nodes.append(int(node))
					cut_copies.append(int(copy))
			cut_offsets.append(len(cut_nodes))
		self.n_rows = len(copy_index)
		self.copy_index = np.ascontiguousarray(copy_index, dtype=np.uint32)
		self.cut_offsets = np.ascontiguousarray(cut_offsets, dtype=np.uint64) if any_cuts else None
		self.cut_nodes = np.ascontiguousarray(cut_nodes, dtype=np.uint64)
		self.cut_copies = np.ascontiguousarray(cut_copies, dtype=np.uint32)
		self.struct = N.RowBatchStruct(
			self.n_rows,
			self.copy_index.ctypes.data if self.n_rows else None,
			self.cut_offsets.ctypes.data if any_cuts else None,
			self.cut_nodes.ctypes.data if len(cut_nodes) else None,
			self.cut_copies.ctypes.data if len(cut_copies) else None,
		)

	@classmethod
	def haplotypes(cls, copies, include_reference=False):
		rows = ([N.V2M_PLOIDY_MAX] if include_reference else []) + [int(c) for c in copies]
		return cls(rows)


class Context:
	def __init__(self, device=0):
		self._lib = N.load()
		h = C.c_void_p()
		rc = self._lib.v2m_ctx_create(device, C.byref(h))
		if rc != N.V2M_OK:
			raise V2MError(rc, self._lib.v2m_last_error(None).decode())
		self._h = h
		self._keepalive = None

	def close(self):
		if getattr(self, "_h", None):
			self._lib.v2m_ctx_destroy(self._h)
			self._h = None

	def __del__(self):
		try:
			self.close()
		except Exception:
			pass

	def __enter__(self):
		return self

	def __exit__(self, *exc):
		self.close()

	def _check(self, rc):
		if rc != N.V2M_OK:
			raise V2MError(rc, self._lib.v2m_last_error(self._h).decode())

	@property
	def stream(self):
		return self._lib.v2m_ctx_stream(self._h)

	@property
	def info(self):
		return self._lib.v2m_ctx_info(self._h).decode()

	def synchronize(self):
		self._check(self._lib.v2m_ctx_synchronize(self._h))

	# ---- transpose_matrix (transpose_matrix.hh:14) --------------------------------------------
	def transpose_matrix(self, words, n_rows, n_cols):
		"""Host form: column-major u64 words of an n_rows x n_cols bit matrix -> words of its transpose."""
		src = np.ascontiguousarray(words, dtype=np.uint64)
		dst = np.zeros(src.size, dtype=np.uint64)
		self._check(self._lib.v2m_transpose_bits(self._h, src.ctypes.data if src.size else None, n_rows, n_cols, dst.ctypes.data if dst.size else None))
		return dst

	def transpose_bits_device(self, d_src, n_rows, n_cols, d_dst):
		self._check(self._lib.v2m_transpose_bits_device(self._h, d_src, n_rows, n_cols, d_dst))

	# ---- graph ---------------------------------------------------------------------------------
	def upload_graph(self, graph, ref_seq):
		ref = bytes(ref_seq)
		lb = graph.label_bytes
		view = N.GraphView(
			graph.node_count, graph.edge_count,
			graph.reference_positions.ctypes.data, graph.aligned_positions.ctypes.data,
			graph.alt_edge_targets.ctypes.data if graph.edge_count else None,
			graph.alt_edge_count_csum.ctypes.data,
			graph.label_offsets.ctypes.data if graph.edge_count else None,
			C.cast(C.c_char_p(lb), C.c_void_p) if lb else None,
			graph.paths_by_chrom_copy_and_edge.ctypes.data if graph.paths_by_chrom_copy_and_edge is not None and graph.paths_by_chrom_copy_and_edge.size else None,
			graph.path_rows, graph.path_cols,
		)
		self._check(self._lib.v2m_upload_graph(self._h, C.byref(view), C.cast(C.c_char_p(ref), C.c_void_p) if ref else None, len(ref)))

	def set_paths_device(self, d_words, path_rows, path_cols):
		self._check(self._lib.v2m_set_paths_device(self._h, d_words, path_rows, path_cols))

	def bind_path_matrix_device(self, d_paths_by_edge_and_chrom_copy, n_rows, n_cols):
		"""v2m_bind_path_matrix_device: transposes a device-resident transpose input (n_rows copies x n_cols edges) into the
		context's own, line-aligned path matrix and binds it.  Asynchronous."""
		self._check(self._lib.v2m_bind_path_matrix_device(self._h, d_paths_by_edge_and_chrom_copy, n_rows, n_cols))

	def upload_path_slice(self, paths_by_edge_and_chrom_copy, n_rows, n_cols, first_copy=0, n_copies=None):
		"""v2m_upload_path_slice: this GPU's chromosome copies [first_copy, first_copy + n_copies) out of the whole host-resident
		transpose input (n_rows copies x n_cols edges), transposed on the GPU and bound as the uploaded graph's path matrix."""
		words = np.ascontiguousarray(paths_by_edge_and_chrom_copy, dtype=np.uint64)
		assert words.size == n_rows // 64 * n_cols
		n_copies = n_rows - first_copy if n_copies is None else n_copies
		self._check(self._lib.v2m_upload_path_slice(self._h, words.ctypes.data if words.size else None, n_rows, n_cols, first_copy, n_copies))

	def upload_path_blocks(self, paths_by_edge_and_chrom_copy, n_rows, n_cols, first_copy, block_copies, stride_copies, copy_end=None):
		"""v2m_upload_path_blocks: every stride_copies-th block of block_copies chromosome copies from first_copy on (up to copy_end):
		the share of one of several GPUs whose rows have to leave in row order.  Local copy l = global copy
		first_copy + (l // block_copies) * stride_copies + l % block_copies."""
		words = np.ascontiguousarray(paths_by_edge_and_chrom_copy, dtype=np.uint64)
		assert words.size == n_rows // 64 * n_cols
		self._check(self._lib.v2m_upload_path_blocks(self._h, words.ctypes.data if words.size else None, n_rows, n_cols, first_copy, block_copies, stride_copies,
			n_rows if copy_end is None else copy_end))

	@property
	def aligned_length(self):
		return self._lib.v2m_aligned_length(self._h)

	@property
	def min_row_pitch(self):
		return self._lib.v2m_min_row_pitch(self._h)

	@property
	def max_unaligned_length(self):
		return self._lib.v2m_max_unaligned_length(self._h)

	# ---- rows ----------------------------------------------------------------------------------
	def splice_rows(self, rows, sink=None, unaligned=False):
		"""v2m_splice_rows.  sink(row_index, body: bytes) is called per row in order; without a sink the
		bodies are collected and returned as a list."""
		if not isinstance(rows, RowBatch):
			rows = RowBatch(rows)
		collected = []
		error = []

		def _cb(_user, row_index, ptr, length):
			try:
				body = C.string_at(ptr, length) if length else b""
				if sink is None:
					collected.append(body)
				else:
					sink(row_index, body)
				return 0
			except BaseException as e:  # propagate through the C frame as V2M_ERR_SINK
				error.append(e)
				return 1

		cb = N.SINK_FN(_cb)
		rc = self._lib.v2m_splice_rows(self._h, C.byref(rows.struct), N.V2M_SPLICE_UNALIGNED if unaligned else 0, cb, None)
		if error:
			raise error[0]
		self._check(rc)
		return collected if sink is None else None

	def splice_rows_held(self, rows, on_row, n_slots=4, unaligned=False):
		"""v2m_splice_rows_held: on_row(row_index, address, length, hold) is called per row in order on this thread and may return
		before it is done with the row; the bytes at `address` stay valid until release_row(hold) is called (any thread, once per
		accepted row).  A callback that raises, or returns a true value, refuses the row and ends the call."""
		if not isinstance(rows, RowBatch):
			rows = RowBatch(rows)
		error = []

		def _cb(_user, row_index, ptr, length, hold):
			try:
				return 1 if on_row(row_index, ptr, length, hold) else 0
			except BaseException as e:
				error.append(e)
				return 1

		cb = N.HOLD_SINK_FN(_cb)
		rc = self._lib.v2m_splice_rows_held(self._h, C.byref(rows.struct), N.V2M_SPLICE_UNALIGNED if unaligned else 0, n_slots, cb, None)
		if error:
			raise error[0]
		self._check(rc)

	def release_row(self, hold):
		self._lib.v2m_row_release(hold)

	def splice_rows_device(self, rows, d_out, row_pitch, unaligned=False, want_lengths=False):
		if not isinstance(rows, RowBatch):
			rows = RowBatch(rows)
		lengths = np.zeros(rows.n_rows, dtype=np.uint64) if want_lengths else None
		self._check(self._lib.v2m_splice_rows_device(self._h, C.byref(rows.struct), N.V2M_SPLICE_UNALIGNED if unaligned else 0,
			d_out, row_pitch, lengths.ctypes.data if want_lengths and rows.n_rows else None))
		return lengths

	def alloc_output(self, nbytes, candidates=3):
		"""v2m_alloc_output: device memory for row output, picked among `candidates` allocations by measured write rate."""
		p = C.c_void_p()
		self._check(self._lib.v2m_alloc_output(self._h, nbytes, candidates, C.byref(p)))
		return p.value

	def free_output(self, ptr):
		self._check(self._lib.v2m_free_output(self._h, ptr))

	def checksum_rows_device(self, d_rows, row_pitch, n_rows, length=0, lengths=None):
		out = np.zeros(n_rows, dtype=np.uint64)
		la = None if lengths is None else np.ascontiguousarray(lengths, dtype=np.uint64)
		self._check(self._lib.v2m_checksum_rows_device(self._h, d_rows, row_pitch, n_rows, length, la.ctypes.data if la is not None else None, out.ctypes.data if n_rows else None))
		return out

	# ---- profiling -----------------------------------------------------------------------------
	def profile_enable(self, enabled=True):
		self._check(self._lib.v2m_profile_enable(self._h, int(enabled)))

	def profile_reset(self):
		self._check(self._lib.v2m_profile_reset(self._h))

	def profile_get(self, kernel):
		n, ms = C.c_uint64(), C.c_double()
		self._check(self._lib.v2m_profile_get(self._h, kernel, C.byref(n), C.byref(ms)))
		return n.value, ms.value


def _profile_launches(self, kernel):
	"""Per-launch device times (ms) of `kernel` since the last profile_reset()."""
	n = C.c_uint64()
	self._check(self._lib.v2m_profile_get_launches(self._h, kernel, None, 0, C.byref(n)))
	out = np.zeros(n.value, dtype=np.float64)
	if n.value:
		self._check(self._lib.v2m_profile_get_launches(self._h, kernel, out.ctypes.data, n.value, C.byref(n)))
	return out


Context.profile_launches = _profile_launches


def checksum_rows_host(rows_bytes):
	"""The checksum of v2m_checksum_rows_device computed with numpy (for comparisons in tests/bench)."""
	out = []
	golden = np.uint64(0x9E3779B97F4A7C15)
	for body in rows_bytes:
		n = len(body)
		pad = (-n) % 8
		words = np.frombuffer(bytes(body) + b"\0" * pad, dtype="<u8")
		idx = np.arange(1, words.size + 1, dtype=np.uint64)
		with np.errstate(over="ignore"):
			acc = _mix64(idx * golden ^ words).sum(dtype=np.uint64) if words.size else np.uint64(0)
			acc = acc + _mix64(np.array([n], dtype=np.uint64))[0]
		out.append(int(acc))
	return np.array(out, dtype=np.uint64)


def _mix64(z):
	z = z.astype(np.uint64, copy=True)
	with np.errstate(over="ignore"):
		z ^= z >> np.uint64(30)
		z *= np.uint64(0xBF58476D1CE4E5B9)
		z ^= z >> np.uint64(27)
		z *= np.uint64(0x94D049BB133111EB)
		z ^= z >> np.uint64(31)
	return z
