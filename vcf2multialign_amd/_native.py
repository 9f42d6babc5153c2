"""ctypes binding of the C ABI declared in include/v2m_hip.h.

The HIP library is the only implementation: if libv2m_hip.so is missing this module raises,
it never substitutes a CPU path.
"""

import ctypes as C
import os

from . import build as _build

_u64p = C.POINTER(C.c_uint64)

V2M_OK = 0
ERROR_NAMES = {
	1: "V2M_ERR_INVALID_ARGUMENT", 2: "V2M_ERR_PRECONDITION", 3: "V2M_ERR_UNSUPPORTED", 4: "V2M_ERR_NO_DEVICE",
	5: "V2M_ERR_HIP", 6: "V2M_ERR_OUT_OF_MEMORY", 7: "V2M_ERR_SINK", 8: "V2M_ERR_STATE",
}
V2M_ERR_INVALID_ARGUMENT, V2M_ERR_PRECONDITION, V2M_ERR_UNSUPPORTED, V2M_ERR_NO_DEVICE = 1, 2, 3, 4
V2M_ERR_HIP, V2M_ERR_OUT_OF_MEMORY, V2M_ERR_SINK, V2M_ERR_STATE = 5, 6, 7, 8
V2M_PLOIDY_MAX = 0xFFFFFFFF
V2M_SPLICE_UNALIGNED = 0x1

KERNEL_TRANSPOSE, KERNEL_RESOLVE, KERNEL_SPLICE_ALIGNED, KERNEL_SPLICE_UNALIGNED, KERNEL_TEMPLATE, KERNEL_UNALIGNED_COUNT = range(6)
KERNEL_NAMES = ["transpose_bits_kernel", "resolve_effective_edges_kernel", "splice_aligned_kernel", "splice_unaligned_kernel", "expand_reference_row_kernel", "count_unaligned_kernel"]
ABI_VERSION = 5


class GraphView(C.Structure):
	_fields_ = [
		("node_count", C.c_uint64), ("edge_count", C.c_uint64),
		("reference_positions", C.c_void_p), ("aligned_positions", C.c_void_p),
		("alt_edge_targets", C.c_void_p), ("alt_edge_count_csum", C.c_void_p),
		("alt_edge_label_offsets", C.c_void_p), ("alt_edge_label_bytes", C.c_void_p),
		("paths_by_chrom_copy_and_edge", C.c_void_p), ("path_rows", C.c_uint64), ("path_cols", C.c_uint64),
	]


class RowBatchStruct(C.Structure):
	_fields_ = [
		("n_rows", C.c_uint64), ("copy_index", C.c_void_p), ("cut_offsets", C.c_void_p),
		("cut_nodes", C.c_void_p), ("cut_copies", C.c_void_p),
	]


SINK_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64)
HOLD_SINK_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p)   # v2m_hold_sink_fn
TRIALS_SINK_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64)   # v2m_trials_sink

# every symbol include/v2m_hip.h declares: (restype, argtypes)
SIGNATURES = {
	"v2m_abi_version": (C.c_uint32, []),
	"v2m_ctx_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
	"v2m_ctx_destroy": (None, [C.c_void_p]),
	"v2m_last_error": (C.c_char_p, [C.c_void_p]),
	"v2m_ctx_synchronize": (C.c_int, [C.c_void_p]),
	"v2m_ctx_stream": (C.c_void_p, [C.c_void_p]),
	"v2m_ctx_info": (C.c_char_p, [C.c_void_p]),
	"v2m_transpose_bits": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]),
	"v2m_transpose_bits_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]),
	"v2m_upload_graph": (C.c_int, [C.c_void_p, C.POINTER(GraphView), C.c_void_p, C.c_uint64]),
	"v2m_set_paths_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64]),
	"v2m_upload_path_slice": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64]),
	"v2m_upload_path_blocks": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64]),
	"v2m_bind_path_matrix_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64]),
	"v2m_aligned_length": (C.c_uint64, [C.c_void_p]),
	"v2m_min_row_pitch": (C.c_uint64, [C.c_void_p]),
	"v2m_max_unaligned_length": (C.c_uint64, [C.c_void_p]),
	"v2m_splice_rows": (C.c_int, [C.c_void_p, C.POINTER(RowBatchStruct), C.c_uint32, SINK_FN, C.c_void_p]),
	"v2m_splice_rows_held": (C.c_int, [C.c_void_p, C.POINTER(RowBatchStruct), C.c_uint32, C.c_uint32, HOLD_SINK_FN, C.c_void_p]),
	"v2m_row_release": (None, [C.c_void_p]),
	"v2m_splice_rows_device": (C.c_int, [C.c_void_p, C.POINTER(RowBatchStruct), C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p]),
	"v2m_pbwt_cut_trials": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p,
		C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
	"v2m_pbwt_cut_trials_streamed": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p,
		C.c_uint64, C.c_void_p, C.c_void_p, TRIALS_SINK_FN, C.c_void_p]),
	"v2m_pbwt_cut_records": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
		C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
	"v2m_alloc_output": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int, C.POINTER(C.c_void_p)]),
	"v2m_free_output": (C.c_int, [C.c_void_p, C.c_void_p]),
	"v2m_checksum_rows_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]),
	"v2m_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
	"v2m_profile_reset": (C.c_int, [C.c_void_p]),
	"v2m_profile_get": (C.c_int, [C.c_void_p, C.c_int, _u64p, C.POINTER(C.c_double)]),
	"v2m_profile_get_launches": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, _u64p]),
}

_lib = None


def library_path():
	"""The product library; V2M_HIP_LIBRARY names another build of the same sources (the tuning build, tools and tests only)."""
	return os.environ.get("V2M_HIP_LIBRARY") or _build.LIB_PATH


def load():
	"""Loads libv2m_hip.so.  Raises ImportError if it has not been built -- there is no fallback."""
	global _lib
	if _lib is not None:
		return _lib
	path = library_path()
	if not os.path.exists(path):
		raise ImportError(
			"%s is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()'). "
			"vcf2multialign_amd has no CPU fallback." % path)
	# One HIP runtime per process: torch ships its own libamdhip64 (SONAME libamdhip64.so.7, the same as
	# /opt/rocm's).  Importing torch first makes our NEEDED entry resolve to the copy torch already
	# loaded; the other order would load two runtimes and the second one finds no GPU.
	try:
		import torch  # noqa: F401
	except ImportError:
		pass
	lib = C.CDLL(path)
	for name, (restype, argtypes) in SIGNATURES.items():
		fn = getattr(lib, name)   # AttributeError if the library does not export a declared symbol
		fn.restype = restype
		fn.argtypes = argtypes
	_lib = lib
	return lib


def hip_runtime():
	"""The HIP runtime this process already has loaded (torch's copy, or /opt/rocm's through libv2m_hip.so), as a ctypes handle --
	found by its mapping, not by a versioned soname.  For device <-> host copies in tools, tests and bench.py."""
	load()
	with open("/proc/self/maps") as f:
		for line in f:
			path = line.split()[-1]
			if "libamdhip64.so" in os.path.basename(path):
				return C.CDLL(path)
	raise ImportError("no libamdhip64 is loaded in this process")
