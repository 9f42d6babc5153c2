"""vcf2multialign_amd -- MI355X (gfx950) implementation of vcf2multialign's haplotype-splice hot path.

The compute lives in libv2m_hip.so (hand-written HIP kernels behind the C ABI of include/v2m_hip.h);
this package is the thin host-side binding used by the tests, bench.py and the Python driver.
There is no CPU implementation in here.
"""

from ._native import V2M_PLOIDY_MAX, load as load_library  # noqa: F401
from .context import Context, RowBatch, V2MError, checksum_rows_host  # noqa: F401
from .output import FounderSequenceGreedyOutput, HaplotypeOutput  # noqa: F401
from .variant_graph import PLOIDY_MAX, VariantGraph  # noqa: F401
