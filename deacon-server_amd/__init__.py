"""MI355X-native read-filtering core for Deacon: HIP kernels behind a C ABI (include/deacon_hip.h) plus a thin
host mirror of the reference's filter interface.  The directory name has a hyphen, so import it through the
`deacon_server_amd` shim at the repository root (or importlib.import_module("deacon-server_amd"))."""
from . import _native, client, distributed, server
from ._native import DeaconHipError, build, declared_symbols
from .filter import (DEFAULT_KMER_LENGTH, DEFAULT_WINDOW_SIZE, FilterProcessor, Index, PendingBatch, PinnedBuffer,
                     concat_reads, get_minimizer_hashes_and_positions, pack_ascii, paired_should_keep,
                     get_minimizer_variant, set_minimizer_variant, stats_allreduce, unpaired_should_keep)

__all__ = ["DeaconHipError", "build", "declared_symbols", "Index", "FilterProcessor", "PinnedBuffer", "PendingBatch", "concat_reads",
           "pack_ascii", "stats_allreduce", "set_minimizer_variant", "get_minimizer_variant",
           "get_minimizer_hashes_and_positions", "unpaired_should_keep", "paired_should_keep",
           "DEFAULT_KMER_LENGTH", "DEFAULT_WINDOW_SIZE"]
