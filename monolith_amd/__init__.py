"""monolith_amd — MI355X-native drop-in for Monolith's sparse-embedding hot path.

Python mirrors of monolith/native_training/{hash_table_ops,hash_filter_ops,distribution_ops,
embedding_combiners,multi_type_hash_table,distributed_ps_sync}.py over the C ABI of
libmonolith_hip.so (include/monolith_hip.h). torch tensors are only buffer carriers.
"""
from monolith_amd import _lib  # noqa: F401  (fails loudly when the HIP library is missing)

__all__ = ["_lib"]
