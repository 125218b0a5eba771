"""xapiand_amd — MI355X-native match/rank path for Xapiand (posting decode, AND/OR/PHRASE
matching, BM25, top-k, shard merge) behind the Xapian::Enquire surface.

The compute path is libxgm.so (hand-written HIP for gfx950, C ABI in include/xgm.h); this Python
package is the host-side mirror of the reference interface used by tests and benchmarks.
"""
from .enquire import (BM25Weight, Database, Enquire, MSet, MSetItem, Query, Unsupported, ValueCountMatchSpy, XgmError,  # noqa: F401
                      get_mset_sharded, merged_stats, plan, search_batch)

__all__ = ["BM25Weight", "Database", "Enquire", "MSet", "MSetItem", "Query", "Unsupported", "ValueCountMatchSpy", "XgmError",
           "get_mset_sharded", "merged_stats", "plan", "search_batch"]
