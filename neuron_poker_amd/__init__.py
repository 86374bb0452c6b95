"""neuron_poker_amd -- MI355X-native Monte-Carlo equity engine behind neuron_poker's equity call surface.

    from neuron_poker_amd import montecarlo_hip
    montecarlo_hip.get_equity({'AH', 'KH'}, set(), 2, 100000)          # tools.montecarlo_python.get_equity
    montecarlo_hip.MonteCarlo().run_montecarlo([['AH', 'KH']], [], 2, 1, maxRuns=10000, timeout=0, ghost_cards='')

The compute path is libmcq_hip.so (hand-written HIP for gfx950, C ABI in include/mcq.h) loaded through
ctypes; there is no CPU fallback -- importing works anywhere, evaluating needs the library and a GPU.
"""
from .cards import card_id, card_str, key_type, TYPES  # noqa: F401
from ._lib import (Engine, MODE_PHILOX, MODE_REPLAY_MT19937, QUERY_DTYPE, QUERY_EXT_DTYPE, RESULT_DTYPE, McqError, McqBusyError,  # noqa: F401
                   load_library, library_path, pack_queries, pack_query_ext, range_bits, class_bit, default_engine, Tables,
                   MultiEngine)

__all__ = ["card_id", "card_str", "key_type", "TYPES", "Engine", "MODE_PHILOX", "MODE_REPLAY_MT19937", "QUERY_DTYPE",
           "QUERY_EXT_DTYPE", "RESULT_DTYPE", "McqError", "McqBusyError", "load_library", "library_path", "pack_queries", "pack_query_ext",
           "range_bits", "class_bit", "default_engine", "Tables", "MultiEngine"]
