#!/usr/bin/env python3
"""Where the time of a one-launch query goes (diagnostic, on the GPU box).  Needs a library built with
-DMCQ_DIRECT_STAMPS (MCQ_LIBRARY=...): block 0's waves write 100 MHz timestamps at the stages of
mcq_eval_direct_kernel; this prints, per stage, the median over launches of (stamp - kernel entry) for wave 0 and
for the wave that is last at that stage."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import neuron_poker_amd as npa  # noqa: E402
from neuron_poker_amd import _lib  # noqa: E402

STAGES = ["entry (table image leaves)", "round-0 generator ready", "tables + work in LDS", "base deck", "(unused)",
          "generator taken", "iterations done", "wave sums (tally.add)", "partials in LDS", "rows stored", "block barrier",
          "system fence", "flag written"]


def main():
    eng = npa.Engine(0, kernel_times=True)
    lib = _lib.load_library()
    lib.mcq_debug_read_stamps.argtypes = [C.c_void_p]
    cases = [("1 x 1 run, heads-up preflop", npa.pack_queries([[50, 46]], [[255] * 5], 2, 1)),
             ("1 x 1000 runs, heads-up preflop", npa.pack_queries([[50, 46]], [[255] * 5], 2, 1000)),
             ("1 x 1000 runs, 6 players preflop", npa.pack_queries([[50, 46]], [[255] * 5], 6, 1000)),
             ("1 x 1000 runs, 4 players flop", npa.pack_queries([[50, 46]], [[0, 13, 30, 255, 255]], 4, 1000))]
    from small_probe import mix
    cases.append(("1024 x 1000 runs (mix), block 0", mix(1024)[0]))
    cases.append(("256 x 1000 runs (mix), block 0", mix(256)[0]))
    for name, q in cases:
        rows, kms = [], []
        for i in range(300):
            eng.eval_batch(q, seed=i)
            kms.append(eng.last_kernel_ms)
            st = np.zeros((16, 16), np.uint64)
            assert lib.mcq_debug_read_stamps(st.ctypes.data) == 0
            rows.append(st.astype(np.int64))
        st = np.stack(rows[50:])                      # [launch, wave, stage]
        t0 = st[:, :, 0].min(axis=1)[:, None, None]
        rel = (st - t0) / 100.0                        # us since the first wave's entry
        print("%s: kernel %.1f us between its timestamps" % (name, np.median(kms[50:]) * 1e3))
        for k, s in enumerate(STAGES):
            w0 = np.median(rel[:, 0, k])
            last = np.median(rel[:, :, k].max(axis=1))
            print("   %-26s wave 0 %6.2f us   last wave %6.2f us" % (s, w0, last))


if __name__ == "__main__":
    main()
