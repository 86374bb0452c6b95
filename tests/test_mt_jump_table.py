"""csrc/mcq_mt_jump_table.inc (tools/mt_jump_table.py): every row is the XOR combination that takes numpy's MT19937 stream
(g * 128 - 1) * 624 words ahead -- checked here against numpy's own generator stepped that far, on seeds the generating
script did not use, so a stale or hand-edited table fails the CPU suite."""
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "neuron_poker_amd", "csrc", "mcq_mt_jump_table.inc")
HPP = os.path.join(ROOT, "neuron_poker_amd", "csrc", "mcq_mt_blocks.hpp")


def _rows():
    text = open(INC).read()
    body = text[text.index("kMtJump"):]
    rows = re.findall(r"\{((?:0x[0-9a-f]+u,?)+)\}", body)
    return [np.array([int(v[:-1], 16) for v in r.split(",") if v], np.uint32) for r in rows]


def test_jump_table_rows_step_numpys_generator():
    hpp = open(HPP).read()
    seg = int(re.search(r"#define MCQ_MTB_SEG (\d+)u", hpp).group(1))
    n_seg = int(re.search(r"#define MCQ_MTB_MAX_SEG (\d+)u", hpp).group(1))
    rows = _rows()
    assert len(rows) == n_seg - 1 and all(len(r) == 624 for r in rows)
    need = ((n_seg - 1) * seg - 1) * 624 + 624
    for seed in (5, 2 ** 31 + 12345):
        w = np.random.RandomState(seed).randint(0, 2 ** 32, size=need, dtype=np.uint64).astype(np.uint32)
        for g, row in enumerate(rows, start=1):
            J = (g * seg - 1) * 624
            bits = np.unpackbits(row.view(np.uint8), bitorder="little")
            sel = np.nonzero(bits)[0]
            assert sel.max() < 19937
            for j in (0, 1, 226, 227, 397, 622, 623):
                assert np.bitwise_xor.reduce(w[sel + j]) == w[J + j], (seed, g, j)
