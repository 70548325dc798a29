"""Index logic of csrc/k_gram.h restated on the CPU (the kernels themselves are covered by the -m gpu parity tests): which
wavefront accumulates which tile of the Gram matrix, where a partial tile lands in memory and how k_gram_reduce reads it."""
import re
import os

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
SRC = open(os.path.join(ROOT, "open_vins_amd", "csrc", "k_gram.h")).read()


def _wave_rows(w):
    return [2 * w, 2 * w + 1, 14 - 2 * w, 15 - 2 * w]  # WaveRows<W>::R0..R3


def test_the_formulas_restated_here_are_the_ones_in_the_header():
    assert "static constexpr int R0 = 2 * W, R1 = 2 * W + 1, R2 = 14 - 2 * W, R3 = 15 - 2 * W;" in SRC
    assert "return ti * NT - (ti * (ti - 1)) / 2 + (tj - ti);" in SRC
    assert "const int q = 2 * (t >> 7) + (t & 1), lane = (t >> 1) & 63;" in SRC
    assert re.search(r"constexpr int GR_ACC = 34;", SRC) and re.search(r"constexpr int GR_LS = 272;", SRC)


def test_every_tile_of_the_upper_triangle_has_exactly_one_wavefront():
    rows = sorted(r for w in range(4) for r in _wave_rows(w))
    assert rows == list(range(16))
    for w in range(4):
        assert sum(16 - r for r in _wave_rows(w)) == 34  # GR_ACC accumulators per wavefront at the full grid
    for nt, loads in ((14, [27, 26, 26, 26]), (16, [34] * 4), (13, None), (1, None)):
        owner = {}
        for w in range(4):
            for r in _wave_rows(w):
                for j in range(r, 16):
                    if r < nt and j < nt:
                        assert (r, j) not in owner
                        owner[(r, j)] = w
        assert len(owner) == nt * (nt + 1) // 2
        if loads:
            assert [sum(1 for v in owner.values() if v == w) for w in range(4)] == loads


def test_partial_tile_addressing_is_a_bijection():
    for nt in (1, 6, 14, 16):
        idx = [ti * nt - (ti * (ti - 1)) // 2 + (tj - ti) for ti in range(nt) for tj in range(ti, nt)]
        assert sorted(idx) == list(range(nt * (nt + 1) // 2))
    # gram_put writes register pair h of lane l as one 16-byte store at slot h * 128 + 2 * l (+ e); k_gram_reduce thread t reads slot t
    slots = {}
    for h in range(2):
        for lane in range(64):
            for e in range(2):
                slots[h * 128 + 2 * lane + e] = (2 * h + e, lane)
    assert sorted(slots) == list(range(256))
    for t in range(256):
        assert slots[t] == (2 * (t >> 7) + (t & 1), (t >> 1) & 63)
    # accumulator layout of v_mfma_f64_16x16x4_f64: register q of lane l is element (4 q + (l >> 4), l & 15) of the tile
    elems = {(4 * q + (lane >> 4), lane & 15) for q in range(4) for lane in range(64)}
    assert len(elems) == 256


def test_stage_load_count_covers_a_stage_for_every_row_length():
    # NQ = 2 * NTC loads of 256 doubles must cover GR_ROWS = 32 rows of LD <= 16 NT doubles, NTC = NT rounded up to even
    for ld in range(2, 257):
        nt = (ld + 15) // 16
        ntc = 2 * ((nt + 1) // 2)
        assert 2 * ntc * 256 >= 32 * ld
        assert ld <= 272 - 1  # the scratch slot (column GR_LS - 1 of row 0) is never a data column
