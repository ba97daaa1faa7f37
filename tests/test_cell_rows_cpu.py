"""The identity behind the cell tables of csrc/me_index.hip (k_level_hist_rows / k_block_counts / k_cell_fill), in numpy: a sorted point
starts a cell of Morton level L exactly when the highest level at which its code differs from its predecessor's is >= L, so a block's row
of the level histogram holds the block's cell count for every level, and the row sums are the occupied cells of every level."""
import numpy as np

CHUNK, LEVELS = 2048, 24


def _rows(codes):
    n = len(codes)
    x = codes[1:] ^ codes[:-1]
    lv = np.full(n, -1)
    nz = x != 0
    lv[1:][nz] = (np.floor(np.log2(x[nz].astype(np.float64))).astype(np.int64)) // 3  # (codes < 2^52: exact in float64)
    nb = (n + CHUNK - 1) // CHUNK
    rows = np.zeros((nb, LEVELS), dtype=np.int64)
    for b in range(nb):
        seg = lv[b * CHUNK:(b + 1) * CHUNK]
        seg = seg[seg >= 0]
        rows[b] = np.bincount(seg, minlength=LEVELS)[:LEVELS]
    return rows


def test_block_rows_give_the_cell_counts_of_every_level():
    rng = np.random.default_rng(5)
    # clustered 3-D lattice coordinates (17 bits per axis), Morton-interleaved, sorted, with duplicates
    xyz = (rng.normal(0, 1, (3 * CHUNK + 777, 3)) * 9000 + 65536).clip(0, 131071).astype(np.uint64)
    xyz = np.concatenate([xyz, xyz[:500]])
    codes = np.zeros(len(xyz), dtype=np.uint64)
    for bit in range(17):
        for a in range(3):
            codes |= ((xyz[:, a] >> np.uint64(bit)) & np.uint64(1)) << np.uint64(3 * bit + a)
    codes.sort()
    rows = _rows(codes)
    nb = len(rows)
    for level in (0, 1, 2, 5, 9, 16, 17):
        cells = codes >> np.uint64(3 * level)
        start = np.ones(len(codes), dtype=bool)
        start[1:] = cells[1:] != cells[:-1]
        direct = np.array([start[b * CHUNK:(b + 1) * CHUNK].sum() for b in range(nb)])
        from_rows = rows[:, level:].sum(1)
        from_rows[0] += 1  # the first point starts a cell of every level
        assert np.array_equal(direct, from_rows)
        assert from_rows.sum() == len(np.unique(cells))  # = level_unique[level] of cloud_build_index
