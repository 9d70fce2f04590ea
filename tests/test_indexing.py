"""Known answers of the reference's own indexing tests (test/indexing.cpp, test/mat_seq.cpp), checked
against the oracle's indexer restatement.  These pin "bit-exact in indexing" (SURVEY.md 8c)."""
import numpy as np

from oracle import np_oracle as npo


def test_regular_indexer_kat():
    # test/indexing.cpp:6-39: mat_regular_indexer(3, 12, 3, 4); begin + 4 - 2 -> index 5
    idx = npo.RegularIndexer(3, 12, 3, 4)
    assert idx.index_begin() == 3 and idx.index_end() == 12
    t = idx.index_begin() + 4 - 2
    assert t == 5 and idx.rows(t) == 3 and idx.cols(t) == 4


def _composed():
    idx = npo.RegularIndexer(0, 12, 3)
    idx2 = npo.RegularIndexer(0, 12, 2)
    idx3 = npo.RegularIndexer(0, 12, 3)
    filtered = npo.PeriodicRowFilter(idx, 3, 2)
    filtered2 = npo.PeriodicRowFilter(idx3, 2, 1)
    return npo.OuterProd(npo.RowConcat(filtered, idx2), filtered2)


def test_compose_indexers_kat():
    # test/indexing.cpp:42-86
    prod = _composed()
    assert prod.max_rows() == 5 and prod.max_cols() == 3                       # :56-60
    expect = [(2, 0), (2, 3), (5, 0), (2, 3), (2, 0), (5, 3)]                   # :62-77
    assert [(prod.rows(t), prod.cols(t)) for t in range(6)] == expect
    assert [prod.stride(t) for t in range(6)] == [0, 6, 0, 6, 0, 15]            # :80-85


def test_mat_seq_offsets_kat():
    # test/mat_seq.cpp:9-52: block 1 (2x3) at offset 0, block 5 (5x3) at offset 12, column-major
    idx = _composed()
    data = np.full(idx.required_memory(), np.nan)
    b1 = npo.mat_block(data, idx, 1)
    assert b1.shape == (2, 3)
    b1[0, 0], b1[1, 0], b1[0, 1] = 1.0, 2.0, 3.0
    b5 = npo.mat_block(data, idx, 5)
    assert b5.shape == (5, 3)
    b5[0, 0], b5[1, 0], b5[0, 1] = 4.0, 5.0, 6.0
    assert data[0 + 2 * 0 + 0] == 1.0 and data[0 + 2 * 0 + 1] == 2.0 and data[0 + 2 * 1 + 0] == 3.0      # :45-47
    assert data[12 + 5 * 0 + 0] == 4.0 and data[12 + 5 * 0 + 1] == 5.0 and data[12 + 5 * 1 + 0] == 6.0   # :49-51


def test_range_filter_and_shift():
    # test/pendulum_ddp.cpp:37-40 + problem.hpp:545 (two time shifts): rows only at t = horizon - 2
    horizon = 200
    unfiltered = npo.RegularIndexer(2, horizon + 2, 1)
    idx = npo.ShiftTimeIdx(npo.ShiftTimeIdx(npo.RangeRowFilter(unfiltered, horizon, horizon + 1), 1), 1)
    assert idx.index_begin() == 0 and idx.index_end() == horizon
    rows = [idx.rows(t) for t in range(horizon)]
    assert sum(rows) == 1 and rows[horizon - 2] == 1
    assert idx.required_memory() == 1


def test_tensor_layout():
    # detail/tensor.hpp:141-147: (i, j, k) at i + j*O + k*O*L
    O, L, R = 3, 4, 5
    flat = np.arange(O * L * R, dtype=np.float64)
    T = npo.tens(flat, 0, O, L, R)
    for (i, j, k) in [(0, 0, 0), (2, 1, 0), (1, 3, 4), (2, 3, 4)]:
        assert T[i, j, k] == i + j * O + k * O * L
