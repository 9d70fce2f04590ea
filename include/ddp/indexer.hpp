// ddp/indexer.hpp -- host-side mirror of the reference's composable time indexers (include/ddp/indexer.hpp:152-493),
// Eigen-free and with run-time dimensions.  Same names and meaning: an indexer maps a time index t to the shape
// (rows(t), cols(t)) and stride(t) = rows*cols of block t of a flat sequence; block t starts at the sum of the
// strides before it (detail/mat_seq.hpp:61-73).  Known answers of test/indexing.cpp are checked in host/test_host.cpp.
#pragma once
#include <cassert>
#include <cstdint>
#include <memory>

namespace ddp {
using index_t = std::int64_t;   // detail/utils.hpp:117

namespace indexing {

struct indexer_base {
  virtual ~indexer_base() = default;
  virtual index_t index_begin() const = 0;
  virtual index_t index_end() const = 0;
  virtual index_t rows(index_t t) const = 0;
  virtual index_t cols(index_t t) const = 0;
  virtual index_t max_rows() const = 0;
  virtual index_t max_cols() const = 0;
  virtual index_t stride(index_t t) const { return rows(t) * cols(t); }
  virtual index_t required_memory() const {
    index_t s = 0;
    for (index_t t = index_begin(); t < index_end(); ++t) s += stride(t);
    return s;
  }
  index_t offset(index_t t) const {   // memory offset of block t (what the reference's iterators accumulate)
    index_t s = 0;
    for (index_t i = index_begin(); i < t; ++i) s += stride(i);
    return s;
  }
};
using indexer_ptr = std::shared_ptr<const indexer_base>;

// regular_indexer_t, indexer.hpp:249-289
struct regular_indexer_t final : indexer_base {
  index_t m_begin, m_end, m_rows, m_cols;
  regular_indexer_t(index_t b, index_t e, index_t r, index_t c) : m_begin(b), m_end(e), m_rows(r), m_cols(c) { assert(b < e); }
  index_t index_begin() const override { return m_begin; }
  index_t index_end() const override { return m_end; }
  index_t rows(index_t) const override { return m_rows; }
  index_t cols(index_t) const override { return m_cols; }
  index_t max_rows() const override { return m_rows; }
  index_t max_cols() const override { return m_cols; }
  index_t required_memory() const override { return m_rows * m_cols * (m_end - m_begin); }
};
// periodic_row_filter_t, indexer.hpp:395-447
struct periodic_row_filter_t final : indexer_base {
  indexer_ptr m_idx; index_t m_period, m_first_offset;
  periodic_row_filter_t(indexer_ptr i, index_t p, index_t f) : m_idx(std::move(i)), m_period(p), m_first_offset(f) { assert(p > 0 && f < p); }
  index_t index_begin() const override { return m_idx->index_begin(); }
  index_t index_end() const override { return m_idx->index_end(); }
  index_t rows(index_t t) const override { return (t - m_idx->index_begin()) % m_period == m_first_offset ? m_idx->rows(t) : 0; }
  index_t cols(index_t) const override { return 1; }
  index_t max_rows() const override { return m_idx->max_rows(); }
  index_t max_cols() const override { return 1; }
};
// range_row_filter_t, indexer.hpp:328-393
struct range_row_filter_t final : indexer_base {
  indexer_ptr m_idx; index_t m_range_begin, m_range_end;
  range_row_filter_t(indexer_ptr i, index_t b, index_t e) : m_idx(std::move(i)), m_range_begin(b), m_range_end(e) {}
  index_t index_begin() const override { return m_idx->index_begin(); }
  index_t index_end() const override { return m_idx->index_end(); }
  index_t rows(index_t t) const override { return (t >= m_range_begin && t < m_range_end) ? m_idx->rows(t) : 0; }
  index_t cols(index_t) const override { return 1; }
  index_t max_rows() const override { return m_idx->max_rows(); }
  index_t max_cols() const override { return 1; }
};
// shift_time_idx_t, indexer.hpp:291-318
struct shift_time_idx_t final : indexer_base {
  indexer_ptr m_idx; index_t m_dt;
  shift_time_idx_t(indexer_ptr i, index_t dt) : m_idx(std::move(i)), m_dt(dt) {}
  index_t index_begin() const override { return m_idx->index_begin() - m_dt; }
  index_t index_end() const override { return m_idx->index_end() - m_dt; }
  index_t rows(index_t t) const override { return m_idx->rows(t + m_dt); }
  index_t cols(index_t t) const override { return m_idx->cols(t + m_dt); }
  index_t max_rows() const override { return m_idx->max_rows(); }
  index_t max_cols() const override { return m_idx->max_cols(); }
  index_t stride(index_t t) const override { return m_idx->stride(t + m_dt); }
};
// row_concat_indexer_t, indexer.hpp:152-195
struct row_concat_indexer_t final : indexer_base {
  indexer_ptr m_idx_l, m_idx_r;
  row_concat_indexer_t(indexer_ptr l, indexer_ptr r) : m_idx_l(std::move(l)), m_idx_r(std::move(r)) {
    assert(m_idx_l->index_begin() == m_idx_r->index_begin() && m_idx_l->index_end() == m_idx_r->index_end());
  }
  index_t index_begin() const override { return m_idx_l->index_begin(); }
  index_t index_end() const override { return m_idx_l->index_end(); }
  index_t rows(index_t t) const override { return m_idx_l->rows(t) + m_idx_r->rows(t); }
  index_t cols(index_t t) const override { return m_idx_l->cols(t); }
  index_t max_rows() const override { return m_idx_l->max_rows() + m_idx_r->max_rows(); }
  index_t max_cols() const override { return m_idx_l->max_cols(); }
  index_t stride(index_t t) const override { return m_idx_l->stride(t) + m_idx_r->stride(t); }
};
// outer_product_indexer_t, indexer.hpp:197-247
struct outer_product_indexer_t final : indexer_base {
  indexer_ptr m_idx_l, m_idx_r;
  outer_product_indexer_t(indexer_ptr l, indexer_ptr r) : m_idx_l(std::move(l)), m_idx_r(std::move(r)) {
    assert(m_idx_l->index_begin() == m_idx_r->index_begin() && m_idx_l->index_end() == m_idx_r->index_end());
  }
  index_t index_begin() const override { return m_idx_l->index_begin(); }
  index_t index_end() const override { return m_idx_l->index_end(); }
  index_t rows(index_t t) const override { return m_idx_l->rows(t); }
  index_t cols(index_t t) const override { return m_idx_r->rows(t); }
  index_t max_rows() const override { return m_idx_l->max_rows(); }
  index_t max_cols() const override { return m_idx_r->max_rows(); }
};

inline indexer_ptr mat_regular_indexer(index_t b, index_t e, index_t r, index_t c) { return std::make_shared<regular_indexer_t>(b, e, r, c); }
inline indexer_ptr vec_regular_indexer(index_t b, index_t e, index_t r) { return std::make_shared<regular_indexer_t>(b, e, r, 1); }
inline indexer_ptr periodic_row_filter(indexer_ptr i, index_t period, index_t first) { return std::make_shared<periodic_row_filter_t>(std::move(i), period, first); }
inline indexer_ptr range_row_filter(indexer_ptr i, index_t b, index_t e) { return std::make_shared<range_row_filter_t>(std::move(i), b, e); }
inline indexer_ptr shift_time_idx(indexer_ptr i, index_t dt) { return std::make_shared<shift_time_idx_t>(std::move(i), dt); }
inline indexer_ptr row_concat(indexer_ptr l, indexer_ptr r) { return std::make_shared<row_concat_indexer_t>(std::move(l), std::move(r)); }
inline indexer_ptr outer_prod(indexer_ptr l, indexer_ptr r) { return std::make_shared<outer_product_indexer_t>(std::move(l), std::move(r)); }

}  // namespace indexing
}  // namespace ddp
