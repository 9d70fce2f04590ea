// ddp/detail/mat_seq.hpp -- host-side mirror of mat_seq_t (include/ddp/detail/mat_seq.hpp:10-128) and of the
// rank-3 tensor view (detail/tensor.hpp:128-147): one contiguous buffer per sequence, NaN-poisoned at
// construction (:34-37), block t column-major at the indexer's offset.  data() is what crosses the C-ABI.
#pragma once
#include <limits>
#include <vector>

#include "ddp/indexer.hpp"

namespace ddp {
namespace detail {
namespace matrix_seq {

template <typename Scalar>
struct mat_view_t {   // column-major block, outer stride = rows (mat_seq.hpp:61-73)
  Scalar* m_data; index_t m_rows, m_cols;
  index_t rows() const { return m_rows; }
  index_t cols() const { return m_cols; }
  Scalar& operator()(index_t r, index_t c) const { assert(r < m_rows && c < m_cols); return m_data[r + c * m_rows]; }
  Scalar* data() const { return m_data; }
};

template <typename Scalar>
struct tensor_view_t {   // (i = out, j = left, k = right) at i + j*O + k*O*L  (tensor.hpp:141-147)
  Scalar* m_data; index_t m_outdim, m_indiml, m_indimr;
  Scalar& operator()(index_t i, index_t j, index_t k) const { return m_data[i + j * m_outdim + k * m_outdim * m_indiml]; }
};

template <typename Scalar>
struct mat_seq_t {
  indexing::indexer_ptr m_idx;
  std::vector<Scalar> m_data;
  explicit mat_seq_t(indexing::indexer_ptr idx)
      : m_idx(std::move(idx)), m_data(static_cast<size_t>(m_idx->required_memory()), std::numeric_limits<Scalar>::quiet_NaN()) {}
  Scalar* data() { return m_data.data(); }
  const Scalar* data() const { return m_data.data(); }
  index_t size() const { return static_cast<index_t>(m_data.size()); }
  mat_view_t<Scalar> operator[](index_t t) { return {m_data.data() + m_idx->offset(t), m_idx->rows(t), m_idx->cols(t)}; }
  mat_view_t<const Scalar> operator[](index_t t) const { return {m_data.data() + m_idx->offset(t), m_idx->rows(t), m_idx->cols(t)}; }
};

template <typename Scalar>
mat_seq_t<Scalar> mat_seq(indexing::indexer_ptr idx) { return mat_seq_t<Scalar>{std::move(idx)}; }

}  // namespace matrix_seq
}  // namespace detail
}  // namespace ddp
