// host/test_host.cpp -- the reference's own indexing tests (test/indexing.cpp, test/mat_seq.cpp) re-expressed against
// the host-side mirror headers.  No GPU needed.  Exit code 0 = all known answers reproduced.
#include <cstdio>

#include "ddp/detail/mat_seq.hpp"
#include "ddp/indexer.hpp"

static int failures = 0;
#define CHECK(cond) do { if (!(cond)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); ++failures; } } while (0)

int main() {
  using namespace ddp;
  {  // test/indexing.cpp:6-39 "regular indexer"
    auto idx = indexing::mat_regular_indexer(3, 12, 3, 4);
    CHECK(idx->index_begin() == 3);
    CHECK(idx->index_end() == 12);
    index_t t = idx->index_begin() + 4 - 2;
    CHECK(t == 5 && idx->rows(t) == 3 && idx->cols(t) == 4);
  }
  auto make_prod = [] {
    auto idx = indexing::vec_regular_indexer(0, 12, 3);
    auto idx2 = indexing::vec_regular_indexer(0, 12, 2);
    auto idx3 = indexing::vec_regular_indexer(0, 12, 3);
    auto filtered = indexing::periodic_row_filter(idx, 3, 2);
    auto filtered2 = indexing::periodic_row_filter(idx3, 2, 1);
    return indexing::outer_prod(indexing::row_concat(filtered, idx2), filtered2);
  };
  {  // test/indexing.cpp:42-86 "compose indexers"
    auto prod = make_prod();
    CHECK(prod->max_rows() == 5 && prod->max_cols() == 3);
    const index_t rows[6] = {2, 2, 5, 2, 2, 5}, cols[6] = {0, 3, 0, 3, 0, 3}, strides[6] = {0, 6, 0, 6, 0, 15};
    for (index_t t = 0; t < 6; ++t) CHECK(prod->rows(t) == rows[t] && prod->cols(t) == cols[t] && prod->stride(t) == strides[t]);
  }
  {  // test/mat_seq.cpp:9-52 "matrix sequence": flat column-major offsets 0 and 12
    auto seq = detail::matrix_seq::mat_seq<double>(make_prod());
    double* ptr = seq.data();
    auto b1 = seq[1];
    CHECK(b1.rows() == 2 && b1.cols() == 3);
    b1(0, 0) = 1.0; b1(1, 0) = 2.0; b1(0, 1) = 3.0;
    auto b5 = seq[5];
    CHECK(b5.rows() == 5 && b5.cols() == 3);
    b5(0, 0) = 4.0; b5(1, 0) = 5.0; b5(0, 1) = 6.0;
    CHECK(ptr[0 + 2 * 0 + 0] == 1.0 && ptr[0 + 2 * 0 + 1] == 2.0 && ptr[0 + 2 * 1 + 0] == 3.0);
    CHECK(ptr[12 + 5 * 0 + 0] == 4.0 && ptr[12 + 5 * 0 + 1] == 5.0 && ptr[12 + 5 * 1 + 0] == 6.0);
    CHECK(seq[0].rows() == 2 && seq[0].cols() == 0);
    CHECK(ptr[3] != ptr[3]);   // untouched storage stays NaN-poisoned (mat_seq.hpp:34-37)
  }
  {  // test/pendulum_ddp.cpp:37-40 + problem.hpp:545: rows only at solver time horizon - 2
    const index_t horizon = 200;
    auto eq = indexing::shift_time_idx(indexing::shift_time_idx(
        indexing::range_row_filter(indexing::vec_regular_indexer(2, horizon + 2, 1), horizon, horizon + 1), 1), 1);
    CHECK(eq->index_begin() == 0 && eq->index_end() == horizon && eq->required_memory() == 1);
    CHECK(eq->rows(horizon - 2) == 1 && eq->rows(horizon - 1) == 0 && eq->rows(0) == 0);
  }
  {  // detail/tensor.hpp:141-147
    double buf[3 * 4 * 5];
    for (int i = 0; i < 60; ++i) buf[i] = i;
    detail::matrix_seq::tensor_view_t<double> T{buf, 3, 4, 5};
    CHECK(T(2, 1, 0) == 2 + 1 * 3 && T(1, 3, 4) == 1 + 3 * 3 + 4 * 12);
  }
  std::printf(failures ? "%d check(s) failed\n" : "host mirror: all known answers reproduced\n", failures);
  return failures ? 1 : 0;
}
