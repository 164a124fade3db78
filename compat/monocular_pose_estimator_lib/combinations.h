// combinations.h — Combinations (reference: lib/include/monocular_pose_estimator_lib/combinations.h:55-191,
// src/combinations.cpp).  Integer tables only, host code; the device kernels unrank the same orders
// on the fly.  Row orders are the reference's: combinations lexicographic; the permutations of one
// combination in the order of its recursive `permutations(K)` table (for K = 3: [c b a], [c a b],
// [b c a], [b a c], [a b c], [a c b]).  factorial() keeps the 32-bit wrap-around of the reference.
#ifndef MPE_COMPAT_COMBINATIONS_H_
#define MPE_COMPAT_COMBINATIONS_H_

#include "facade_namespace.h"
#include "datatypes.h"

MPE_FACADE_BEGIN

class Combinations {
 public:
  static unsigned factorial(int N) {  // combinations.cpp:34-40
    unsigned f = 1;
    for (int i = 2; i <= N; ++i) f *= (unsigned)i;
    return f;
  }
  static unsigned consecutiveProduct(unsigned k, unsigned n) {
    unsigned prod = 1;
    for (unsigned i = k; i <= n; ++i) prod *= i;
    return prod;
  }
  static std::vector<double> cumulativeProduct(unsigned k, unsigned n) {
    std::vector<double> out;
    double prod = 1;
    for (unsigned i = k; i <= n; ++i) {
      prod *= i;
      out.push_back(prod);
    }
    return out;
  }
  static unsigned numCombinations(unsigned N, unsigned K) {  // combinations.cpp:42-45
    return factorial((int)N) / (factorial((int)K) * factorial((int)(N - K)));
  }
  static unsigned numPermutations(unsigned N, unsigned K) { return factorial((int)N) / factorial((int)(N - K)); }

  static MatrixXYu combinationsNoReplacement(unsigned N, unsigned K) {
    RowXu list(N);
    for (unsigned i = 0; i < N; ++i) list[i] = i + 1;
    return combinationsNoReplacement(list, K);
  }
  //! K-subsets of `list` in lexicographic order of their positions (K == 1: a column, K == N: one row)
  static MatrixXYu combinationsNoReplacement(const RowXu& list, unsigned K) {
    const unsigned N = (unsigned)list.size();
    if (K == 0 || K > N) return MatrixXYu();
    std::vector<std::vector<unsigned> > rows;
    std::vector<unsigned> pos(K);
    for (unsigned i = 0; i < K; ++i) pos[i] = i;
    for (;;) {
      std::vector<unsigned> r(K);
      for (unsigned i = 0; i < K; ++i) r[i] = list[pos[i]];
      rows.push_back(r);
      int i = (int)K - 1;
      while (i >= 0 && pos[i] == N - K + (unsigned)i) --i;
      if (i < 0) break;
      ++pos[i];
      for (unsigned j = (unsigned)i + 1; j < K; ++j) pos[j] = pos[j - 1] + 1;
    }
    return fromRows(rows, K);
  }
  //! all N! orderings of 1..N in the reference's recursive order: leading N over permutations(N-1),
  //! then leading i = N-1 .. 1 over permutations(N-1) with i replaced by N
  static MatrixXYu permutations(unsigned N) {
    std::vector<std::vector<unsigned> > P(1, std::vector<unsigned>(1, 1u));
    for (unsigned n = 2; n <= N; ++n) {
      std::vector<std::vector<unsigned> > next;
      for (size_t r = 0; r < P.size(); ++r) next.push_back(prefixed(n, P[r], 0, 0));
      for (unsigned i = n - 1; i >= 1; --i)
        for (size_t r = 0; r < P.size(); ++r) next.push_back(prefixed(i, P[r], i, n));
      P.swap(next);
    }
    return fromRows(P, N);
  }
  static MatrixXYu permutationsNoReplacement(unsigned N, unsigned K) {
    if (K == 0 || K > N) return MatrixXYu();
    if (N == K) return permutations(N);
    const MatrixXYu C = combinationsNoReplacement(N, K);
    if (K == 1) return C;
    const MatrixXYu P = permutations(K);
    MatrixXYu out(C.rows() * P.rows(), K);
    for (size_t c = 0; c < C.rows(); ++c) {
      RowXu wv(K);
      for (unsigned j = 0; j < K; ++j) wv[j] = C(c, j);
      const MatrixXYu block = assignFromMatrixAsIndices(wv, P);
      for (size_t r = 0; r < P.rows(); ++r)
        for (unsigned j = 0; j < K; ++j) out(c * P.rows() + r, j) = block(r, j);
    }
    return out;
  }
  static MatrixXYu assignFromMatrixAsIndices(const RowXu& vec, const MatrixXYu& indices) {
    MatrixXYu out(indices.rows(), indices.cols());
    for (size_t r = 0; r < indices.rows(); ++r)
      for (size_t c = 0; c < indices.cols(); ++c) out(r, c) = vec[indices(r, c) - 1];
    return out;
  }
  static MatrixXYu setWhereEqualToConst(const MatrixXYu& matrix, const unsigned& checkNum, const unsigned& setNum) {
    MatrixXYu out = matrix;
    for (size_t r = 0; r < out.rows(); ++r)
      for (size_t c = 0; c < out.cols(); ++c)
        if (out(r, c) == checkNum) out(r, c) = setNum;
    return out;
  }

 private:
  static std::vector<unsigned> prefixed(unsigned lead, const std::vector<unsigned>& row, unsigned from, unsigned to) {
    std::vector<unsigned> r(1, lead);
    for (size_t i = 0; i < row.size(); ++i) r.push_back(from && row[i] == from ? to : row[i]);
    return r;
  }
  static MatrixXYu fromRows(const std::vector<std::vector<unsigned> >& rows, unsigned K) {
    MatrixXYu out(rows.size(), K);
    for (size_t r = 0; r < rows.size(); ++r)
      for (unsigned c = 0; c < K; ++c) out(r, c) = rows[r][c];
    return out;
  }
};

MPE_FACADE_END  // namespace monocular_pose_estimator
#endif
