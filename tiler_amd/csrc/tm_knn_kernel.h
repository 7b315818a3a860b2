// tm_knn_kernel.h -- what the KNN stage's kernels and its host code share (see tm_knn.hip for the scheme, tm_knn3_kernel.h for the scan).
#pragma once
#include <climits>
#include <type_traits>

#include "tm_common.h"

namespace tmx {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// Database tiles (32 rows): [6 low chunks | HT high chunks] x [64 lanes] x 16 B, then 32 u32 norms |t-c|^2, then the tile's box.
// Query tiles (32 queries): [6 low chunks | HQ high chunks] of the NEGATED centred values, then 32 u32 (|q-c|^2 >> 1).
// Columns are permuted so that the columns carrying a high digit are a prefix on each side (nested sets), widest first.
//   X = 65536 T_H . Q_H (m = min(HT, HQ) chunks) + 256 (T_L[:HQ] . Q_H + T_H . Q_L[:HT]) + T_L . Q_L (6 chunks)
//   d''  = |t-c|^2 + 2 X + 2 (|q-c|^2 >> 1) = SSD - (|q-c|^2 & 1), exact mod 2^32.
//
// Pruning.  Both sides arrive sorted along a Morton curve of the three widest feature columns (the Y/U/V DC terms for tile features) and
// the radial coordinate below, so 32 consecutive rows form a compact box.  For ND dimensions the database keeps per-tile bounding boxes
// (and boxes of runs of KNN_GROUP tiles); the squared box-to-box distance is a lower bound of every SSD between a query sub-tile's
// queries and a tile's rows.  Exactness: the minimum VALUE is exact (a tile is only skipped when bound > best + 1); when a second row
// reaches the same value the tie flag is raised and k_knn_ties settles the lowest-index rule.
constexpr int KNN_NC = 6;        // bounding-box columns
// One more box dimension, radial: R = |v - c| over all the OTHER columns.  |R(q) - R(t)|^2 <= the squared distance over those
// columns (reverse triangle inequality), so its gap adds to the lower bound like a column's; it tells a noisy tile from a smooth
// one of the same mean colour, which the widest (low-frequency) columns cannot.  Stored as integers rounded outwards.
constexpr int KNN_ND = KNN_NC + 1;
constexpr int KNN_GROUP = 128;   // tiles per second-level box

struct KnnBoxes {
  const int *lo, *hi;   // [KNN_ND][n_ttiles] bounding boxes of the database tiles
  const int *glo, *ghi; // [KNN_ND][ceil(n_ttiles / KNN_GROUP)] boxes of runs of KNN_GROUP tiles: a run no sub-tile can use is skipped whole
  const uint32_t *tkey; // [n_ttiles] curve key of each tile's first row (ascending)
  int col[KNN_NC];      // source feature column of each box dimension
  int cen[KNN_NC];      // the digit plan's centre of that column (the radial dimension is measured from the centres)
};

}  // namespace tmx
