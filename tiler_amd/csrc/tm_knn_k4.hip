// generated shape of tm_knn_kernel.h: database high-digit chunks HT = 4
#include "tm_knn_kernel.h"
namespace tmx {
TM_KNN_DEFINE_HT(4)
}
