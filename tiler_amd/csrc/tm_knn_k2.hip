// generated shape of tm_knn_kernel.h: database high-digit chunks HT = 2
#include "tm_knn_kernel.h"
namespace tmx {
TM_KNN_DEFINE_HT(2)
}
