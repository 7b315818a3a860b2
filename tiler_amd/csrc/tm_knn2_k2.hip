// generated shape of tm_knn2_kernel.h: database high-digit chunks HT = 2
#include "tm_knn2_kernel.h"
namespace tmx {
TM_KNN2_DEFINE_HT(2)
}
