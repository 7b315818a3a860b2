// generated shape of tm_knn2_kernel.h: database high-digit chunks HT = 4
#include "tm_knn2_kernel.h"
namespace tmx {
TM_KNN2_DEFINE_HT(4)
}
