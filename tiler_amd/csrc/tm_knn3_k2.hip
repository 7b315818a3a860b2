// one database digit plan (HT = 2 high chunks) of the third scan shape: seed and consume kernels for every query plan
#include "tm_knn3_kernel.h"
namespace tmx {
TM_KNN3_DEFINE_HT(2)
}
