// development aid: what a 4-byte read-back behind a small kernel costs -- pageable destination against a pinned one
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_touch(int *p) { p[0] += 1; }
int main() {
  int *d; hipMalloc(&d, 4); hipMemset(d, 0, 4);
  int *pin; hipHostMalloc(&pin, 4);
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  int stack = 0;
  for (int mode = 0; mode < 3; mode++) {
    for (int rep = 0; rep < 2; rep++) {
      auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < 200; i++) {
        hipLaunchKernelGGL(k_touch, dim3(1), dim3(1), 0, s, d);
        if (mode == 0) { hipMemcpyAsync(&stack, d, 4, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); }
        else if (mode == 1) { hipMemcpyAsync(pin, d, 4, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); stack = *pin; }
        else { hipStreamSynchronize(s); }
      }
      auto t1 = std::chrono::steady_clock::now();
      if (rep) printf("mode %d (%s): %.2f us per launch + read-back\n", mode, mode == 0 ? "pageable" : mode == 1 ? "pinned" : "sync only",
                      std::chrono::duration<double, std::micro>(t1 - t0).count() / 200);
    }
  }
  return stack == 12345;
}
