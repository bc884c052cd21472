#include <hip/hip_runtime.h>
#include <rocsolver/rocsolver.h>
#include <cstdio>
#include <vector>
#include <chrono>
int main() {
  for (int n : {216, 384, 768}) {
    std::vector<double> A(size_t(n) * n, 0.0);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) A[size_t(i) * n + j] = (i == j ? n : 0.0) + 1.0 / (1 + abs(i - j));
    double* dA; int* info; double* dB;
    hipMalloc(&dA, A.size() * 8); hipMalloc(&info, 4); hipMalloc(&dB, n * 8);
    rocblas_handle h; rocblas_create_handle(&h);
    hipStream_t s; hipStreamCreate(&s); rocblas_set_stream(h, s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 6; ++rep) {
      hipMemcpyAsync(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice, s);
      hipMemcpyAsync(dB, A.data(), n * 8, hipMemcpyHostToDevice, s);
      hipEventRecord(e0, s);
      rocsolver_dpotrf(h, rocblas_fill_upper, n, dA, n, info);
      hipEventRecord(e1, s);
      hipStreamSynchronize(s);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      hipEventRecord(e0, s);
      rocblas_dtrsv(h, rocblas_fill_upper, rocblas_operation_transpose, rocblas_diagonal_non_unit, n, dA, n, dB, 1);
      hipEventRecord(e1, s);
      hipStreamSynchronize(s);
      float ms2; hipEventElapsedTime(&ms2, e0, e1);
      if (rep >= 3) printf("n %d: dpotrf %.1f us, dtrsv %.1f us\n", n, ms * 1e3, ms2 * 1e3);
    }
  }
  return 0;
}
