// tools/probe_mall.hip -- R&D micro-benchmark (not part of the product).
//
// Question: does the 256 MiB Infinity Cache serve a re-read of lines that a
// kernel has just WRITTEN, and is a two-pass sweep (F -> G, then G -> F) faster
// when the second pass follows the first by a few chunks instead of by the
// whole array? That is the memory pattern of two LB time steps pipelined over
// x slabs: step t+1 of slab j-1 reads what step t of slab j-1 wrote one launch
// earlier.
//
//   baseline : copy F -> G (all), copy G -> F (all)              4 units of HBM traffic
//   pipelined: launch j = { F[j] -> G[j]  ||  G[j-1] -> F[j-1] }  3 units if G[j-1] is still on die
//
// "copy" adds 1.0 so that nothing can be elided. Prints the effective rate
// (4 * bytes / time) for chunk sizes from 5 to 320 MB.
//
// Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/probe_mall.hip -o tools/probe_mall

#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { \
  printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

// two ranges in one launch: blocks [0, nb1) do a1 -> b1 over n1 doubles,
// blocks [nb1, nb1 + nb2) do a2 -> b2 over n2 doubles; one double per lane
__global__ __launch_bounds__(256)
void k_two(const double * __restrict__ a1, double * __restrict__ b1, size_t n1, unsigned nb1,
	   const double * __restrict__ a2, double * __restrict__ b2, size_t n2) {
  unsigned b = blockIdx.x;
  if (b < nb1) {
    size_t i = (size_t) b*256 + threadIdx.x;
    if (i < n1) b1[i] = a1[i] + 1.0;
  }
  else {
    size_t i = (size_t) (b - nb1)*256 + threadIdx.x;
    if (i < n2) b2[i] = a2[i] + 1.0;
  }
}

int main() {
  const size_t total = (size_t) 2560 << 20;          // bytes per array (2.5 GiB: D3Q19 256^3 is 2.6 GB)
  const size_t n = total/8;
  double *F, *G;
  CHECK(hipMalloc(&F, total));
  CHECK(hipMalloc(&G, total));
  CHECK(hipMemset(F, 0, total));
  CHECK(hipMemset(G, 0, total));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));

  auto launch = [&](const double * a1, double * b1, size_t n1, const double * a2, double * b2, size_t n2) {
    unsigned nb1 = (unsigned) ((n1 + 255)/256), nb2 = (unsigned) ((n2 + 255)/256);
    if (nb1 + nb2 == 0) return;
    hipLaunchKernelGGL(k_two, dim3(nb1 + nb2), dim3(256), 0, 0, a1, b1, n1, nb1, a2, b2, n2);
  };

  for (int rep = 0; rep < 2; rep++) {
    // baseline: two whole passes
    CHECK(hipEventRecord(e0));
    for (int it = 0; it < 4; it++) {
      launch(F, G, n, nullptr, nullptr, 0);
      launch(G, F, n, nullptr, nullptr, 0);
    }
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("baseline (two whole passes)        : %8.3f ms per pair of passes, %6.0f GB/s effective\n",
	   ms/4, 4.0*total/(ms/4*1e-3)*1e-9);

    const int chunk_mb[] = {5, 10, 20, 40, 80, 160, 320};
    for (int c : chunk_mb) {
      size_t cn = ((size_t) c << 20)/8;
      size_t nchunk = (n + cn - 1)/cn;
      CHECK(hipEventRecord(e0));
      for (int it = 0; it < 4; it++) {
	for (size_t j = 0; j <= nchunk; j++) {
	  size_t o1 = j*cn, o2 = (j - 1)*cn;
	  size_t n1 = (j < nchunk) ? ((o1 + cn <= n) ? cn : n - o1) : 0;
	  size_t n2 = (j > 0) ? ((o2 + cn <= n) ? cn : n - o2) : 0;
	  launch(F + (n1 ? o1 : 0), G + (n1 ? o1 : 0), n1, G + (n2 ? o2 : 0), F + (n2 ? o2 : 0), n2);
	}
      }
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      printf("pipelined, chunks of %3d MB (%4zu launches): %8.3f ms per pair of passes, %6.0f GB/s effective\n",
	     c, nchunk + 1, ms/4, 4.0*total/(ms/4*1e-3)*1e-9);
    }
  }
  return 0;
}
