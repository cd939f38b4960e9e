// tools/probe_streams2.hip -- R&D micro-benchmark (not part of the product).
//
// How does MI355X HBM throughput depend on (a) the number of concurrent
// streams, (b) reads vs writes, (c) bytes per stream per wave (SPT chunks,
// p-major), (d) resident waves per CU (dynamic-LDS cap)?
// All kernels: NS read streams and/or NS write streams of `n` doubles each,
// separated like the populations of a 258^3 lattice; one site per lane.
//
// Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/probe_streams2.hip -o tools/probe_streams2

#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <functional>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { \
  printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

__global__ void k_fill_random(double * a, size_t n) {
  size_t i = (size_t) blockIdx.x*blockDim.x + threadIdx.x;
  size_t stride = (size_t) gridDim.x*blockDim.x;
  for (; i < n; i += stride) {
    unsigned long long s = i*6364136223846793005ULL + 1442695040888963407ULL;
    s ^= s >> 29; s *= 0xBF58476D1CE4E5B9ULL; s ^= s >> 32;
    a[i] = 0.05*(1.0 + 1.0e-3*((double) (s >> 11)*(1.0/9007199254740992.0) - 0.5));
  }
}

__device__ __forceinline__ bool lblock(unsigned nblk, unsigned group, unsigned & lb) {
  unsigned xcd = blockIdx.x & 7u, j = blockIdx.x >> 3;
  unsigned grp = j/group, within = j - grp*group;
  lb = (grp*8u + xcd)*group + within;
  return lb < nblk;
}

// MODE 0: copy (NS reads + NS writes); 1: read only (sum to 1 write stream);
// 2: write only. SPT chunks of 64 sites per wave, p-major.
template <int NS, int MODE, int SPT, int SHIFT>
__global__ void k_streams(const double * __restrict__ f, double * __restrict__ fp,
			  size_t nsite, long long i0, long long i1, unsigned nblk, unsigned group) {
  extern __shared__ int lds_unused[];
  unsigned lb;
  if (!lblock(nblk, group, lb)) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  long long base = i0 + ((long long) lb*wpb + wave)*(64*SPT) + lane;
  double v[NS][SPT];
  if constexpr (MODE != 2) {
#pragma unroll
    for (int p = 0; p < NS; p++) {
      long long sh = SHIFT ? ((p % 3) - 1) + 258*(((p/3) % 3) - 1) : 0;
#pragma unroll
      for (int k = 0; k < SPT; k++) {
	long long i = base + 64*k;
	v[p][k] = (i < i1) ? f[nsite*p + i - sh] : 0.0;
      }
    }
  } else {
#pragma unroll
    for (int p = 0; p < NS; p++)
#pragma unroll
      for (int k = 0; k < SPT; k++) v[p][k] = 1.0 + p + k;
  }
  if constexpr (MODE == 3) {
    // in-place: write back (modified) to the addresses just read, as the
    // AA-pattern does; f and fp are the same array
    double * fw = const_cast<double *>(f);
#pragma unroll
    for (int p = 0; p < NS; p++) {
      long long sh = SHIFT ? ((p % 3) - 1) + 258*(((p/3) % 3) - 1) : 0;
#pragma unroll
      for (int k = 0; k < SPT; k++) {
	long long i = base + 64*k;
	if (i < i1) fw[nsite*p + i - sh] = v[p][k]*1.0000001;
      }
    }
  } else if constexpr (MODE == 1) {
#pragma unroll
    for (int k = 0; k < SPT; k++) {
      double s = 0.0;
#pragma unroll
      for (int p = 0; p < NS; p++) s += v[p][k];
      long long i = base + 64*k;
      if (i < i1 && s == 123.456) fp[i] = s;     // never true: keeps loads live
    }
  } else {
#pragma unroll
    for (int p = 0; p < NS; p++) {
#pragma unroll
      for (int k = 0; k < SPT; k++) {
	long long i = base + 64*k;
	if (i < i1) fp[nsite*p + i] = v[p][k];
      }
    }
  }
}

static double time_it(hipStream_t st, int reps, const std::function<void()> & launch) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int r = 0; r < 2; r++) launch();
  CHECK(hipStreamSynchronize(st));
  CHECK(hipEventRecord(e0, st));
  for (int r = 0; r < reps; r++) launch();
  CHECK(hipEventRecord(e1, st));
  CHECK(hipStreamSynchronize(st));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  CHECK(hipGetLastError());
  return ms/reps;
}

template <int NS, int MODE, int SPT, int SHIFT>
void run(const char * name, hipStream_t st, const double * a, double * b, size_t nsite,
	 int bs, unsigned lds, unsigned group, int ioff = 0) {
  const size_t strx = 258*258;
  const long long i0 = 2*strx + ioff, i1 = 255LL*strx;
  const long long per = (long long) (bs/64)*64*SPT;
  unsigned nblk = (unsigned) ((i1 - i0 + per - 1)/per);
  unsigned q = 8u*group;
  unsigned grid = ((nblk + q - 1)/q)*q;
  auto kern = k_streams<NS, MODE, SPT, SHIFT>;
  if (lds > 65536) CHECK(hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
  double ms = time_it(st, 10, [&]{ hipLaunchKernelGGL(kern, dim3(grid), dim3(bs), lds, st, a, b, nsite, i0, i1, nblk, group); });
  double nstream = (MODE == 0 || MODE == 3) ? 2.0*NS : (double) NS;
  double gb = nstream*8.0*(double) (i1 - i0)*1e-9;
  printf("%-28s NS=%2d spt=%d bs=%4d lds=%6u g=%3u i0%%16=%2d  %7.3f ms %8.1f GB/s\n", name, NS, SPT, bs, lds, group, (int) (i0 % 16), ms, gb/ms*1e3);
}

int main() {
  const size_t nsite = 258ULL*258*258;
  const size_t ntot = nsite*27;
  double * a, * b;
  CHECK(hipMalloc(&a, ntot*sizeof(double)));
  CHECK(hipMalloc(&b, ntot*sizeof(double)));
  hipLaunchKernelGGL(k_fill_random, dim3(4096), dim3(256), 0, 0, a, ntot);
  hipLaunchKernelGGL(k_fill_random, dim3(4096), dim3(256), 0, 0, b, ntot);
  CHECK(hipDeviceSynchronize());
  hipStream_t st;
  CHECK(hipStreamCreate(&st));

  for (int rep = 0; rep < 2; rep++) {
    for (unsigned lds : {0u, 32768u, 65536u}) {
      run<19, 0, 1, 1>("copy shift (2 arrays)", st, a, b, nsite, 256, lds, 16, 0);
      run<19, 3, 1, 0>("in-place aligned", st, a, b, nsite, 256, lds, 16, 0);
      run<19, 3, 1, 1>("in-place shifted", st, a, b, nsite, 256, lds, 16, 0);
      run<27, 3, 1, 1>("in-place shifted", st, a, b, nsite, 256, lds, 16, 0);
    }
  }
  return 0;
}
