// tools/probe_blocked.hip -- R&D micro-benchmark (not part of the product).
//
// Does a block-contiguous ("AoSoA", [block][population][BW sites]) order of
// the distributions move more bytes per second than the SoA order with its
// 19 far-apart streams? D3Q19 pull pattern on a 258^3 lattice: every site
// reads population p from site i - c_p and writes it at site i.
//   R = soa|blk: order of the array read;  W = soa|blk: order of the array
//   written. BW = sites per block of the blocked order.
//
// Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/probe_blocked.hip -o tools/probe_blocked

#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <functional>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { \
  printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int NS = 19;
__constant__ int c_shift[NS];

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

template <int BLK, int BW>
__device__ __forceinline__ size_t addr(size_t nsite, int p, long long i) {
  if constexpr (BLK) {
    return (size_t) (i/BW)*(size_t) (NS*BW) + (size_t) p*BW + (size_t) (i % BW);
  } else {
    return nsite*(size_t) p + (size_t) i;
  }
}

template <int RB, int WB, int BW, int BS>
__global__ __launch_bounds__(BS)
void k_pull(const double * __restrict__ f, double * __restrict__ fp,
	    size_t nsite, long long i0, long long i1, unsigned nblk, unsigned group) {
  extern __shared__ int lds_unused[];
  unsigned lb;
  if (!lblock(nblk, group, lb)) return;
  long long i = i0 + (long long) lb*BS + threadIdx.x;
  if (i >= i1) return;
  double v[NS];
#pragma unroll
  for (int p = 0; p < NS; p++) v[p] = f[addr<RB, BW>(nsite, p, i - c_shift[p])];
#pragma unroll
  for (int p = 0; p < NS; p++) fp[addr<WB, BW>(nsite, p, i)] = v[p]*1.0000001;
}

static double time_it(hipStream_t st, int reps, const std::function<void()> & launch);

// Two consecutive sites per thread, 16-byte accesses (global_load_dwordx4
// with 8-byte alignment for the z-shifted pulls). Blocked order, BW sites per
// block of BW/2 threads.
struct __attribute__((packed, aligned(8))) d2 { double a, b; };

template <int BW>
__global__ __launch_bounds__(BW/2)
void k_pull2(const double * __restrict__ f, double * __restrict__ fp,
	     size_t nsite, long long i0, long long i1, unsigned nblk, unsigned group) {
  extern __shared__ int lds_unused[];
  unsigned lb;
  if (!lblock(nblk, group, lb)) return;
  long long i = i0 + (long long) lb*BW + 2*threadIdx.x;
  if (i >= i1) return;
  d2 v[NS];
#pragma unroll
  for (int p = 0; p < NS; p++) {
    long long j = i - c_shift[p];
    // both sites of the pair lie in one block row unless j is the last site of a row
    if ((j % BW) != BW - 1) {
      v[p] = *reinterpret_cast<const d2 *>(&f[addr<1, BW>(nsite, p, j)]);
    } else {
      v[p].a = f[addr<1, BW>(nsite, p, j)];
      v[p].b = f[addr<1, BW>(nsite, p, j + 1)];
    }
  }
#pragma unroll
  for (int p = 0; p < NS; p++) {
    d2 w = {v[p].a*1.0000001, v[p].b*1.0000001};
    *reinterpret_cast<d2 *>(&fp[addr<1, BW>(nsite, p, i)]) = w;
  }
}

template <int BW>
void run2(hipStream_t st, const double * a, double * b, size_t nsite, unsigned lds,
	  unsigned group) {
  const long long strx = 258*258;
  const long long i0 = ((2*strx + 1023)/1024)*1024, i1 = ((255LL*strx)/1024)*1024;
  unsigned nblk = (unsigned) ((i1 - i0 + BW - 1)/BW);
  unsigned q = 8u*group;
  unsigned grid = ((nblk + q - 1)/q)*q;
  auto kern = k_pull2<BW>;
  double ms = time_it(st, 10, [&]{ hipLaunchKernelGGL(kern, dim3(grid), dim3(BW/2), lds, st, a, b, nsite, i0, i1, nblk, group); });
  double gb = 2.0*NS*8.0*(double) (i1 - i0)*1e-9;
  printf("pull2 (16 B/lane) blk->blk BW=%4d bs=%4d lds=%6u g=%3u  %7.3f ms %8.1f GB/s\n",
	 BW, BW/2, lds, group, ms, gb/ms*1e3);
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

template <int RB, int WB, int BW, int BS>
void run(hipStream_t st, const double * a, double * b, size_t nsite, unsigned lds,
	 unsigned group) {
  const long long strx = 258*258;
  // whole blocks only, well inside the array (shifts reach one x-plane)
  const long long i0 = ((2*strx + 1023)/1024)*1024, i1 = ((255LL*strx)/1024)*1024;
  unsigned nblk = (unsigned) ((i1 - i0 + BS - 1)/BS);
  unsigned q = 8u*group;
  unsigned grid = ((nblk + q - 1)/q)*q;
  auto kern = k_pull<RB, WB, BW, BS>;
  double ms = time_it(st, 10, [&]{ hipLaunchKernelGGL(kern, dim3(grid), dim3(BS), lds, st, a, b, nsite, i0, i1, nblk, group); });
  double gb = 2.0*NS*8.0*(double) (i1 - i0)*1e-9;
  printf("pull R=%s W=%s BW=%4d bs=%4d lds=%6u g=%3u  %7.3f ms %8.1f GB/s\n",
	 RB ? "blk" : "soa", WB ? "blk" : "soa", BW, BS, lds, group, ms, gb/ms*1e3);
}

int main() {
  const size_t nsite = 258ULL*258*258;
  const size_t ntot = nsite*NS + 4096*NS;
  double * a, * b;
  CHECK(hipMalloc(&a, ntot*sizeof(double)));
  CHECK(hipMalloc(&b, ntot*sizeof(double)));
  hipLaunchKernelGGL(k_fill_random, dim3(4096), dim3(256), 0, 0, a, ntot);
  hipLaunchKernelGGL(k_fill_random, dim3(4096), dim3(256), 0, 0, b, ntot);
  CHECK(hipDeviceSynchronize());
  {
    int sh[NS], n = 0;
    for (int x = -1; x <= 1; x++) for (int y = -1; y <= 1; y++) for (int z = -1; z <= 1; z++) {
      if (x*x + y*y + z*z <= 2) sh[n++] = x*258*258 + y*258 + z;
    }
    if (n != NS) { printf("bad set\n"); return 1; }
    CHECK(hipMemcpyToSymbol(HIP_SYMBOL(c_shift), sh, sizeof(sh)));
  }
  hipStream_t st;
  CHECK(hipStreamCreate(&st));
  for (int rep = 0; rep < 2; rep++) {
    for (unsigned lds : {0u, 65536u}) {
      run<0, 0, 256, 256>(st, a, b, nsite, lds, 16);
      run<0, 1, 256, 256>(st, a, b, nsite, lds, 16);
      run<1, 0, 256, 256>(st, a, b, nsite, lds, 16);
      run<1, 1, 256, 256>(st, a, b, nsite, lds, 16);
      run<1, 1, 64, 256>(st, a, b, nsite, lds, 16);
      run<1, 1, 1024, 256>(st, a, b, nsite, lds, 16);
      run<1, 1, 512, 512>(st, a, b, nsite, lds, 16);
      run<1, 1, 1024, 1024>(st, a, b, nsite, lds, 16);
    }
    run<1, 1, 256, 256>(st, a, b, nsite, 0, 1);
    run<1, 1, 256, 256>(st, a, b, nsite, 0, 4);
    run<1, 1, 256, 256>(st, a, b, nsite, 0, 64);
    run<1, 1, 256, 256>(st, a, b, nsite, 32768, 16);
    for (unsigned lds : {0u, 32768u, 65536u}) {
      run2<256>(st, a, b, nsite, lds, 16);
      run2<512>(st, a, b, nsite, lds, 16);
      run2<512>(st, a, b, nsite, lds, 32);
    }
  }
  return 0;
}
