// tools/probe_stream.hip -- R&D micro-benchmark (not part of the product).
//
// Measures, on one MI355X, what the memory system delivers for the access
// patterns that matter to the D3Q19 pull-stream kernel on the reference's
// SoA layout (19 population arrays of nsite doubles, z fastest, row length
// nall_z = 258 doubles): single-stream copies (calibration), 19-stream
// copies with and without the +-1 site shifts, 8 vs 16 bytes per lane,
// nontemporal hints, XCD-chunked vs round-robin block mapping, block size.
//
// Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/probe_stream.hip -o tools/probe_stream
// Run:   tools/probe_stream            (prints one line per variant)

#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <functional>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { \
  printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int NVEL = 19;
__constant__ int c_cv[NVEL][3];
static const int h_cv[NVEL][3] = {
  { 0, 0, 0},
  { 1, 1, 0}, { 1, 0, 1}, { 1, 0, 0}, { 1, 0,-1}, { 1,-1, 0},
  { 0, 1, 1}, { 0, 1, 0}, { 0, 1,-1}, { 0, 0, 1}, { 0, 0,-1},
  { 0,-1, 1}, { 0,-1, 0}, { 0,-1,-1},
  {-1, 1, 0}, {-1, 0, 1}, {-1, 0, 0}, {-1, 0,-1}, {-1,-1, 0}};

struct Off { int o[NVEL]; };

__global__ void k_fill_random(double * a, size_t n) {
  size_t i = (size_t) blockIdx.x*blockDim.x + threadIdx.x;
  size_t stride = (size_t) gridDim.x*blockDim.x;
  for (; i < n; i += stride) {
    unsigned long long s = i*6364136223846793005ULL + 1442695040888963407ULL;
    s ^= s >> 29; s *= 0xBF58476D1CE4E5B9ULL; s ^= s >> 32;
    a[i] = 0.05*(1.0 + 1.0e-3*((double) (s >> 11)*(1.0/9007199254740992.0) - 0.5));
  }
}

// ---- single-stream calibration copies ------------------------------------

template <typename T>
__global__ void k_copy1(const T * __restrict__ a, T * __restrict__ b, size_t n) {
  size_t i = (size_t) blockIdx.x*blockDim.x + threadIdx.x;
  size_t stride = (size_t) gridDim.x*blockDim.x;
  for (; i < n; i += stride) b[i] = a[i];
}

template <typename T>
__global__ void k_copy1_flat(const T * __restrict__ a, T * __restrict__ b, size_t n) {
  size_t i = (size_t) blockIdx.x*blockDim.x + threadIdx.x;
  if (i < n) b[i] = a[i];
}

// ---- 19-stream copies ---------------------------------------------------------

template <bool XCD>
__device__ __forceinline__ bool lblock(unsigned nblk, unsigned & lb) {
  if constexpr (XCD) {
    unsigned per = (nblk + 7u) >> 3;
    lb = (blockIdx.x & 7u)*per + (blockIdx.x >> 3);
    return lb < nblk;
  } else {
    lb = blockIdx.x;
    return lb < nblk;
  }
}

// decode + mask exactly as lbmi_kernels.hip:k_propagate does
template <bool XCD>
__global__ void k_soa19_decode(const double * __restrict__ f, double * __restrict__ fp,
			       size_t nsite, int i0, int i1, unsigned nblk, Off off,
			       int strx, int stry, int nh, int nly, int nlz) {
  unsigned lb;
  if (!lblock<XCD>(nblk, lb)) return;
  int i = i0 + (int) (lb*blockDim.x + threadIdx.x);
  if (i >= i1) return;
  int x = i / strx;
  int r = i - x*strx;
  int y = r / stry;
  int z = r - y*stry;
  int m = ((y >= nh) && (y < nh + nly) && (z >= nh) && (z < nh + nlz)) ? 1 : 0;
#pragma unroll
  for (int p = 0; p < NVEL; p++) {
    fp[nsite*p + i] = f[nsite*p + (i - m*off.o[p])];
  }
}

typedef double double2_t __attribute__((ext_vector_type(2)));
typedef double double2_u __attribute__((ext_vector_type(2), aligned(8)));

// VEC = 1: one site per lane (8 B); VEC = 2: two sites per lane (16 B)
template <int VEC, bool SHIFT, bool NT, bool XCD>
__global__ void k_soa19(const double * __restrict__ f, double * __restrict__ fp,
			size_t nsite, long long i0, long long i1, unsigned nblk, Off off) {
  unsigned lb;
  if (!lblock<XCD>(nblk, lb)) return;
  long long i = i0 + ((long long) lb*blockDim.x + threadIdx.x)*VEC;
  if (i >= i1) return;
#pragma unroll
  for (int p = 0; p < NVEL; p++) {
    long long src = i - (SHIFT ? off.o[p] : 0);
    if constexpr (VEC == 1) {
      double v;
      if constexpr (NT) v = __builtin_nontemporal_load(&f[nsite*p + src]);
      else v = f[nsite*p + src];
      if constexpr (NT) __builtin_nontemporal_store(v, &fp[nsite*p + i]);
      else fp[nsite*p + i] = v;
    } else {
      double2_t v;
      const double2_u * s = reinterpret_cast<const double2_u *>(&f[nsite*p + src]);
      double2_u * d = reinterpret_cast<double2_u *>(&fp[nsite*p + i]);
      if constexpr (NT) v = __builtin_nontemporal_load(s);
      else v = *s;
      if constexpr (NT) __builtin_nontemporal_store(v, d);
      else *d = v;
    }
  }
}

// loads all first (as the collision kernel must), then stores
template <int VEC, bool SHIFT, bool NT, bool XCD>
__global__ void k_soa19_ls(const double * __restrict__ f, double * __restrict__ fp,
			   size_t nsite, long long i0, long long i1, unsigned nblk, Off off) {
  unsigned lb;
  if (!lblock<XCD>(nblk, lb)) return;
  long long i = i0 + ((long long) lb*blockDim.x + threadIdx.x)*VEC;
  if (i >= i1) return;
  double2_t v[NVEL];
#pragma unroll
  for (int p = 0; p < NVEL; p++) {
    long long src = i - (SHIFT ? off.o[p] : 0);
    if constexpr (VEC == 1) {
      v[p].x = NT ? __builtin_nontemporal_load(&f[nsite*p + src]) : f[nsite*p + src];
    } else {
      const double2_u * s = reinterpret_cast<const double2_u *>(&f[nsite*p + src]);
      v[p] = NT ? __builtin_nontemporal_load(s) : *s;
    }
  }
#pragma unroll
  for (int p = 0; p < NVEL; p++) {
    if constexpr (VEC == 1) {
      if constexpr (NT) __builtin_nontemporal_store(v[p].x, &fp[nsite*p + i]);
      else fp[nsite*p + i] = v[p].x;
    } else {
      double2_u * d = reinterpret_cast<double2_u *>(&fp[nsite*p + i]);
      if constexpr (NT) __builtin_nontemporal_store(v[p], d);
      else *d = v[p];
    }
  }
}





static double time_it(hipStream_t st, int reps, const std::function<void()> & launch) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int r = 0; r < 3; r++) launch();
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

int main(int argc, char ** argv) {
  const int nx = 256, nall = 258;
  const size_t strx = (size_t) nall*nall, stry = nall;
  const size_t nsite = (size_t) nall*nall*nall;
  const size_t ntot = nsite*NVEL;
  double * a, * b;
  CHECK(hipMalloc(&a, ntot*sizeof(double)));
  CHECK(hipMalloc(&b, ntot*sizeof(double)));
  if (argc > 1 && atoi(argv[1]) == 0) {
    CHECK(hipMemset(a, 0, ntot*sizeof(double)));
    CHECK(hipMemset(b, 0, ntot*sizeof(double)));
    printf("# data: zeros\n");
  } else {
    hipLaunchKernelGGL(k_fill_random, dim3(4096), dim3(256), 0, 0, a, ntot);
    hipLaunchKernelGGL(k_fill_random, dim3(4096), dim3(256), 0, 0, b, ntot);
    CHECK(hipDeviceSynchronize());
    printf("# data: random\n");
  }
  hipStream_t st;
  CHECK(hipStreamCreate(&st));
  Off off;
  for (int p = 0; p < NVEL; p++) off.o[p] = (int) (h_cv[p][0]*strx + h_cv[p][1]*stry + h_cv[p][2]);

  const int reps = 20;
  // sites processed by the 19-stream kernels: x planes 2..255 (stay in bounds)
  const long long i0 = 2*strx, i1 = (long long) (nx - 1)*strx;
  const double sites = (double) (i1 - i0);
  const double gb19 = 2.0*NVEL*8.0*sites*1e-9;

  printf("# nsite %zu, 19-stream sites %.0f, bytes moved per launch %.3f GB\n", nsite, sites, gb19);

  {
    size_t n4 = ntot/2;   // double2 elements
    double ms = time_it(st, reps, [&]{ hipLaunchKernelGGL((k_copy1<double2_t>), dim3(2048), dim3(256), 0, st, (const double2_t*) a, (double2_t*) b, n4); });
    printf("copy1 16B gridstride 2048x256      %8.3f ms %8.1f GB/s\n", ms, 2.0*ntot*8*1e-6/ms);
    ms = time_it(st, reps, [&]{ hipLaunchKernelGGL((k_copy1<double2_t>), dim3(8192), dim3(256), 0, st, (const double2_t*) a, (double2_t*) b, n4); });
    printf("copy1 16B gridstride 8192x256      %8.3f ms %8.1f GB/s\n", ms, 2.0*ntot*8*1e-6/ms);
    ms = time_it(st, reps, [&]{ hipLaunchKernelGGL((k_copy1_flat<double2_t>), dim3((n4 + 255)/256), dim3(256), 0, st, (const double2_t*) a, (double2_t*) b, n4); });
    printf("copy1 16B flat                     %8.3f ms %8.1f GB/s\n", ms, 2.0*ntot*8*1e-6/ms);
    ms = time_it(st, reps, [&]{ hipLaunchKernelGGL((k_copy1_flat<double>), dim3((ntot + 255)/256), dim3(256), 0, st, (const double*) a, (double*) b, ntot); });
    printf("copy1  8B flat                     %8.3f ms %8.1f GB/s\n", ms, 2.0*ntot*8*1e-6/ms);
    ms = time_it(st, reps, [&]{ hipLaunchKernelGGL((k_copy1<double>), dim3(8192), dim3(256), 0, st, (const double*) a, (double*) b, ntot); });
    printf("copy1  8B gridstride 8192x256      %8.3f ms %8.1f GB/s\n", ms, 2.0*ntot*8*1e-6/ms);
  }

#define RUN(NAME, KERN, VEC, BS) do { \
    unsigned nblk = (unsigned) ((i1 - i0 + (long long) (BS)*(VEC) - 1)/((long long) (BS)*(VEC))); \
    unsigned grid = ((nblk + 7u)/8u)*8u; \
    double ms = time_it(st, reps, [&]{ hipLaunchKernelGGL(KERN, dim3(grid), dim3(BS), 0, st, (const double*) a, b, nsite, i0, i1, nblk, off); }); \
    printf("%-34s %8.3f ms %8.1f GB/s\n", NAME, ms, gb19/ms*1e3); } while (0)

  RUN("soa19 x1 aligned        b256 xcd", (k_soa19<1,false,false,true>), 1, 256);
  RUN("soa19 x1 shift          b256 xcd", (k_soa19<1,true,false,true>), 1, 256);
  RUN("soa19 x1 shift          b256 rr ", (k_soa19<1,true,false,false>), 1, 256);
  RUN("soa19 x1 shift nt       b256 xcd", (k_soa19<1,true,true,true>), 1, 256);
  RUN("soa19 x2 aligned        b256 xcd", (k_soa19<2,false,false,true>), 2, 256);
  RUN("soa19 x2 shift          b256 xcd", (k_soa19<2,true,false,true>), 2, 256);
  RUN("soa19 x2 shift          b256 rr ", (k_soa19<2,true,false,false>), 2, 256);
  RUN("soa19 x2 shift nt       b256 xcd", (k_soa19<2,true,true,true>), 2, 256);
  RUN("soa19 x2 shift          b128 xcd", (k_soa19<2,true,false,true>), 2, 128);
  RUN("soa19 x2 shift          b512 xcd", (k_soa19<2,true,false,true>), 2, 512);
  RUN("soa19 x1 shift          b128 xcd", (k_soa19<1,true,false,true>), 1, 128);
  RUN("soa19 x1 shift          b512 xcd", (k_soa19<1,true,false,true>), 1, 512);
  RUN("soa19 x1 shift          b1024 xcd", (k_soa19<1,true,false,true>), 1, 1024);
  RUN("soa19ls x1 shift        b256 xcd", (k_soa19_ls<1,true,false,true>), 1, 256);
  RUN("soa19ls x1 shift nt     b256 xcd", (k_soa19_ls<1,true,true,true>), 1, 256);
  RUN("soa19ls x2 shift        b256 xcd", (k_soa19_ls<2,true,false,true>), 2, 256);
  RUN("soa19ls x2 shift nt     b256 xcd", (k_soa19_ls<2,true,true,true>), 2, 256);
  RUN("soa19ls x2 shift        b128 xcd", (k_soa19_ls<2,true,false,true>), 2, 128);
  RUN("soa19ls x2 shift        b256 rr ", (k_soa19_ls<2,true,false,false>), 2, 256);
  RUN("soa19ls x1 shift        b256 rr ", (k_soa19_ls<1,true,false,false>), 1, 256);

  {
    unsigned nblk = (unsigned) ((i1 - i0 + 255)/256);
    unsigned grid = ((nblk + 7u)/8u)*8u;
    double ms = time_it(st, reps, [&]{ hipLaunchKernelGGL((k_soa19_decode<true>), dim3(grid), dim3(256), 0, st, (const double*) a, b, nsite, (int) i0, (int) i1, nblk, off, (int) strx, (int) stry, 1, 256, 256); });
    printf("%-34s %8.3f ms %8.1f GB/s\n", "soa19 x1 shift decode+mask b256 xcd", ms, gb19/ms*1e3);
  }
  CHECK(hipFree(a)); CHECK(hipFree(b));
  return 0;
}
