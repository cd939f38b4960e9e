// tools/probe_handover.hip -- R&D micro-benchmark (not part of the product).
//
// What does a cross-stream dependency cost on MI355X when the event it waits
// for completed long ago? The slab step (lbmi_fused_step) runs a ~110 us
// interior launch on the compute stream, a ~20 us boundary launch on a second
// stream, and joins them every step; the kernel trace shows ~14 us between
// the end of one interior launch and the start of the next, against ~2 us for
// launches that follow each other in one stream. Patterns, per step:
//   A  S: big                                           (one stream)
//   B  S: record e0; big; wait e1      B: wait e0; small; record e1
//   C  as B, small FIRST in host order (boundary before interior)
//   D  S: big; small                                    (serial, one stream)
//   E  S: record e0; big               B: wait e0; small; record e1
//      and S waits e1 of the PREVIOUS step before its record (one step late)
//   F  as B, events created with hipEventDisableTiming
//   G  S: record e0; big                                (the record alone)
//   H  S: record e0; big               B: wait e0; small          (S never waits)
//   I  S: big; wait e1                 B: small; record e1        (B never waits)
//   J  as B with hipStreamWriteValue32 / hipStreamWaitValue32 on signal memory
//      in place of the events
//   K  as B, the small kernel launched with a hipExtLaunch stop event (no record call)
//   L  pattern B, 10 steps captured into ONE hipGraph (fork / join per step), launched 40 times
//   M  pattern D (serial), 10 steps in one graph
// big = copy of `nbig` MB, small = copy of nbig/15.
//
// Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/probe_handover.hip -o tools/probe_handover

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdint>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { \
  printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

__global__ __launch_bounds__(256)
void k_copy(const double * __restrict__ a, double * __restrict__ b, size_t n) {
  size_t i = (size_t) blockIdx.x*256 + threadIdx.x;
  if (i < n) b[i] = a[i] + 1.0;
}

static void copy(hipStream_t s, const double * a, double * b, size_t n) {
  hipLaunchKernelGGL(k_copy, dim3((unsigned) ((n + 255)/256)), dim3(256), 0, s, a, b, n);
}

int main(int argc, char ** argv) {
  const size_t mb = (argc > 1) ? (size_t) atoi(argv[1]) : 300;     // read + written by `big`: 2 x mb
  const int steps = 400;
  const size_t n = (mb << 20)/8, ns = n/15;
  double *a, *b, *c, *d;
  CHECK(hipMalloc(&a, n*8)); CHECK(hipMalloc(&b, n*8));
  CHECK(hipMalloc(&c, ns*8)); CHECK(hipMalloc(&d, ns*8));
  CHECK(hipMemset(a, 0, n*8)); CHECK(hipMemset(c, 0, ns*8));
  hipStream_t S, B;
  CHECK(hipStreamCreateWithFlags(&S, hipStreamNonBlocking));
  CHECK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
  hipEvent_t e0, e1, e1p, f0, f1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); CHECK(hipEventCreate(&e1p));
  CHECK(hipEventCreateWithFlags(&f0, hipEventDisableTiming));
  CHECK(hipEventCreateWithFlags(&f1, hipEventDisableTiming));

  uint32_t * flag = nullptr;
  bool haveflag = (hipExtMallocWithFlags((void **) &flag, 64, hipMallocSignalMemory) == hipSuccess);
  if (haveflag) CHECK(hipMemset(flag, 0, 64));
  uint32_t seq = 0;
  for (int pat = 0; pat < 13; pat++) {
    if (pat == 9 && !haveflag) { printf("pattern J: no signal memory\n"); continue; }
    if (pat >= 11) {
      /* graphs: capture 10 steps on S (B forked from it and joined), launch 40 times */
      hipGraph_t g; hipGraphExec_t ge;
      CHECK(hipStreamBeginCapture(S, hipStreamCaptureModeGlobal));
      for (int s = 0; s < 10; s++) {
	double * x = (s & 1) ? b : a, * y = (s & 1) ? a : b;
	if (pat == 11) {
	  CHECK(hipEventRecord(f0, S));
	  copy(S, x, y, n);
	  CHECK(hipStreamWaitEvent(B, f0, 0));
	  copy(B, c, d, ns);
	  CHECK(hipEventRecord(f1, B));
	  CHECK(hipStreamWaitEvent(S, f1, 0));
	}
	else {
	  copy(S, x, y, n);
	  copy(S, c, d, ns);
	}
      }
      CHECK(hipStreamEndCapture(S, &g));
      CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      for (int rep = 0; rep < 2; rep++) {
	CHECK(hipDeviceSynchronize());
	auto t0 = std::chrono::steady_clock::now();
	for (int k = 0; k < 40; k++) CHECK(hipGraphLaunch(ge, S));
	auto t1 = std::chrono::steady_clock::now();
	CHECK(hipDeviceSynchronize());
	auto t2 = std::chrono::steady_clock::now();
	if (rep == 1) {
	  printf("pattern %c  %4zu MB: issued %.2f us/step, finished %.2f us/step\n", 'A' + pat, mb,
		 1e6*std::chrono::duration<double>(t1 - t0).count()/400,
		 1e6*std::chrono::duration<double>(t2 - t0).count()/400);
	  fflush(stdout);
	}
      }
      CHECK(hipGraphExecDestroy(ge));
      CHECK(hipGraphDestroy(g));
      continue;
    }
    for (int rep = 0; rep < 2; rep++) {
      CHECK(hipDeviceSynchronize());
      auto t0 = std::chrono::steady_clock::now();
      for (int s = 0; s < steps; s++) {
	double * x = (s & 1) ? b : a, * y = (s & 1) ? a : b;
	switch (pat) {
	case 0:
	  copy(S, x, y, n);
	  break;
	case 1:
	  CHECK(hipEventRecord(e0, S));
	  copy(S, x, y, n);
	  CHECK(hipStreamWaitEvent(B, e0, 0));
	  copy(B, c, d, ns);
	  CHECK(hipEventRecord(e1, B));
	  CHECK(hipStreamWaitEvent(S, e1, 0));
	  break;
	case 2:
	  CHECK(hipEventRecord(e0, S));
	  CHECK(hipStreamWaitEvent(B, e0, 0));
	  copy(B, c, d, ns);
	  CHECK(hipEventRecord(e1, B));
	  copy(S, x, y, n);
	  CHECK(hipStreamWaitEvent(S, e1, 0));
	  break;
	case 3:
	  copy(S, x, y, n);
	  copy(S, c, d, ns);
	  break;
	case 4:
	  if (s > 0) CHECK(hipStreamWaitEvent(S, (s & 1) ? e1 : e1p, 0));
	  CHECK(hipEventRecord(e0, S));
	  copy(S, x, y, n);
	  CHECK(hipStreamWaitEvent(B, e0, 0));
	  copy(B, c, d, ns);
	  CHECK(hipEventRecord((s & 1) ? e1p : e1, B));
	  break;
	case 6:
	  CHECK(hipEventRecord(e0, S));
	  copy(S, x, y, n);
	  break;
	case 7:
	  CHECK(hipEventRecord(e0, S));
	  copy(S, x, y, n);
	  CHECK(hipStreamWaitEvent(B, e0, 0));
	  copy(B, c, d, ns);
	  break;
	case 8:
	  copy(S, x, y, n);
	  copy(B, c, d, ns);
	  CHECK(hipEventRecord(e1, B));
	  CHECK(hipStreamWaitEvent(S, e1, 0));
	  break;
	case 9:
	  seq += 1;
	  CHECK(hipStreamWriteValue32(S, flag, seq, 0));
	  copy(S, x, y, n);
	  CHECK(hipStreamWaitValue32(B, flag, seq, hipStreamWaitValueGte, 0xffffffffu));
	  copy(B, c, d, ns);
	  CHECK(hipStreamWriteValue32(B, flag + 8, seq, 0));
	  CHECK(hipStreamWaitValue32(S, flag + 8, seq, hipStreamWaitValueGte, 0xffffffffu));
	  break;
	case 10:
	  CHECK(hipEventRecord(f0, S));
	  copy(S, x, y, n);
	  CHECK(hipStreamWaitEvent(B, f0, 0));
	  hipExtLaunchKernelGGL(k_copy, dim3((unsigned) ((ns + 255)/256)), dim3(256), 0, B,
				nullptr, f1, 0, (const double *) c, d, ns);
	  CHECK(hipStreamWaitEvent(S, f1, 0));
	  break;
	case 5:
	  CHECK(hipEventRecord(f0, S));
	  copy(S, x, y, n);
	  CHECK(hipStreamWaitEvent(B, f0, 0));
	  copy(B, c, d, ns);
	  CHECK(hipEventRecord(f1, B));
	  CHECK(hipStreamWaitEvent(S, f1, 0));
	  break;
	}
      }
      auto t1 = std::chrono::steady_clock::now();
      CHECK(hipDeviceSynchronize());
      auto t2 = std::chrono::steady_clock::now();
      if (rep == 1) {
	printf("pattern %c  %4zu MB: issued %.2f us/step, finished %.2f us/step\n", 'A' + pat, mb,
	       1e6*std::chrono::duration<double>(t1 - t0).count()/steps,
	       1e6*std::chrono::duration<double>(t2 - t0).count()/steps);
	fflush(stdout);
      }
    }
  }
  return 0;
}
