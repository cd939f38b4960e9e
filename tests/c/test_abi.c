/*
 * tests/c/test_abi.c -- the C-ABI used from ANSI C, the way Ludwig's host
 * code (or the shim of INTEGRATION.md) uses it: no Python, no torch, no C++.
 *
 * Builds a uniform-flow D3Q19 state (the reference's lb_init_uniform,
 * distribution_rt.c:504-533, i.e. the regression case serial-dist-3du),
 * runs N steps of lb_collide / lb_halo / lb_propagation in EAGER, FUSED,
 * INPLACE and FUSED_HALO mode and checks the conserved quantities the
 * reference prints for that case (tests/regression/d3q19-short/serial-dist-3du.log):
 *   [rho] 32768.00 1.00000000000 ...   momentum 6.5536e+01 9.8304e+01 1.31072e+02
 *
 * Build + run (on a MI355X): see tests/test_gpu_c_abi.py.
 * Exit code 0 = pass.
 */

#define __HIP_PLATFORM_AMD__ 1

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "lbmi.h"

#define CHECK(call)							\
  do {									\
    int rc_ = (call);							\
    if (rc_ != 0) {							\
      fprintf(stderr, "FAIL %s:%d: %s -> %d (%s)\n", __FILE__, __LINE__, \
	      #call, rc_, lbmi_last_error());				\
      return 1;								\
    }									\
  } while (0)

#define HIPC(call)							\
  do {									\
    hipError_t e_ = (call);						\
    if (e_ != hipSuccess) {						\
      fprintf(stderr, "FAIL %s:%d: %s: %s\n", __FILE__, __LINE__, #call, \
	      hipGetErrorString(e_));					\
      return 1;								\
    }									\
  } while (0)

static int run_mode(int mode, const char * name) {

  enum {N = 32, NH = 1, NALL = N + 2*NH, NVEL = 19, NSTEPS = 10};
  const double u0[3] = {0.002, 0.003, 0.004};
  const size_t nsite = (size_t) NALL*NALL*NALL;

  lbmi_options_t opts;
  lbmi_t * lb = NULL;
  lbmi_hydro_t hydro;
  int8_t cv[NVEL][3];
  double wv[NVEL], na[NVEL];
  double * ma = (double *) malloc(sizeof(double)*NVEL*NVEL);
  double * fh = (double *) calloc(nsite*NVEL, sizeof(double));
  double * f = NULL, * fprime = NULL, * rho = NULL, * u = NULL, * force = NULL;
  double out[9];
  double zero3[3] = {0.0, 0.0, 0.0};

  CHECK(lbmi_model(NVEL, &cv[0][0], wv, na, ma));

  /* second-order equilibrium, model.c:915-941 */
  for (int p = 0; p < NVEL; p++) {
    double udotc = 0.0, sdotq = 0.0;
    for (int a = 0; a < 3; a++) {
      udotc += u0[a]*cv[p][a];
      for (int b = 0; b < 3; b++) {
	sdotq += (cv[p][a]*cv[p][b] - (a == b)/3.0)*u0[a]*u0[b];
      }
    }
    for (int i = NH; i < NH + N; i++)
      for (int j = NH; j < NH + N; j++)
	for (int k = NH; k < NH + N; k++)
	  fh[nsite*p + ((size_t) i*NALL + j)*NALL + k]
	    = 1.0*wv[p]*(1.0 + 3.0*udotc + 4.5*sdotq);
  }

  CHECK(lbmi_options_default(&opts));
  opts.nvel = NVEL;
  opts.nlocal[0] = N; opts.nlocal[1] = N; opts.nlocal[2] = N;
  opts.nhalo = NH;
  opts.mode = mode;
  opts.halo_scheme = LBMI_HALO_FULL;
  CHECK(lbmi_create(&opts, &lb));
  CHECK(lbmi_set_relaxation(lb, LBMI_RELAXATION_M10, 1.0, 0.1, 0.1));

  /* caller-owned device arrays, as lb->target->f etc. are in Ludwig */
  HIPC(hipMalloc((void **) &f, sizeof(double)*nsite*NVEL));
  HIPC(hipMalloc((void **) &fprime, sizeof(double)*nsite*NVEL));
  HIPC(hipMalloc((void **) &rho, sizeof(double)*nsite));
  HIPC(hipMalloc((void **) &u, sizeof(double)*nsite*3));
  HIPC(hipMalloc((void **) &force, sizeof(double)*nsite*3));
  HIPC(hipMemset(fprime, 0, sizeof(double)*nsite*NVEL));
  CHECK(lbmi_lb_bind(lb, f, fprime));
  CHECK(lbmi_lb_memcpy_h2d(lb, fh));

  memset(&hydro, 0, sizeof(hydro));
  hydro.force = force; hydro.status = NULL; hydro.rho = rho; hydro.u = u;

  for (int n = 0; n < NSTEPS; n++) {
    CHECK(lbmi_hydro_field_set(lb, force, 3, zero3));    /* hydro_f_zero */
    CHECK(lbmi_hydro_field_set(lb, u, 3, zero3));        /* hydro_u_zero */
    CHECK(lbmi_lb_collide(lb, &hydro));
    CHECK(lbmi_lb_halo(lb));
    CHECK(lbmi_lb_propagation(lb));
  }
  CHECK(lbmi_lb_moments(lb, NULL, out));

  {
    const double gref[3] = {6.5536000e+01, 9.8304000e+01, 1.3107200e+02};
    int bad = 0;
    double mean = out[1]/out[0];
    double var = fabs(out[2]/out[0] - mean*mean);
    if (out[0] != 32768.0) bad = 1;
    if (fabs(out[1] - 32768.00) > 0.005) bad = 1;
    if (fabs(mean - 1.0) > 0.5e-11) bad = 1;
    if (var > 1.0e-12) bad = 1;
    if (fabs(out[3] - 1.0) > 0.5e-11 || fabs(out[4] - 1.0) > 0.5e-11) bad = 1;
    for (int a = 0; a < 3; a++) {
      if (fabs(out[5 + a] - gref[a]) > 0.5e-6*gref[a]*1e-1) bad = 1;
      if (fabs(out[5 + a] - gref[a])/gref[a] > 1.0e-12) bad = 1;
    }
    printf("%-8s rho %.2f mean %.11f var %.3e min %.11f max %.11f "
	   "g %.7e %.7e %.7e  %s\n", name, out[1], mean, var, out[3], out[4],
	   out[5], out[6], out[7], bad ? "FAIL" : "ok");
    if (bad) return 1;
  }

  CHECK(lbmi_free(lb));
  HIPC(hipFree(f)); HIPC(hipFree(fprime)); HIPC(hipFree(rho));
  HIPC(hipFree(u)); HIPC(hipFree(force));
  free(fh); free(ma);

  return 0;
}

int main(void) {
  if (run_mode(LBMI_MODE_EAGER, "eager")) return 1;
  if (run_mode(LBMI_MODE_FUSED, "fused")) return 1;
  if (run_mode(LBMI_MODE_INPLACE, "inplace")) return 1;
  if (run_mode(LBMI_MODE_FUSED_HALO, "fused_halo")) return 1;
  printf("C-ABI test passed\n");
  return 0;
}
