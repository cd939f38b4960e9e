/*****************************************************************************
 *
 *  unit_main.c
 *
 *  TEST INFRASTRUCTURE ONLY (oracle side). Never linked into the product.
 *
 *  Runs the reference's OWN unit-test suites for the objects on and around
 *  the hot path -- compiled by oracle/Makefile from tests/unit/ *.c where
 *  they lie under /root/reference, for its HIP target on gfx950 -- twice:
 *
 *      unit_hip_d3q19        against the reference as it is
 *      unit_hip_d3q19_shim   against the same objects with
 *                            integration/ludwig_shim.c bound in + liblbmi.so
 *
 *  so that lb_halo / lb_propagation / lb_memcpy / lb_io_* (test_model.c,
 *  test_halo.c, test_prop.c), wall_* (test_wall.c), hydro_* (test_hydro.c),
 *  field_halo, field_grad_compute (test_field.c, test_field_grad.c) of the
 *  binding face the checks the reference's authors wrote for the originals.
 *  (The reference's tests.c runs all of its 100 suites; this is its list cut
 *  to the suites that reach a bound symbol, plus the model tables.)
 *
 *  Each suite prints "PASS     ./unit/test_xxx" when it returns; a failed
 *  check aborts (assert / test_assert), so a missing PASS line is a failure.
 *
 *  usage: unit_hip_d3q19[_shim] [suite ...]      (no argument: all)
 *
 *****************************************************************************/

#include <assert.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <mpi.h>

#include "target.h"

int test_lb_d3q19_suite(void);
int test_lb_d3q27_suite(void);
int test_lb_model_suite(void);
int test_model_suite(void);
int test_halo_suite(void);
int test_lb_prop_suite(void);
int test_lb_bc_inflow_rhou_suite(void);
int test_lb_bc_outflow_rhou_suite(void);
int test_wall_suite(void);
int test_hydro_suite(void);
int test_field_suite(void);
int test_field_grad_suite(void);
int test_map_suite(void);
int test_le_suite(void);
int test_phi_ch_suite(void);

/* tests.h: the check used inside the suites (the reference defines it next
 * to its main, tests.c; this one prints and aborts on either side) */

__host__ __device__ void test_assert_info(const int lvalue, int line,
					  const char * file) {
  if (!lvalue) {
    printf("Line %d file %s Failed test assertion\n", line, file);
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_trap();
#else
    fflush(stdout);
    abort();
#endif
  }
}

typedef struct { const char * name; int (* run)(void); } suite_t;

__host__ int main(int argc, char ** argv) {

  const suite_t suite[] = {
    {"lb_d3q19", test_lb_d3q19_suite}, {"lb_d3q27", test_lb_d3q27_suite},
    {"lb_model", test_lb_model_suite}, {"model", test_model_suite},
    {"halo", test_halo_suite}, {"prop", test_lb_prop_suite},
    {"lb_bc_inflow_rhou", test_lb_bc_inflow_rhou_suite},
    {"lb_bc_outflow_rhou", test_lb_bc_outflow_rhou_suite},
    {"wall", test_wall_suite}, {"hydro", test_hydro_suite},
    {"field", test_field_suite}, {"field_grad", test_field_grad_suite},
    {"map", test_map_suite}, {"le", test_le_suite}, {"phi_ch", test_phi_ch_suite}};
  const int nsuite = (int) (sizeof(suite)/sizeof(suite[0]));

  MPI_Init(&argc, &argv);

  for (int n = 0; n < nsuite; n++) {
    int wanted = (argc == 1);
    for (int a = 1; a < argc; a++) {
      if (strcmp(argv[a], suite[n].name) == 0) wanted = 1;
    }
    if (!wanted) continue;
    printf("RUN      %s\n", suite[n].name);
    fflush(stdout);
    suite[n].run();
    printf("DONE     %s\n", suite[n].name);
    fflush(stdout);
  }

  MPI_Finalize();

  return 0;
}
