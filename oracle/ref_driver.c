/*****************************************************************************
 *
 *  ref_driver.c
 *
 *  TEST INFRASTRUCTURE ONLY (oracle side). Never linked into the product.
 *
 *  A small driver (our code) that links against the *compiled reference*
 *  (objects built by oracle/Makefile from the sources where they lie under
 *  /root/reference; outputs only in oracle/_ref/) and exercises the hot path
 *  exactly as ludwig.c does per time step:
 *
 *      lb_collide()      reference src/collision.c:143   (ludwig.c:802)
 *      lb_halo()         reference src/model.c:553       (ludwig.c:816)
 *      lb_propagation()  reference src/propagation.c:43  (ludwig.c:860)
 *
 *  It has two jobs:
 *   (1) "dump" mode: write golden vectors (raw little-endian doubles) for
 *       the parity tests: tests/golden/ is produced from these by
 *       oracle/make_golden.py;
 *   (2) "time" mode: time the reference CPU path (the cpu_baseline leg of
 *       bench.py, kind "reference").
 *
 *  NVEL is a compile-time choice in the reference (-D_D3Q19_ / -D_D3Q27_),
 *  so one binary per model is built. Memory order is -DADDR_SOA.
 *
 *  Usage:
 *    ref_driver dump <prefix> nx ny nz nhalo scheme eta zeta fx fy fz \
 *               fieldforce solid nsteps [visc [kt ghosts]]
 *               (kt > 0: isothermal fluctuations, noise->on[NOISE_RHO];
 *               the generator states before and after are dumped as well)
 *    ref_driver time nx ny nz scheme eta zeta nsteps
 *    ref_driver fe <prefix> nx ny nz a b kappa mobility [gradnpt advorder]
 *               (symmetric free energy: field_halo, field_grad_compute with
 *               the 7- or 27-point stencil, pth_stress_compute,
 *               pth_force_fluid_driver, then phi_cahn_hilliard with advection
 *               of order 1..4 in a prescribed velocity field; nhalo = 2)
 *
 *    ref_driver binary <prefix> nx ny nz a b kappa mobility eta zeta fx nsteps [kt]
 *               (two-distribution symmetric_lb step, collision.c:610-1027;
 *               kt > 0: with isothermal fluctuations, generator states dumped)
 *    ref_driver relax <prefix> nx ny nz a b kappa mobility eta zeta fx nsteps
 *               (ONE distribution, fe->use_stress_relaxation: collision.c:413)
 *    ref_driver wall <prefix> nx ny nz bx by bz uboty utopy solid nsteps
 *               [sbx sby sbz stx sty stz]     (partial slip, wall.c:285-316)
 *               (flat walls: lb_collide, lb_halo, wall_bbl, lb_propagation)
 *    ref_driver io <dir> nx ny nz timestep      (lb_io_write into <dir>)
 *    ref_driver ioread <dir> nx ny nz timestep  (lb_io_read from <dir>)
 *
 *  scheme: m10 | bgk | trt
 *
 *****************************************************************************/

#include <assert.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "pe.h"
#include "coords.h"
#include "physics.h"
#include "lb_data.h"
#include "collision.h"
#include "propagation.h"
#include "hydro.h"
#include "map.h"
#include "noise.h"
#include "leesedwards.h"
#include "field.h"
#include "field_grad.h"
#include "gradient_3d_7pt_fluid.h"
#include "gradient_3d_27pt_fluid.h"
#include "symmetric.h"
#include "phi_force_stress.h"
#include "phi_force_colloid.h"
#include "phi_cahn_hilliard.h"
#include "advection.h"
#include "phi_lb_coupler.h"
#include "wall.h"

#define PI_ 3.14159265358979323846

typedef struct {
  int ntotal[3];
  int nhalo;
  lb_relaxation_enum_t nrelax;
  double eta, zeta;
  double fbody[3];
  int fieldforce;       /* per-site force field on/off */
  int solid;            /* a block of MAP_BOUNDARY sites on/off */
  int nsteps;
  int visc;             /* viscosity model: local eta from hydro->eta */
  double kt;            /* > 0: isothermal fluctuations at this temperature */
  int ghosts;           /* 0: ghost modes off (lb_collision_ghost_modes_off) */
} case_t;

static uint32_t lcg_state = 12345u;

static double lcg_uniform(void) {
  lcg_state = 1664525u*lcg_state + 1013904223u;
  return lcg_state/4294967296.0;
}

static double wtime(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1.0e-9*ts.tv_nsec;
}

static void dump_i32(const char * prefix, const char * name, const int * a,
		     size_t n);

static void dump(const char * prefix, const char * name, const double * a,
		 size_t n) {
  char fn[1024];
  FILE * fp = NULL;
  snprintf(fn, sizeof(fn), "%s.%s.f64", prefix, name);
  fp = fopen(fn, "wb");
  if (fp == NULL) { perror(fn); exit(1); }
  if (fwrite(a, sizeof(double), n, fp) != n) { perror(fn); exit(1); }
  fclose(fp);
}

static lb_relaxation_enum_t scheme_from_string(const char * s) {
  if (strcmp(s, "m10") == 0) return LB_RELAXATION_M10;
  if (strcmp(s, "bgk") == 0) return LB_RELAXATION_BGK;
  if (strcmp(s, "trt") == 0) return LB_RELAXATION_TRT;
  fprintf(stderr, "unknown scheme %s\n", s);
  exit(1);
}

/* Synthetic initial condition of SURVEY.md section 8(d): second-order
 * equilibrium of a smooth (rho, u) field times (1 + 1e-3 (r - 1/2)) with r
 * from the 32-bit LCG advanced in (x,y,z,p) order. */

static void init_f(cs_t * cs, lb_t * lb, const case_t * c) {

  int nlocal[3];
  cs_nlocal(cs, nlocal);
  lcg_state = 12345u;

  for (int ic = 1; ic <= nlocal[X]; ic++) {
    for (int jc = 1; jc <= nlocal[Y]; jc++) {
      for (int kc = 1; kc <= nlocal[Z]; kc++) {
	int index = cs_index(cs, ic, jc, kc);
	double x = (ic - 1.0)/c->ntotal[X];
	double y = (jc - 1.0)/c->ntotal[Y];
	double z = (kc - 1.0)/c->ntotal[Z];
	double rho = 1.0 + 0.01*cos(2.0*PI_*(x + y + z));
	double u[3];
	u[X] = 0.01*sin(2.0*PI_*y);
	u[Y] = 0.01*sin(2.0*PI_*z);
	u[Z] = 0.01*sin(2.0*PI_*x);
	lb_1st_moment_equilib_set(lb, index, rho, u);
	for (int p = 0; p < lb->model.nvel; p++) {
	  double f = 0.0;
	  double r = lcg_uniform();
	  lb_f(lb, index, p, LB_RHO, &f);
	  lb_f_set(lb, index, p, LB_RHO, f*(1.0 + 1.0e-3*(r - 0.5)));
	}
      }
    }
  }
}

static void init_force(cs_t * cs, hydro_t * hydro, const case_t * c) {

  int nlocal[3];
  cs_nlocal(cs, nlocal);

  for (int ic = 1; ic <= nlocal[X]; ic++) {
    for (int jc = 1; jc <= nlocal[Y]; jc++) {
      for (int kc = 1; kc <= nlocal[Z]; kc++) {
	int index = cs_index(cs, ic, jc, kc);
	double x = (ic - 1.0)/c->ntotal[X];
	double y = (jc - 1.0)/c->ntotal[Y];
	double z = (kc - 1.0)/c->ntotal[Z];
	double f[3] = {0.0, 0.0, 0.0};
	if (c->fieldforce) {
	  f[X] = 1.0e-5*cos(2.0*PI_*x);
	  f[Y] = 1.0e-5*cos(2.0*PI_*y);
	  f[Z] = 1.0e-5*cos(2.0*PI_*z);
	}
	for (int ia = 0; ia < 3; ia++) {
	  hydro->force->data[addr_rank1(hydro->nsite, 3, index, ia)] = f[ia];
	}
      }
    }
  }
}

/* A 2x2x2 block of MAP_BOUNDARY sites at (2..3, 2..3, 2..3) */

static void init_map(cs_t * cs, map_t * map, const case_t * c) {
  if (c->solid == 0) return;
  for (int ic = 2; ic <= 3; ic++) {
    for (int jc = 2; jc <= 3; jc++) {
      for (int kc = 2; kc <= 3; kc++) {
	map_status_set(map, cs_index(cs, ic, jc, kc), MAP_BOUNDARY);
      }
    }
  }
}

/* Symmetric free-energy force chain of BASELINE config 4 (ludwig.c:563-717):
 * field_halo(phi) -> field_grad_compute (3d_7pt_fluid) -> pth_stress_compute
 * -> pth_force_fluid_driver (what phi_force_calculation runs for
 * FE_FORCE_METHOD_STRESS_DIVERGENCE without walls, phi_force.c:100-108). */

static int run_fe(int argc, char ** argv) {

  const char * prefix = argv[2];
  int ntotal[3] = {atoi(argv[3]), atoi(argv[4]), atoi(argv[5])};
  fe_symm_param_t param = {0};
  pe_t * pe = NULL;
  cs_t * cs = NULL;
  lees_edw_t * le = NULL;
  field_t * phi = NULL;
  field_grad_t * dphi = NULL;
  fe_symm_t * fe = NULL;
  pth_t * pth = NULL;
  hydro_t * hydro = NULL;
  physics_t * phys = NULL;
  int nlocal[3];

  double mobility = atof(argv[9]);
  int gradnpt = (argc == 12) ? atoi(argv[10]) : 7;     /* 7 | 27 */
  int advorder = (argc == 12) ? atoi(argv[11]) : 1;    /* 1 .. 4 */
  phi_ch_t * pch = NULL;

  param.a = atof(argv[6]);
  param.b = atof(argv[7]);
  param.kappa = atof(argv[8]);

  MPI_Init(&argc, &argv);
  pe_create(MPI_COMM_WORLD, PE_QUIET, &pe);
  cs_create(pe, &cs);
  cs_ntotal_set(cs, ntotal);
  cs_nhalo_set(cs, 2);
  cs_init(cs);
  cs_nlocal(cs, nlocal);
  physics_create(pe, &phys);
  physics_mobility_set(phys, mobility);

  {
    lees_edw_options_t opts = {0};
    opts.nplanes = 0;
    lees_edw_create(pe, cs, &opts, &le);
  }
  {
    field_options_t opts = field_options_ndata_nhalo(1, 2);
    field_create(pe, cs, le, "phi", &opts, &phi);
  }
  field_grad_create(pe, phi, 2, &dphi);
  if (gradnpt == 27) field_grad_set(dphi, grad_3d_27pt_fluid_d2, NULL);
  else field_grad_set(dphi, grad_3d_7pt_fluid_d2, NULL);
  fe_symm_create(pe, cs, phi, dphi, &fe);
  fe_symm_param_set(fe, param);
  pth_create(pe, cs, FE_FORCE_METHOD_STRESS_DIVERGENCE, &pth);
  {
    hydro_options_t hopts = hydro_options_nhalo(1);
    hydro_create(pe, cs, le, &hopts, &hydro);
  }

  lcg_state = 12345u;
  for (int ic = 1; ic <= nlocal[X]; ic++) {
    for (int jc = 1; jc <= nlocal[Y]; jc++) {
      for (int kc = 1; kc <= nlocal[Z]; kc++) {
	double x = (ic - 1.0)/ntotal[X];
	double y = (jc - 1.0)/ntotal[Y];
	double z = (kc - 1.0)/ntotal[Z];
	double r = lcg_uniform();
	double v = 0.3*sin(2.0*PI_*x)*cos(2.0*PI_*y) + 0.2*sin(2.0*PI_*z + 1.0)
	  + 0.05*(r - 0.5);
	field_scalar_set(phi, cs_index(cs, ic, jc, kc), v);
      }
    }
  }

  field_halo(phi);
  field_grad_compute(dphi);
  {
    double fzero[3] = {0.0, 0.0, 0.0};
    hydro_f_zero(hydro, fzero);
  }
  pth_stress_compute(pth, (fe_t *) fe);
  pth_force_fluid_driver(pth, hydro);

  {
    size_t ns = (size_t) phi->nsites;
    int nall[3];
    char fn[1024];
    FILE * fp = NULL;
    cs_nall(cs, nall);
    dump(prefix, "phi", phi->data, ns);
    dump(prefix, "grad", dphi->grad, 3*ns);
    dump(prefix, "delsq", dphi->delsq, ns);
    dump(prefix, "stress", pth->str, 9*ns);
    dump(prefix, "force", hydro->force->data, 3*ns);

    /* Cahn-Hilliard update (ludwig.c:759): prescribed velocity field,
     * upwind advection (advection.c:542-640), diffusive flux of mu
     * (phi_cahn_hilliard.c:349-402), forward step (:981-1060) */
    {
      phi_ch_info_t options = {0};
      phi_ch_create(pe, cs, le, &options, &pch);
      advection_order_set(advorder);
      for (int ic = 1; ic <= nlocal[X]; ic++) {
	for (int jc = 1; jc <= nlocal[Y]; jc++) {
	  for (int kc = 1; kc <= nlocal[Z]; kc++) {
	    int index = cs_index(cs, ic, jc, kc);
	    double x = (ic - 1.0)/ntotal[X];
	    double y = (jc - 1.0)/ntotal[Y];
	    double z = (kc - 1.0)/ntotal[Z];
	    double u[3];
	    u[X] = 0.05*sin(2.0*PI_*y) + 0.01*(lcg_uniform() - 0.5);
	    u[Y] = 0.05*sin(2.0*PI_*z) + 0.01*(lcg_uniform() - 0.5);
	    u[Z] = 0.05*sin(2.0*PI_*x) + 0.01*(lcg_uniform() - 0.5);
	    hydro_u_set(hydro, index, u);
	  }
	}
      }
      phi_cahn_hilliard(pch, (fe_t *) fe, phi, hydro, NULL, NULL);
      dump(prefix, "u", hydro->u->data, 3*ns);       /* after hydro_u_halo */
      dump(prefix, "phi_new", phi->data, ns);
      phi_ch_free(pch);
    }
    snprintf(fn, sizeof(fn), "%s.json", prefix);
    fp = fopen(fn, "w");
    fprintf(fp, "{\"nlocal\": [%d, %d, %d], \"nhalo\": 2, \"nall\": [%d, %d, %d],"
	    " \"nsite\": %d, \"a\": %.17g, \"b\": %.17g, \"kappa\": %.17g,"
	    " \"mobility\": %.17g, \"grad_npt\": %d, \"advection_order\": %d,"
	    " \"layout\": \"soa\"}\n",
	    ntotal[X], ntotal[Y], ntotal[Z],
	    nall[X], nall[Y], nall[Z], (int) ns, param.a, param.b, param.kappa,
	    mobility, gradnpt, advorder);
    fclose(fp);
  }

  hydro_free(hydro);
  pth_free(pth);
  fe_symm_free(fe);
  field_grad_free(dphi);
  field_free(phi);
  lees_edw_free(le);
  physics_free(phys);
  cs_free(cs);
  pe_free(pe);
  MPI_Finalize();

  return 0;
}

/*****************************************************************************
 *
 *  run_binary
 *
 *  "binary" mode: ref_driver binary <prefix> nx ny nz a b kappa mobility \
 *                 eta zeta fx nsteps
 *    The two-distribution step of free_energy symmetric_lb (ludwig.c:
 *    558-578, 802-860): phi_lb_to_field, field_halo, field_grad_compute
 *    (27-point), hydro_u_zero, lb_collide -> lb_collision_binary
 *    (collision.c:610-1027), lb_halo, lb_propagation; nhalo = 1.
 *    Dumps f0 (both distributions), and after the first collision phi,
 *    grad, delsq, f_collide, u; f_final after nsteps.
 *
 *****************************************************************************/

static int run_binary(int argc, char ** argv) {

  /* "relax": ONE distribution and fe->use_stress_relaxation = 1, i.e. the
   * single-fluid collision with the symmetric stress in the equilibrium
   * stress (collision.c:413-429, FE_FORCE_METHOD_RELAXATION_SYMM,
   * ludwig.c:1235-1237); phi is a fixed analytic field */
  const int relax = (strcmp(argv[1], "relax") == 0);
  const char * prefix = argv[2];
  case_t c = {0};
  fe_symm_param_t param = {0};
  double mobility = atof(argv[9]);
  double fzero[3] = {0.0, 0.0, 0.0};
  int nsteps = atoi(argv[13]);
  double kt = (argc == 15) ? atof(argv[14]) : 0.0;

  pe_t * pe = NULL;
  cs_t * cs = NULL;
  lees_edw_t * le = NULL;
  physics_t * phys = NULL;
  lb_t * lb = NULL;
  field_t * phi = NULL;
  field_grad_t * dphi = NULL;
  fe_symm_t * fe = NULL;
  hydro_t * hydro = NULL;
  map_t * map = NULL;
  noise_t * noise = NULL;
  int nlocal[3];

  c.ntotal[X] = atoi(argv[3]);
  c.ntotal[Y] = atoi(argv[4]);
  c.ntotal[Z] = atoi(argv[5]);
  c.nhalo = 1;
  param.a = atof(argv[6]);
  param.b = atof(argv[7]);
  param.kappa = atof(argv[8]);
  c.eta = atof(argv[10]);
  c.zeta = atof(argv[11]);
  c.fbody[X] = atof(argv[12]);

  MPI_Init(&argc, &argv);
  pe_create(MPI_COMM_WORLD, PE_QUIET, &pe);
  cs_create(pe, &cs);
  cs_ntotal_set(cs, c.ntotal);
  cs_nhalo_set(cs, c.nhalo);
  cs_init(cs);
  cs_nlocal(cs, nlocal);

  physics_create(pe, &phys);
  physics_rho0_set(phys, 1.0);
  physics_eta_shear_set(phys, c.eta);
  physics_eta_bulk_set(phys, c.zeta);
  physics_fbody_set(phys, c.fbody);
  physics_mobility_set(phys, mobility);

  {
    lees_edw_options_t opts = {0};
    opts.nplanes = 0;
    lees_edw_create(pe, cs, &opts, &le);
  }
  {
    lb_data_options_t opts = lb_data_options_default();
    opts.ndim = NDIM;
    opts.nvel = NVEL;
    opts.ndist = relax ? 1 : 2;
    opts.nrelax = LB_RELAXATION_M10;
    opts.halo = LB_HALO_TARGET;
    lb_data_create(pe, cs, &opts, &lb);
  }
  {
    field_options_t opts = field_options_ndata_nhalo(1, 1);
    field_create(pe, cs, le, "phi", &opts, &phi);
  }
  field_grad_create(pe, phi, 2, &dphi);
  field_grad_set(dphi, grad_3d_27pt_fluid_d2, NULL);
  fe_symm_create(pe, cs, phi, dphi, &fe);
  fe_symm_param_set(fe, param);
  if (relax) fe->super.use_stress_relaxation = 1;
  {
    hydro_options_t hopts = hydro_options_nhalo(1);
    hydro_create(pe, cs, le, &hopts, &hydro);
  }
  map_create(pe, cs, 0, &map);
  noise_create(pe, cs, &noise);
  noise_init(noise, 0);
  if (kt > 0.0) {
    physics_kt_set(phys, kt);
    noise_present_set(noise, NOISE_RHO, 1);
    noise_memcpy(noise, tdpMemcpyHostToDevice);
  }

  init_f(cs, lb, &c);
  init_map(cs, map, &c);
  hydro_f_zero(hydro, fzero);

  /* second distribution: g_p = w_p phi0 (1 + 1e-2 (r - 1/2)) */
  for (int ic = 1; ic <= nlocal[X]; ic++) {
    for (int jc = 1; jc <= nlocal[Y]; jc++) {
      for (int kc = 1; kc <= nlocal[Z]; kc++) {
	int index = cs_index(cs, ic, jc, kc);
	double x = (ic - 1.0)/c.ntotal[X];
	double y = (jc - 1.0)/c.ntotal[Y];
	double z = (kc - 1.0)/c.ntotal[Z];
	double phi0 = 0.4*sin(2.0*PI_*x)*cos(2.0*PI_*y) + 0.3*sin(2.0*PI_*z + 1.0);
	if (relax) {
	  field_scalar_set(phi, index, phi0 + 0.02*(lcg_uniform() - 0.5));
	  continue;
	}
	for (int p = 0; p < lb->model.nvel; p++) {
	  double r = lcg_uniform();
	  lb_f_set(lb, index, p, LB_PHI,
		   lb->model.wv[p]*phi0*(1.0 + 1.0e-2*(r - 0.5)));
	}
      }
    }
  }

  {
    size_t nf = (size_t) lb->nsite*lb->model.nvel*(relax ? 1 : 2);
    size_t ns = (size_t) lb->nsite;
    int nall[3];
    char fn[1024];
    FILE * fp = NULL;

    dump(prefix, "f0", lb->f, nf);
    if (kt > 0.0) {
      dump_i32(prefix, "noise0", (const int *) noise->state,
	       (size_t) NNOISE_STATE*noise->nsites);
    }
    /* device builds (no-ops on the CPU): the state goes over, dumps come back */
    lb_memcpy(lb, tdpMemcpyHostToDevice);
    field_memcpy(phi, tdpMemcpyHostToDevice);
    hydro_memcpy(hydro, tdpMemcpyHostToDevice);
    map_memcpy(map, tdpMemcpyHostToDevice);
    for (int n = 0; n < nsteps; n++) {
      if (!relax) phi_lb_to_field(phi, lb);
      field_halo(phi);
      field_grad_compute(dphi);
      hydro_u_zero(hydro, fzero);
      lb_collide(lb, hydro, map, noise, (fe_t *) fe, NULL);
      if (n == 0) {
	lb_memcpy(lb, tdpMemcpyDeviceToHost);
	field_memcpy(phi, tdpMemcpyDeviceToHost);
	field_grad_memcpy(dphi, tdpMemcpyDeviceToHost);
	hydro_memcpy(hydro, tdpMemcpyDeviceToHost);
	dump(prefix, "phi", phi->data, ns);
	dump(prefix, "grad", dphi->grad, 3*ns);
	dump(prefix, "delsq", dphi->delsq, ns);
	dump(prefix, "f_collide", lb->f, nf);
	dump(prefix, "u", hydro->u->data, 3*ns);
	dump(prefix, "rho", hydro->rho->data, ns);
      }
      lb_halo(lb);
      lb_propagation(lb);
    }
    lb_memcpy(lb, tdpMemcpyDeviceToHost);
    dump(prefix, "f_final", lb->f, nf);
    if (kt > 0.0) {
      noise_memcpy(noise, tdpMemcpyDeviceToHost);
      dump_i32(prefix, "noise_final", (const int *) noise->state,
	       (size_t) NNOISE_STATE*noise->nsites);
    }

    cs_nall(cs, nall);
    snprintf(fn, sizeof(fn), "%s.json", prefix);
    fp = fopen(fn, "w");
    fprintf(fp, "{\"nvel\": %d, \"ndist\": %d, \"nlocal\": [%d, %d, %d],"
	    " \"nhalo\": 1, \"nall\": [%d, %d, %d], \"nsite\": %d,"
	    " \"a\": %.17g, \"b\": %.17g, \"kappa\": %.17g,"
	    " \"mobility\": %.17g, \"eta\": %.17g, \"zeta\": %.17g,"
	    " \"fbody\": [%.17g, 0.0, 0.0], \"nsteps\": %d, \"kt\": %.17g,"
	    " \"layout\": \"soa\"}\n",
	    NVEL, relax ? 1 : 2, c.ntotal[X], c.ntotal[Y], c.ntotal[Z],
	    nall[X], nall[Y], nall[Z], lb->nsite, param.a, param.b,
	    param.kappa, mobility, c.eta, c.zeta, c.fbody[X], nsteps, kt);
    fclose(fp);
  }

  noise_free(noise);
  map_free(map);
  hydro_free(hydro);
  fe_symm_free(fe);
  field_grad_free(dphi);
  field_free(phi);
  lb_free(lb);
  lees_edw_free(le);
  physics_free(phys);
  cs_free(cs);
  pe_free(pe);
  MPI_Finalize();

  return 0;
}

/*****************************************************************************
 *
 *  run_wall
 *
 *  "wall" mode: ref_driver wall <prefix> nx ny nz bx by bz uboty utopy \
 *               solid nsteps
 *    Flat walls in the directions with b? = 1 (wall_commit: wall_init_map,
 *    wall_init_boundaries, wall_init_uw, wall.c:166-186, 381-470, 864-890,
 *    1219-1268) moving with (0, uboty, 0) / (0, utopy, 0), optionally the
 *    MAP_BOUNDARY block of init_map; the step of ludwig.c:802-860 with
 *    walls: lb_collide, lb_halo, wall_bbl (wall.c:960-1107), lb_propagation.
 *    Dumps the status map, the links, f after the first wall_bbl, f_final and
 *    the accumulated wall momentum.
 *
 *****************************************************************************/

static void dump_i32(const char * prefix, const char * name, const int * a,
		     size_t n) {
  char fn[1024];
  FILE * fp = NULL;
  snprintf(fn, sizeof(fn), "%s.%s.i32", prefix, name);
  fp = fopen(fn, "wb");
  if (fp == NULL) { perror(fn); exit(1); }
  if (n > 0 && fwrite(a, sizeof(int), n, fp) != n) { perror(fn); exit(1); }
  fclose(fp);
}

static int run_wall(int argc, char ** argv) {

  const char * prefix = argv[2];
  case_t c = {0};
  int periodic[3];
  wall_param_t wp = {0};
  int nsteps = atoi(argv[12]);
  int colloid = 0;

  pe_t * pe = NULL;
  cs_t * cs = NULL;
  physics_t * phys = NULL;
  lb_t * lb = NULL;
  hydro_t * hydro = NULL;
  map_t * map = NULL;
  noise_t * noise = NULL;
  wall_t * wall = NULL;

  c.ntotal[X] = atoi(argv[3]);
  c.ntotal[Y] = atoi(argv[4]);
  c.ntotal[Z] = atoi(argv[5]);
  c.nhalo = 1;
  c.nrelax = LB_RELAXATION_M10;
  c.eta = 0.1;
  c.zeta = 0.3;
  wp.iswall = 1;
  wp.isboundary[X] = atoi(argv[6]);
  wp.isboundary[Y] = atoi(argv[7]);
  wp.isboundary[Z] = atoi(argv[8]);
  wp.ubot[Y] = atof(argv[9]);
  wp.utop[Y] = atof(argv[10]);
  c.solid = atoi(argv[11]);
  if (c.solid == 2) {
    /* 2: no solid block, but MAP_COLLOID marks on some fluid sites after the
     * links exist (the accounting-only branch, wall.c:1048-1061, 1148-1161) */
    c.solid = 0;
    colloid = 1;
  }
  for (int ia = 0; ia < 3; ia++) periodic[ia] = 1 - wp.isboundary[ia];
  if (argc == 19) {
    /* partial slip: sbot[3] stop[3] (wall_slip, wall.c:285-316) */
    double sbot[3] = {atof(argv[13]), atof(argv[14]), atof(argv[15])};
    double stop[3] = {atof(argv[16]), atof(argv[17]), atof(argv[18])};
    wp.slip = wall_slip(sbot, stop);
    if (!wall_slip_valid(&wp.slip)) {
      fprintf(stderr, "ref_driver: invalid slip fractions\n");
      return 1;
    }
  }

  MPI_Init(&argc, &argv);
  pe_create(MPI_COMM_WORLD, PE_QUIET, &pe);
  cs_create(pe, &cs);
  cs_ntotal_set(cs, c.ntotal);
  cs_nhalo_set(cs, c.nhalo);
  cs_periodicity_set(cs, periodic);
  cs_init(cs);

  physics_create(pe, &phys);
  physics_rho0_set(phys, 1.0);
  physics_eta_shear_set(phys, c.eta);
  physics_eta_bulk_set(phys, c.zeta);

  {
    lb_data_options_t opts = lb_data_options_default();
    opts.ndim = NDIM;
    opts.nvel = NVEL;
    opts.ndist = 1;
    opts.nrelax = c.nrelax;
    opts.halo = LB_HALO_TARGET;
    lb_data_create(pe, cs, &opts, &lb);
  }
  {
    hydro_options_t hopts = hydro_options_nhalo(1);
    hydro_create(pe, cs, NULL, &hopts, &hydro);
  }
  map_create(pe, cs, 0, &map);
  noise_create(pe, cs, &noise);
  noise_init(noise, 0);

  init_f(cs, lb, &c);
  init_map(cs, map, &c);
  wall_create(pe, cs, map, lb, &wall);
  wall_commit(wall, &wp);
  if (colloid) {
    for (int ic = 1; ic <= c.ntotal[X]; ic++) {
      for (int jc = 1; jc <= c.ntotal[Y]; jc++) {
	for (int kc = 1; kc <= c.ntotal[Z]; kc++) {
	  if ((ic + 2*jc + 3*kc) % 5 != 0) continue;
	  map_status_set(map, cs_index(cs, ic, jc, kc), MAP_COLLOID);
	}
      }
    }
  }

  {
    size_t nf = (size_t) lb->nsite*lb->model.nvel;
    size_t ns = (size_t) lb->nsite;
    int nall[3];
    double fnet[3] = {0.0, 0.0, 0.0};
    int * status = (int *) calloc(ns, sizeof(int));
    char fn[1024];
    FILE * fp = NULL;

    for (size_t i = 0; i < ns; i++) {
      int st = 0;
      map_status(map, (int) i, &st);
      status[i] = st;
    }
    dump_i32(prefix, "status", status, ns);
    dump_i32(prefix, "linki", wall->linki, wall->nlink);
    dump_i32(prefix, "linkj", wall->linkj, wall->nlink);
    dump_i32(prefix, "linkp", wall->linkp, wall->nlink);
    dump_i32(prefix, "linku", wall->linku, wall->nlink);
    if (wp.slip.active) {
      int * tmp = (int *) calloc((size_t) wall->nlink + 1, sizeof(int));
      dump_i32(prefix, "linkk", wall->linkk, wall->nlink);
      for (int n = 0; n < wall->nlink; n++) tmp[n] = wall->linkq[n];
      dump_i32(prefix, "linkq", tmp, wall->nlink);
      for (int n = 0; n < wall->nlink; n++) tmp[n] = wall->links[n];
      dump_i32(prefix, "links", tmp, wall->nlink);
      free(tmp);
    }
    dump(prefix, "f0", lb->f, nf);

    /* device builds (no-ops on the CPU): the state goes over, dumps come back */
    lb_memcpy(lb, tdpMemcpyHostToDevice);
    hydro_memcpy(hydro, tdpMemcpyHostToDevice);
    map_memcpy(map, tdpMemcpyHostToDevice);

    for (int n = 0; n < nsteps; n++) {
      lb_collide(lb, hydro, map, noise, NULL, NULL);
      lb_halo(lb);
      wall_bbl(wall);
      if (n == 0) {
	lb_memcpy(lb, tdpMemcpyDeviceToHost);
	dump(prefix, "f_bbl", lb->f, nf);
      }
      lb_propagation(lb);
    }
    lb_memcpy(lb, tdpMemcpyDeviceToHost);
    dump(prefix, "f_final", lb->f, nf);
    wall_momentum(wall, fnet);

    cs_nall(cs, nall);
    snprintf(fn, sizeof(fn), "%s.json", prefix);
    fp = fopen(fn, "w");
    fprintf(fp, "{\"nvel\": %d, \"nlocal\": [%d, %d, %d], \"nhalo\": 1,"
	    " \"nall\": [%d, %d, %d], \"nsite\": %d, \"scheme\": 0,"
	    " \"eta\": %.17g, \"zeta\": %.17g, \"rho0\": 1.0,"
	    " \"fbody\": [0.0, 0.0, 0.0], \"isboundary\": [%d, %d, %d],"
	    " \"ubot\": [0.0, %.17g, 0.0], \"utop\": [0.0, %.17g, 0.0],"
	    " \"solid\": %d, \"nlink\": %d, \"nsteps\": %d,"
	    " \"slip\": %d, \"sbot\": [%.17g, %.17g, %.17g],"
	    " \"stop\": [%.17g, %.17g, %.17g],"
	    " \"fnet\": [%.17g, %.17g, %.17g], \"layout\": \"soa\"}\n",
	    NVEL, c.ntotal[X], c.ntotal[Y], c.ntotal[Z],
	    nall[X], nall[Y], nall[Z], lb->nsite, c.eta, c.zeta,
	    wp.isboundary[X], wp.isboundary[Y], wp.isboundary[Z],
	    wp.ubot[Y], wp.utop[Y], 2*colloid + c.solid, wall->nlink, nsteps,
	    wp.slip.active, wp.slip.s[WALL_SLIP_XBOT], wp.slip.s[WALL_SLIP_YBOT],
	    wp.slip.s[WALL_SLIP_ZBOT], wp.slip.s[WALL_SLIP_XTOP],
	    wp.slip.s[WALL_SLIP_YTOP], wp.slip.s[WALL_SLIP_ZTOP],
	    fnet[X], fnet[Y], fnet[Z]);
    fclose(fp);
    free(status);
  }

  wall_free(wall);
  noise_free(noise);
  map_free(map);
  hydro_free(hydro);
  lb_free(lb);
  physics_free(phys);
  cs_free(cs);
  pe_free(pe);
  MPI_Finalize();

  return 0;
}

/*****************************************************************************
 *
 *  run_io
 *
 *  "io" mode: ref_driver io <dir> nx ny nz timestep [ndist [ascii|single|wallz]]
 *    lb_io_write (model.c:1568-1614, MPI-IO mode; "single": the old-style
 *    i/o of a run without i/o keys, io_harness.c) of the synthetic state in
 *    <dir>: the metadata file dist.json and the data file; also <dir>/f0.f64.
 *  "ioread" mode: ref_driver ioread <dir> nx ny nz timestep
 *    lb_io_read (model.c:1622-1649) of the data file found in <dir>, then
 *    <dir>/readback.f64 = the whole f array.
 *
 *****************************************************************************/

static int run_io(int argc, char ** argv) {

  int reading = (strcmp(argv[1], "ioread") == 0);
  const char * dir = argv[2];
  case_t c = {0};
  int timestep = atoi(argv[6]);
  pe_t * pe = NULL;
  cs_t * cs = NULL;
  lb_t * lb = NULL;

  c.ntotal[X] = atoi(argv[3]);
  c.ntotal[Y] = atoi(argv[4]);
  c.ntotal[Z] = atoi(argv[5]);
  c.nhalo = 1;

  if (chdir(dir) != 0) {
    fprintf(stderr, "cannot chdir to %s\n", dir);
    return 1;
  }

  MPI_Init(&argc, &argv);
  pe_create(MPI_COMM_WORLD, PE_QUIET, &pe);
  cs_create(pe, &cs);
  cs_ntotal_set(cs, c.ntotal);
  cs_nhalo_set(cs, c.nhalo);
  if (argc == 9 && strcmp(argv[8], "wallz") == 0) {
    /* as a run with walls in z has it: the metadata prints the periodicity */
    int periodic[3] = {1, 1, 0};
    cs_periodicity_set(cs, periodic);
  }
  cs_init(cs);
  {
    lb_data_options_t opts = lb_data_options_default();
    opts.ndim = NDIM;
    opts.nvel = NVEL;
    opts.ndist = (argc >= 8) ? atoi(argv[7]) : 1;
    opts.iodata.input = io_options_with_mode(IO_MODE_MPIIO);
    opts.iodata.output = io_options_with_mode(IO_MODE_MPIIO);
    if (argc == 9 && strcmp(argv[8], "ascii") == 0) {
      /* distribution_io_format ascii (io_options_rt.c): text records */
      opts.iodata.input.iorformat = IO_RECORD_ASCII;
      opts.iodata.output.iorformat = IO_RECORD_ASCII;
    }
    if (argc == 9 && strcmp(argv[8], "single") == 0) {
      /* what a run with no i/o key in its input has (io_options_default():
       * IO_MODE_SINGLE): the old-style files of io_harness.c */
      opts.iodata.input = io_options_default();
      opts.iodata.output = io_options_default();
    }
    lb_data_create(pe, cs, &opts, &lb);
    if (argc == 9 && strcmp(argv[8], "single") == 0) {
      /* as distribution_rt.c:258-273 */
      io_info_t * io_info = NULL;
      io_info_args_t tmp = io_info_args_default();
      io_info_create(pe, cs, &tmp, &io_info);
      io_info_metadata_filestub_set(io_info, "dist");
      lb_io_info_set(lb, io_info, IO_FORMAT_BINARY, IO_FORMAT_BINARY);
    }
  }
  {
    io_event_t event = {0};
    size_t nf = (size_t) lb->nsite*lb->model.nvel*lb->ndist;
    if (reading) {
      lb_io_read(lb, timestep, &event);
      /* (ludwig.c:333-340 copies the host array to the device afterwards;
       * here the host array is what is dumped) */
      dump("readback", "f", lb->f, nf);
    }
    else {
      init_f(cs, lb, &c);
      if (lb->ndist == 2) {
	/* something to tell the second distribution from the first */
	for (int ic = 1; ic <= c.ntotal[X]; ic++) {
	  for (int jc = 1; jc <= c.ntotal[Y]; jc++) {
	    for (int kc = 1; kc <= c.ntotal[Z]; kc++) {
	      int index = cs_index(cs, ic, jc, kc);
	      for (int p = 0; p < lb->model.nvel; p++) {
		lb_f_set(lb, index, p, LB_PHI, 0.1*(lcg_uniform() - 0.5));
	      }
	    }
	  }
	}
      }
      dump("written", "f0", lb->f, nf);
      lb_memcpy(lb, tdpMemcpyHostToDevice);       /* device builds */
      lb_io_write(lb, timestep, &event);
    }
  }
  lb_free(lb);
  cs_free(cs);
  pe_free(pe);
  MPI_Finalize();

  return 0;
}

int main(int argc, char ** argv) {

  int timing = 0;
  char prefix[512] = "";
  case_t c = {0};

  pe_t * pe = NULL;
  cs_t * cs = NULL;
  physics_t * phys = NULL;
  lb_t * lb = NULL;
  hydro_t * hydro = NULL;
  map_t * map = NULL;
  noise_t * noise = NULL;

  if ((argc == 10 || argc == 12) && strcmp(argv[1], "fe") == 0) {
    return run_fe(argc, argv);
  }
  if ((argc == 13 || argc == 19) && strcmp(argv[1], "wall") == 0) {
    return run_wall(argc, argv);
  }
  if ((argc == 14 || argc == 15) && (strcmp(argv[1], "binary") == 0 ||
		     strcmp(argv[1], "relax") == 0)) {
    return run_binary(argc, argv);
  }
  if ((argc >= 7 && argc <= 9) && (strcmp(argv[1], "io") == 0 ||
				   strcmp(argv[1], "ioread") == 0)) {
    return run_io(argc, argv);
  }

  if (argc >= 2 && strcmp(argv[1], "dump") == 0 &&
      (argc == 16 || argc == 17 || argc == 19)) {
    int a = 2;
    strncpy(prefix, argv[a++], sizeof(prefix) - 1);
    c.ntotal[X] = atoi(argv[a++]);
    c.ntotal[Y] = atoi(argv[a++]);
    c.ntotal[Z] = atoi(argv[a++]);
    c.nhalo = atoi(argv[a++]);
    c.nrelax = scheme_from_string(argv[a++]);
    c.eta = atof(argv[a++]);
    c.zeta = atof(argv[a++]);
    c.fbody[X] = atof(argv[a++]);
    c.fbody[Y] = atof(argv[a++]);
    c.fbody[Z] = atof(argv[a++]);
    c.fieldforce = atoi(argv[a++]);
    c.solid = atoi(argv[a++]);
    c.nsteps = atoi(argv[a++]);
    c.ghosts = 1;
    if (argc >= 17) c.visc = atoi(argv[a++]);
    if (argc == 19) {
      c.kt = atof(argv[a++]);
      c.ghosts = atoi(argv[a++]);
    }
  }
  else if (argc >= 2 && strcmp(argv[1], "time") == 0 && argc == 9) {
    int a = 2;
    timing = 1;
    c.ntotal[X] = atoi(argv[a++]);
    c.ntotal[Y] = atoi(argv[a++]);
    c.ntotal[Z] = atoi(argv[a++]);
    c.nhalo = 1;
    c.nrelax = scheme_from_string(argv[a++]);
    c.eta = atof(argv[a++]);
    c.zeta = atof(argv[a++]);
    c.nsteps = atoi(argv[a++]);
  }
  else {
    fprintf(stderr, "usage: see header of ref_driver.c\n");
    return 1;
  }

  MPI_Init(&argc, &argv);

  pe_create(MPI_COMM_WORLD, PE_QUIET, &pe);
  cs_create(pe, &cs);
  cs_ntotal_set(cs, c.ntotal);
  cs_nhalo_set(cs, c.nhalo);
  cs_init(cs);

  physics_create(pe, &phys);
  physics_rho0_set(phys, 1.0);
  physics_eta_shear_set(phys, c.eta);
  physics_eta_bulk_set(phys, c.zeta);
  physics_fbody_set(phys, c.fbody);

  {
    lb_data_options_t opts = lb_data_options_default();
    opts.ndim = NDIM;
    opts.nvel = NVEL;
    opts.ndist = 1;
    opts.nrelax = c.nrelax;
    opts.halo = LB_HALO_TARGET;
    lb_data_create(pe, cs, &opts, &lb);
  }
  {
    hydro_options_t hopts = hydro_options_nhalo(1);
    hydro_create(pe, cs, NULL, &hopts, &hydro);
  }
  map_create(pe, cs, 0, &map);
  noise_create(pe, cs, &noise);
  noise_init(noise, 0);
  assert(noise->on[NOISE_RHO] == 0);
  if (c.kt > 0.0) {
    /* isothermal_fluctuations on; temperature kt (ludwig.c, physics_rt.c) */
    physics_kt_set(phys, c.kt);
    noise_present_set(noise, NOISE_RHO, 1);
    if (c.ghosts == 0) lb_collision_ghost_modes_off(lb);
    noise_memcpy(noise, tdpMemcpyHostToDevice);
  }

  init_f(cs, lb, &c);
  init_force(cs, hydro, &c);
  init_map(cs, map, &c);

  if (c.visc) {
    /* what a viscosity model (visc_t::update) would leave in hydro->eta;
     * lb_collide only tests the visc pointer (collision.c:1947) */
    int nlocal[3];
    cs_nlocal(cs, nlocal);
    for (int ic = 1; ic <= nlocal[X]; ic++) {
      for (int jc = 1; jc <= nlocal[Y]; jc++) {
	for (int kc = 1; kc <= nlocal[Z]; kc++) {
	  double x = (ic - 1.0)/c.ntotal[X];
	  double z = (kc - 1.0)/c.ntotal[Z];
	  double eta = c.eta*(1.0 + 0.5*sin(2.0*PI_*x)*cos(2.0*PI_*z))
	    + 0.01*c.eta*(lcg_uniform() - 0.5);
	  hydro->eta->data[addr_rank0(hydro->nsite, cs_index(cs, ic, jc, kc))] = eta;
	}
      }
    }
  }

  /* Device builds (the reference's HIP target, Makefile target "hip"): the
   * initial state goes to the device, and every dump below comes back from
   * it first. On the CPU builds target == host and these calls do nothing. */
  lb_memcpy(lb, tdpMemcpyHostToDevice);
  hydro_memcpy(hydro, tdpMemcpyHostToDevice);
  map_memcpy(map, tdpMemcpyHostToDevice);

  {
    size_t nf = (size_t) lb->nsite*lb->model.nvel;
    size_t ns = (size_t) lb->nsite;
    visc_t * visc = c.visc ? (visc_t *) hydro : NULL;   /* non-NULL flag */

    if (timing == 0) {

      dump(prefix, "f0", lb->f, nf);
      dump(prefix, "force", hydro->force->data, 3*ns);
      if (c.visc) dump(prefix, "eta", hydro->eta->data, ns);
      if (c.kt > 0.0) {
	dump_i32(prefix, "noise0", (const int *) noise->state,
		 (size_t) NNOISE_STATE*noise->nsites);
      }

      for (int n = 0; n < c.nsteps; n++) {
	lb_collide(lb, hydro, map, noise, NULL, visc);
	if (n == 0) {
	  lb_memcpy(lb, tdpMemcpyDeviceToHost);
	  hydro_memcpy(hydro, tdpMemcpyDeviceToHost);
	  dump(prefix, "f_collide", lb->f, nf);
	  dump(prefix, "rho", hydro->rho->data, ns);
	  dump(prefix, "u", hydro->u->data, 3*ns);
	}
	lb_halo(lb);
	if (n == 0) {
	  lb_memcpy(lb, tdpMemcpyDeviceToHost);
	  dump(prefix, "f_halo", lb->f, nf);
	}
	lb_propagation(lb);
	if (n == 0) {
	  lb_memcpy(lb, tdpMemcpyDeviceToHost);
	  dump(prefix, "f_prop", lb->f, nf);
	}
      }
      lb_memcpy(lb, tdpMemcpyDeviceToHost);
      dump(prefix, "f_final", lb->f, nf);
      if (c.kt > 0.0) {
	noise_memcpy(noise, tdpMemcpyDeviceToHost);
	dump_i32(prefix, "noise_final", (const int *) noise->state,
		 (size_t) NNOISE_STATE*noise->nsites);
      }
      {
	/* The on-disk record stream of lb_io_aggr_pack (model.c:1479-1510):
	 * lb_write_buf for every interior site in (ic, jc, kc) order */
	int nlocal[3];
	size_t nrec;
	size_t szrec = (size_t) lb->model.nvel*sizeof(double);
	char * buf = NULL;
	size_t ib = 0;
	cs_nlocal(cs, nlocal);
	nrec = (size_t) nlocal[X]*nlocal[Y]*nlocal[Z];
	buf = (char *) malloc(nrec*szrec);
	assert(buf);
	for (int ic = 1; ic <= nlocal[X]; ic++) {
	  for (int jc = 1; jc <= nlocal[Y]; jc++) {
	    for (int kc = 1; kc <= nlocal[Z]; kc++) {
	      lb_write_buf(lb, cs_index(cs, ic, jc, kc), buf + ib*szrec);
	      ib += 1;
	    }
	  }
	}
	dump(prefix, "records", (const double *) buf, nrec*lb->model.nvel);
	free(buf);
      }
      {
	/* Header: everything a reader needs; consumed by make_golden.py */
	char fn[1024];
	FILE * fp = NULL;
	int nall[3];
	cs_nall(cs, nall);
	snprintf(fn, sizeof(fn), "%s.json", prefix);
	fp = fopen(fn, "w");
	fprintf(fp, "{\"nvel\": %d, \"nlocal\": [%d, %d, %d], \"nhalo\": %d,"
		" \"nall\": [%d, %d, %d], \"nsite\": %d, \"scheme\": %d,"
		" \"eta\": %.17g, \"zeta\": %.17g, \"rho0\": 1.0,"
		" \"fbody\": [%.17g, %.17g, %.17g], \"fieldforce\": %d,"
		" \"solid\": %d, \"nsteps\": %d, \"visc\": %d,"
		" \"kt\": %.17g, \"ghosts\": %d,"
		" \"layout\": \"soa\"}\n",
		NVEL, c.ntotal[X], c.ntotal[Y], c.ntotal[Z], c.nhalo,
		nall[X], nall[Y], nall[Z], lb->nsite, (int) c.nrelax,
		c.eta, c.zeta, c.fbody[X], c.fbody[Y], c.fbody[Z],
		c.fieldforce, c.solid, c.nsteps, c.visc, c.kt, c.ghosts);
	fclose(fp);
      }
    }
    else {
      double tc = 0.0, th = 0.0, tp = 0.0, t0, t1;
      double sites = 1.0*c.ntotal[X]*c.ntotal[Y]*c.ntotal[Z];
      int nthreads = 1;
#ifdef _OPENMP
      nthreads = omp_get_max_threads();
#endif
      /* the hydro housekeeping of ludwig.c's loop belongs to the step: the
       * force field is zeroed at its start (ludwig.c:537), the velocity before
       * the collision (ludwig.c:791) */
      const double zero3[3] = {0.0, 0.0, 0.0};
      /* one untimed warm-up step */
      hydro_f_zero(hydro, zero3);
      hydro_u_zero(hydro, zero3);
      lb_collide(lb, hydro, map, noise, NULL, NULL);
      lb_halo(lb);
      lb_propagation(lb);
      tdpDeviceSynchronize();
      double tall = wtime();
      for (int n = 0; n < c.nsteps; n++) {
	t0 = wtime();
	hydro_f_zero(hydro, zero3);
	hydro_u_zero(hydro, zero3);
	lb_collide(lb, hydro, map, noise, NULL, NULL);
	t1 = wtime(); tc += t1 - t0; t0 = t1;
	lb_halo(lb);
	t1 = wtime(); th += t1 - t0; t0 = t1;
	lb_propagation(lb);
	t1 = wtime(); tp += t1 - t0;
      }
      /* (device builds: a binding may leave work in flight; the stage times
       * are then times to enqueue, the total is what counts) */
      tdpDeviceSynchronize();
      tall = wtime() - tall;
      printf("{\"nvel\": %d, \"nlocal\": [%d, %d, %d], \"steps\": %d,"
	     " \"threads\": %d, \"t_collide\": %.6f, \"t_halo\": %.6f,"
	     " \"t_propagation\": %.6f, \"t_total\": %.6f, \"mlups\": %.4f}\n",
	     NVEL, c.ntotal[X], c.ntotal[Y], c.ntotal[Z], c.nsteps, nthreads,
	     tc, th, tp, tall, 1.0e-6*sites*c.nsteps/tall);
    }
  }

  noise_free(noise);
  map_free(map);
  hydro_free(hydro);
  lb_free(lb);
  physics_free(phys);
  cs_free(cs);
  pe_free(pe);

  MPI_Finalize();

  return 0;
}
