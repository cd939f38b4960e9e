/*****************************************************************************
 *
 *  ludwig_shim.c
 *
 *  The reference-side binding of liblbmi: what a Ludwig maintainer adds to
 *  src/ so that the main loop (ludwig.c:802-860) runs the MI355X-native LB
 *  step unchanged. It defines, with the reference's own signatures,
 *
 *      lb_collide()      collision.h:27    (replaces collision.c:143-163)
 *      lb_halo()         lb_data.h:159     (replaces model.c:553-563)
 *      lb_halo_swap()    lb_data.h:160     (replaces model.c:565-595)
 *      lb_propagation()  propagation.h:21  (replaces propagation.c:43-98)
 *      lb_memcpy()       lb_data.h:156     (wraps  model.c:228-266)
 *      lb_io_write(), lb_io_read()  lb_data.h  (replace model.c:1568-1649 in
 *                        MPI-IO mode, one file, binary records)
 *      wall_bbl()        wall.h:99         (replaces wall.c:960-989, slip included)
 *      phi_lb_to_field() phi_lb_coupler.h  (replaces phi_lb_coupler.c:39-64)
 *      hydro_u_zero(), hydro_f_zero()  hydro.h:64-65 (hydro.c:279-330)
 *      hydro_memcpy()    hydro.h:56        (the original, after rho and u that a
 *                        collision has left on demand have been formed; a copy to
 *                        the device is reported to the library as a foreign write)
 *      lb_bc_inflow_rhou_create(), lb_bc_outflow_rhou_create()
 *                        lb_bc_inflow_rhou.h:39, lb_bc_outflow_rhou.h:36: the
 *                        originals, counted (a run with open boundaries keeps the
 *                        reference's state between lb_halo and lb_propagation)
 *      field_halo()      field.h:96        (field.c; FIELD_HALO_TARGET only)
  *      field_grad_compute() field_grad.h:49 (3d_7pt_fluid / 3d_27pt_fluid d2)
 *      wall_set_wall_distributions() wall.h:100, bounce_back_on_links()
 *                        bbl.h:26: the originals, after making sure that the
 *                        distributions they work on are the reference's
 *      lb_free(), field_free(), map_free(), wall_free(): the originals, after
 *                        forgetting what this file remembers of the object
 *      stats_distribution_print(), stats_distribution_momentum()
 *                        stats_distribution.h (stats_distribution.c:55-139): the
 *                        density statistics and the Kahan-summed momentum
 *      cahn_hilliard_stats(), cahn_hilliard_stats_time0()
 *                        cahn_hilliard_stats.h: the statistics of phi
 *      phi_force_calculation() phi_force.h, phi_cahn_hilliard()
 *                        phi_cahn_hilliard.h: the symmetric free energy in
 *                        the plain periodic fluid case (LBMI_FE=0: never)
 *
 *  by unpacking lb_t / hydro_t / map_t and calling the C-ABI of
 *  include/lbmi.h. The other contents of collision.c / model.c /
 *  propagation.c (relaxation setters, moments, I/O, ...) stay as they are:
 *  the three reference files are compiled with
 *
 *      -Dlb_collide=lb_collide_ref -Dlb_halo=lb_halo_ref
 *      -Dlb_halo_swap=lb_halo_swap_ref -Dlb_propagation=lb_propagation_ref
 *      -Dlb_memcpy=lb_memcpy_ref -Dlb_io_write=lb_io_write_ref
 *      -Dlb_io_read=lb_io_read_ref -Dlb_free=lb_free_ref
 *
 *  (and wall.c with -Dwall_bbl=wall_bbl_ref -Dwall_set_wall_distributions=
 *  wall_set_wall_distributions_ref -Dwall_free=wall_free_ref, bbl.c with
 *  -Dbounce_back_on_links=bounce_back_on_links_ref, map.c with -Dmap_free=
 *  map_free_ref, phi_lb_coupler.c with
 *  -Dphi_lb_to_field=phi_lb_to_field_ref, hydro.c with -Dhydro_u_zero=
 *  hydro_u_zero_ref -Dhydro_f_zero=hydro_f_zero_ref -Dhydro_memcpy=
 *  hydro_memcpy_ref -Dhydro_free=hydro_free_ref, field.c with -Dfield_halo=
 *  field_halo_ref -Dfield_free=field_free_ref, field_grad.c with
 *  -Dfield_grad_compute=field_grad_compute_ref -Dfield_grad_free=
 *  field_grad_free_ref, stats_distribution.c with
 *  -Dstats_distribution_print=stats_distribution_print_ref
 *  -Dstats_distribution_momentum=stats_distribution_momentum_ref, phi_force.c
 *  with -Dphi_force_calculation=phi_force_calculation_ref, phi_cahn_hilliard.c
 *  with -Dphi_cahn_hilliard=phi_cahn_hilliard_ref, cahn_hilliard_stats.c with
 *  -Dcahn_hilliard_stats=cahn_hilliard_stats_ref -Dcahn_hilliard_stats_time0=
 *  cahn_hilliard_stats_time0_ref) so that their originals remain
 *  available as fall-backs (colloids, Lees-Edwards, host halo
 *  schemes, noise; lb_bc_inflow_rhou.c with -Dlb_bc_inflow_rhou_create=
 *  lb_bc_inflow_rhou_create_ref, lb_bc_outflow_rhou.c likewise), and this file
 *  is compiled
 *  with the same -D_D3Q19_|-D_D3Q27_ -DADDR_SOA as the rest of libludwig.a
 *  and linked with -llbmi. See INTEGRATION.md.
 *
 *  No environment variable is needed: a run starts in LBMI_MODE_FUSED with
 *  hydro->rho, u on demand and the binding demotes it per consumer it detects
 *  (shim_handle, lb_collide, shim_needs_canonical_f below).
 *
 *  This file is compile-checked against the reference headers by
 *  __graft_entry__.build() when /root/reference is present (gcc
 *  -fsyntax-only); it cannot be linked or run in this repository because
 *  the reference tree is not part of it.
 *
 *****************************************************************************/

#include <assert.h>
#include <float.h>
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#include "pe.h"
#include "coords.h"
#include "physics.h"
#include "lb_data.h"
#include "collision.h"
#include "propagation.h"
#include "hydro.h"
#include "map.h"
#include "noise.h"
#include "util.h"
#include "field.h"
#include "field_grad.h"
#include "symmetric.h"
#include "wall.h"
#include "phi_lb_coupler.h"
#include "gradient_3d_7pt_fluid.h"
#include "gradient_3d_27pt_fluid.h"
#include "leesedwards.h"
#include "bbl.h"
#include "colloids.h"
#include "stats_distribution.h"
#include "phi_force.h"
#include "phi_force_stress.h"
#include "phi_cahn_hilliard.h"
#include "advection.h"
#include "cahn_hilliard_stats.h"
#include "fe_force_method.h"
#include "lb_bc_inflow_rhou.h"
#include "lb_bc_outflow_rhou.h"
#include "control.h"

#include "lbmi.h"

/* The originals, renamed on the command line (see above) */
int lb_collide_ref(lb_t * lb, hydro_t * hydro, map_t * map, noise_t * noise,
		   fe_t * fe, visc_t * visc);
int lb_halo_swap_ref(lb_t * lb, lb_halo_enum_t flag);
int lb_propagation_ref(lb_t * lb);
int lb_memcpy_ref(lb_t * lb, tdpMemcpyKind flag);
int lb_io_write_ref(lb_t * lb, int timestep, io_event_t * event);
int lb_io_read_ref(lb_t * lb, int timestep, io_event_t * event);
int wall_bbl_ref(wall_t * wall);
int wall_set_wall_distributions_ref(wall_t * wall);
int wall_free_ref(wall_t * wall);
int bounce_back_on_links_ref(bbl_t * bbl, lb_t * lb, wall_t * wall,
			     colloids_info_t * cinfo);
int lb_free_ref(lb_t * lb);
int field_free_ref(field_t * obj);
int map_free_ref(map_t * obj);
int noise_free_ref(noise_t * obj);
int stats_distribution_print_ref(lb_t * lb, map_t * map);
int stats_distribution_momentum_ref(lb_t * lb, map_t * map, double g[3]);
int phi_force_calculation_ref(pe_t * pe, cs_t * cs, lees_edw_t * le,
			      wall_t * wall, pth_t * pth, fe_t * fe,
			      map_t * map, field_t * phi, hydro_t * hydro);
int phi_cahn_hilliard_ref(phi_ch_t * pch, fe_t * fe, field_t * phi,
			  hydro_t * hydro, map_t * map, noise_t * noise);
int cahn_hilliard_stats_ref(phi_ch_t * pch, field_t * phi, map_t * map);
int cahn_hilliard_stats_time0_ref(phi_ch_t * pch, field_t * phi, map_t * map);
int phi_lb_to_field_ref(field_t * phi, lb_t * lb);
int hydro_u_zero_ref(hydro_t * hydro, const double uzero[3]);
int hydro_f_zero_ref(hydro_t * hydro, const double fzero[3]);
int hydro_memcpy_ref(hydro_t * hydro, tdpMemcpyKind flag);
int hydro_free_ref(hydro_t * hydro);
int field_halo_ref(field_t * field);
int lb_bc_inflow_rhou_create_ref(pe_t * pe, cs_t * cs,
				 const lb_bc_inflow_opts_t * options,
				 lb_bc_inflow_rhou_t ** inflow);
int lb_bc_outflow_rhou_create_ref(pe_t * pe, cs_t * cs,
				  const lb_bc_outflow_opts_t * options,
				  lb_bc_outflow_rhou_t ** outflow);
int field_grad_compute_ref(field_grad_t * fgrad);
void field_grad_free_ref(field_grad_t * obj);

/* LBMI_REPORT=1: at exit, how many calls of every bound symbol the library
 * took and how many went to the original (what the binding did in this run;
 * tools/unit_suites.sh and tools/regression_sweep.py read it). */

enum {S_LB_COLLIDE, S_LB_HALO_SWAP, S_LB_PROPAGATION, S_LB_IO_WRITE, S_LB_IO_READ, S_PHI_LB_TO_FIELD, S_WALL_BBL, S_HYDRO_U_ZERO, S_HYDRO_F_ZERO, S_FIELD_HALO, S_FIELD_GRAD_COMPUTE, S_STATS_DISTRIBUTION_PRINT, S_STATS_DISTRIBUTION_MOMENTUM, S_PHI_FORCE_CALCULATION, S_PHI_CAHN_HILLIARD, S_CAHN_HILLIARD_STATS, S_CAHN_HILLIARD_STATS_TIME0, S_NSYM};

static const char * const shim_sym_[S_NSYM] = {
  "lb_collide",
  "lb_halo_swap",
  "lb_propagation",
  "lb_io_write",
  "lb_io_read",
  "phi_lb_to_field",
  "wall_bbl",
  "hydro_u_zero",
  "hydro_f_zero",
  "field_halo",
  "field_grad_compute",
  "stats_distribution_print",
  "stats_distribution_momentum",
  "phi_force_calculation",
  "phi_cahn_hilliard",
  "cahn_hilliard_stats",
  "cahn_hilliard_stats_time0"};
static long shim_calls_[S_NSYM][2];              /* [.][1]: liblbmi, [.][0]: original */

static const char * shim_mode_name(int mode);
static void shim_report_policy(void);

static void shim_report(void) {
  shim_report_policy();
  fprintf(stderr, "liblbmi report: %-28s %10s %10s\n", "symbol", "liblbmi", "original");
  for (int n = 0; n < S_NSYM; n++) {
    if (shim_calls_[n][0] + shim_calls_[n][1] == 0) continue;
    fprintf(stderr, "liblbmi report: %-28s %10ld %10ld\n", shim_sym_[n],
	    shim_calls_[n][1], shim_calls_[n][0]);
  }
}

static void shim_pointers_settle(void);

static void shim_note(int sym, int bound) {
  static int first = 1;
  if (first) {
    const char * e = getenv("LBMI_REPORT");
    first = 0;
    if (e != NULL && e[0] == '1') atexit(shim_report);
  }
  shim_calls_[sym][bound] += 1;
  /* an original is about to run: it may launch one of the reference's
   * kernels that take f from lb->target (bbl.c:277, wall.c:913, 980,
   * phi_lb_coupler.c:59, stats_distribution.c:240) */
  if (!bound) shim_pointers_settle();
}

/* (a function-like macro is not expanded inside its own expansion: every
 * call of an original below is counted on its way) */
/* three originals that are not counted but read lb->target all the same
 * (model.c:228-266, wall.c:913, bbl.c:277) */
#define lb_memcpy_ref(...) (shim_pointers_settle(), lb_memcpy_ref(__VA_ARGS__))
#define wall_set_wall_distributions_ref(...) (shim_pointers_settle(), wall_set_wall_distributions_ref(__VA_ARGS__))
#define bounce_back_on_links_ref(...) (shim_pointers_settle(), bounce_back_on_links_ref(__VA_ARGS__))
#define lb_collide_ref(...) (shim_note(S_LB_COLLIDE, 0), lb_collide_ref(__VA_ARGS__))
#define lb_halo_swap_ref(...) (shim_note(S_LB_HALO_SWAP, 0), lb_halo_swap_ref(__VA_ARGS__))
#define lb_propagation_ref(...) (shim_note(S_LB_PROPAGATION, 0), lb_propagation_ref(__VA_ARGS__))
#define lb_io_write_ref(...) (shim_note(S_LB_IO_WRITE, 0), lb_io_write_ref(__VA_ARGS__))
#define lb_io_read_ref(...) (shim_note(S_LB_IO_READ, 0), lb_io_read_ref(__VA_ARGS__))
#define phi_lb_to_field_ref(...) (shim_note(S_PHI_LB_TO_FIELD, 0), phi_lb_to_field_ref(__VA_ARGS__))
#define wall_bbl_ref(...) (shim_note(S_WALL_BBL, 0), wall_bbl_ref(__VA_ARGS__))
#define hydro_u_zero_ref(...) (shim_note(S_HYDRO_U_ZERO, 0), hydro_u_zero_ref(__VA_ARGS__))
#define hydro_f_zero_ref(...) (shim_note(S_HYDRO_F_ZERO, 0), hydro_f_zero_ref(__VA_ARGS__))
#define field_halo_ref(...) (shim_note(S_FIELD_HALO, 0), field_halo_ref(__VA_ARGS__))
#define field_grad_compute_ref(...) (shim_note(S_FIELD_GRAD_COMPUTE, 0), field_grad_compute_ref(__VA_ARGS__))
#define stats_distribution_print_ref(...) (shim_note(S_STATS_DISTRIBUTION_PRINT, 0), stats_distribution_print_ref(__VA_ARGS__))
#define stats_distribution_momentum_ref(...) (shim_note(S_STATS_DISTRIBUTION_MOMENTUM, 0), stats_distribution_momentum_ref(__VA_ARGS__))
#define phi_force_calculation_ref(...) (shim_note(S_PHI_FORCE_CALCULATION, 0), phi_force_calculation_ref(__VA_ARGS__))
#define phi_cahn_hilliard_ref(...) (shim_note(S_PHI_CAHN_HILLIARD, 0), phi_cahn_hilliard_ref(__VA_ARGS__))
#define cahn_hilliard_stats_ref(...) (shim_note(S_CAHN_HILLIARD_STATS, 0), cahn_hilliard_stats_ref(__VA_ARGS__))
#define cahn_hilliard_stats_time0_ref(...) (shim_note(S_CAHN_HILLIARD_STATS_TIME0, 0), cahn_hilliard_stats_time0_ref(__VA_ARGS__))

/* One liblbmi handle per lb_t (Ludwig has one lb_t per rank) */

typedef struct shim_s {
  lb_t * lb;
  lbmi_t * h;
  int mode;                       /* lbmi_mode_t in use */
  wall_t * wall;                  /* whose links the handle holds a copy of */
  int wall_nlink;
  int colloids;                   /* bounce_back_on_links has seen colloids */
  int ncollide;                   /* bound collisions so far */
  int nlazy;                      /* ... of which left rho, u on demand */
  int nlazy_rho;                  /* ... of which left rho alone on demand */
  int automode;                   /* LBMI_MODE unset: fused until a consumer of
				     the state between lb_collide and
				     lb_propagation shows up */
  int param_valid;                /* param_committed is what the device has */
  lb_collide_param_t param_committed;
} shim_t;

static shim_t shim_;              /* zero: no handle, LBMI_MODE_EAGER */
static int shim_openbc_ = 0;      /* open-boundary objects created (below) */

/* The free-energy sector folded into the collision (the symmetric free energy
 * in the plain periodic one-rank case, BASELINE config 4): ludwig.c:563-802
 * calls field_halo(phi), field_grad_compute, phi_force_calculation,
 * phi_cahn_hilliard, hydro_u_zero and lb_collide one after the other. When
 * the step before went through all of them in the case the binding covers,
 * the calls of this step are only noted as they come, and lb_collide runs
 * lbmi_symmetric_lb_collide: ONE kernel in which the thread that collides a
 * site evaluates its force and its Cahn-Hilliard update from the phi around
 * it (no gradient arrays, no force array, no halo swaps of phi and u).
 * Whatever comes out of turn -- any other bound symbol, another object, a
 * call that does not qualify -- first runs what was noted, in order
 * (shim_fuse_flush). Not on a step that reports or writes anything
 * (control.c's is_*_step): there every array is what the reference has.
 * What stays behind on a folded step: the halo of phi, the gradient arrays
 * and hydro->force are not refreshed (their readers are the calls folded),
 * phi and hydro->u hold the new values at the interior sites. */

enum {FUSE_NONE = 0, FUSE_HALO, FUSE_GRAD, FUSE_FORCE, FUSE_CH, FUSE_UZERO};

typedef struct shim_fuse_s {
  int armed;                      /* the step before qualified */
  int stage;                      /* what has been noted this step */
  int cand;                       /* normal step: bit 0 force, bit 1 CH bound */
  /* the objects of the sequence */
  field_t * phi;
  field_grad_t * fgrad;
  hydro_t * hydro;
  phi_ch_t * pch;
  fe_t * fe;
  /* the arguments of the noted calls */
  pe_t * pe; cs_t * cs; lees_edw_t * le; wall_t * wall; pth_t * pth;
  map_t * map_force; map_t * map_ch; noise_t * noise;
  double uzero[3];
  /* candidates seen on a normal step */
  field_grad_t * seen_fgrad;
  /* The second arrays of phi and u (device). The kernel reads one pair and
   * writes the other; from one folded step to the next the new values STAY
   * where they were written (phi_in_q, u_in_b: the latest interior values are
   * in the second array, not in the reference's) and go back into the
   * reference's arrays when somebody else is about to look
   * (shim_fuse_settle: every bound symbol that is not the next one of the
   * sequence and not one of the per-step LB calls). */
  double * phinew; double * uprev; size_t sites;
  int phi_in_q; int u_in_b;
  int nfused;                     /* collisions that took the sector along */
} shim_fuse_t;

static shim_fuse_t fuse_;

static void shim_fuse_flush(void);      /* run what was noted; settle */
static void shim_fuse_flush_lb(void);   /* run what was noted (a per-step LB call: phi, u not looked at) */
static int shim_fuse_wanted(void);

static const char * shim_mode_name(int mode) {
  if (mode == LBMI_MODE_FUSED) return "fused";
  if (mode == LBMI_MODE_FUSED_HALO) return "halo";
  if (mode == LBMI_MODE_INPLACE) return "inplace";
  return "eager";
}

/* LBMI_REPORT=1: how the run ended up being executed (what the policy of
 * shim_handle / lb_collide below made of it) */

static shim_t shim_ended_;        /* what shim_ was when lb_free dropped the handle */

static void shim_report_policy(void) {
  const shim_t * sh = (shim_.h != NULL) ? &shim_ : &shim_ended_;
  if (sh->lb == NULL) return;
  fprintf(stderr, "liblbmi report: execution mode %s (%s); rho, u on demand in "
	  "%d of %d collisions, rho alone in %d; free-energy sector folded into "
	  "%d\n", shim_mode_name(sh->mode),
	  sh->automode ? "chosen by the binding" : "LBMI_MODE",
	  sh->nlazy, sh->ncollide, sh->nlazy_rho, fuse_.nfused);
}

#define SHIM_CHECK(lb, call)						\
  do {									\
    int ifail_ = (call);						\
    if (ifail_ != 0) {							\
      pe_fatal((lb)->pe, "liblbmi: %s (%s:%d)\n", lbmi_last_error(),	\
	       __FILE__, __LINE__);					\
    }									\
  } while (0)

/* Can liblbmi take this lb_t? SoA build, device halo scheme, any Cartesian
 * decomposition whose ranks are numbered as MPI_Cart_create numbers them
 * without reordering: slabs (grid N_1_1, 1_N_1 or 1_1_N, coords_rt.c:46-47)
 * with their own fused steps, grids of ranks in more than one direction as
 * the reference's sequence of halo passes. Anything else uses the originals. */

static int shim_slab_dim(lb_t * lb, int * dim) {
  int cartsz[3];
  int ndec = 0;
  cs_cartsz(lb->cs, cartsz);
  *dim = X;
  for (int d = 0; d < 3; d++) {
    if (cartsz[d] > 1) {
      ndec += 1;
      *dim = d;
    }
  }
  return (ndec <= 1);
}

/* More than one direction decomposed (the reference's own choice for N ranks
 * without a `grid` key is MPI_Dims_create's, coords.c:520-560: 2_2_2 for
 * eight): liblbmi numbers the ranks of a grid as MPI_Cart_create does when
 * it does not reorder them -- which is what this checks, together with the
 * six neighbours, before such a run is taken on. */

static int shim_cart_general_ok(lb_t * lb) {
  int cartsz[3], coords[3];
  int rank = -1;
  MPI_Comm comm;
  cs_cartsz(lb->cs, cartsz);
  cs_cart_coords(lb->cs, coords);
  cs_cart_comm(lb->cs, &comm);
  MPI_Comm_rank(comm, &rank);
  if (rank != (coords[X]*cartsz[Y] + coords[Y])*cartsz[Z] + coords[Z]) return 0;
  if (rank != cs_cart_rank(lb->cs)) return 0;
  for (int d = 0; d < 3; d++) {
    int c[3] = {coords[X], coords[Y], coords[Z]};
    c[d] = (coords[d] + 1) % cartsz[d];
    if (cs_cart_neighb(lb->cs, CS_FORW, d) != (c[X]*cartsz[Y] + c[Y])*cartsz[Z] + c[Z]) return 0;
    c[d] = (coords[d] + cartsz[d] - 1) % cartsz[d];
    if (cs_cart_neighb(lb->cs, CS_BACK, d) != (c[X]*cartsz[Y] + c[Y])*cartsz[Z] + c[Z]) return 0;
  }
  return 1;
}

static int shim_supported(lb_t * lb) {
  int dim = X;
  if (DATA_MODEL != DATA_MODEL_SOA) return 0;
  if (lb->ndist != 1 && lb->ndist != 2) return 0;
  if (lb->model.nvel != 19 && lb->model.nvel != 27) return 0;
  if (lb->haloscheme != LB_HALO_TARGET) return 0;
  if (!shim_slab_dim(lb, &dim)) {
    static int told = 0;
    static int ok = -1;
    int cartsz[3];
    if (ok < 0) ok = shim_cart_general_ok(lb);
    if (ok) return 1;
    cs_cartsz(lb->cs, cartsz);
    if (!told) {
      pe_info(lb->pe, "liblbmi: decomposition %d_%d_%d with ranks numbered "
	      "otherwise than MPI_Cart_create numbers them without reordering: "
	      "this run uses the reference's own lattice Boltzmann kernels\n",
	      cartsz[X], cartsz[Y], cartsz[Z]);
    }
    told = 1;
    return 0;
  }
  return 1;
}

/* Device array pointers are members of the DEVICE copy of each struct:
 * fetch them the way the reference does (model.c:578, propagation.c:240) */

static void shim_device_f(lb_t * lb, double ** f, double ** fprime) {
  tdpAssert(tdpMemcpy(f, &lb->target->f, sizeof(double *),
		      tdpMemcpyDeviceToHost));
  tdpAssert(tdpMemcpy(fprime, &lb->target->fprime, sizeof(double *),
		      tdpMemcpyDeviceToHost));
}

/* Device data pointers are fixed once an object exists, and every fetch is a
 * blocking copy that drains the stream: remember them ... */

#define SHIM_NCACHE 32
static struct { const void * obj; void * data; } shim_cache_[SHIM_NCACHE];
static int shim_ncache_ = 0;

static void * shim_cached(const void * obj, const void * device_member,
			  size_t sz) {
  void * data = NULL;
  for (int n = 0; n < shim_ncache_; n++) {
    if (shim_cache_[n].obj == obj) return shim_cache_[n].data;
  }
  tdpAssert(tdpMemcpy(&data, device_member, sz, tdpMemcpyDeviceToHost));
  if (shim_ncache_ < SHIM_NCACHE) {
    shim_cache_[shim_ncache_].obj = obj;
    shim_cache_[shim_ncache_].data = data;
    shim_ncache_ += 1;
  }
  return data;
}

/* ... and forgotten when the object goes (lb_free, field_free, map_free
 * below): a later object at the same address has other device arrays */

static void shim_forget(const void * obj) {
  int n = 0;
  while (n < shim_ncache_) {
    if (shim_cache_[n].obj == obj) {
      shim_cache_[n] = shim_cache_[shim_ncache_ - 1];
      shim_ncache_ -= 1;
    }
    else {
      n += 1;
    }
  }
}

/* the gradient arrays of a field_grad_t: keyed by the address of the host
 * member, forgotten with the field they belong to */

static void shim_grad_arrays(field_grad_t * fg, double ** grad, double ** delsq) {
  *grad = (double *) shim_cached(&fg->grad, &fg->target->grad, sizeof(double *));
  *delsq = (double *) shim_cached(&fg->delsq, &fg->target->delsq, sizeof(double *));
}

static double * shim_field_data(field_t * field) {
  return (double *) shim_cached(field, &field->target->data, sizeof(double *));
}

/* After any call that swaps f and fprime, make lb->target->f/fprime point
 * at the current arrays, so that foreign kernels (wall.c:930-950,
 * bbl.c:294-360, stats_distribution.c:322) keep working. */

static double * last_f = NULL;         /* what lb->target holds now */
static double * last_fprime = NULL;

static int ptr_owed_ = 0;              /* ... and that is not the current pair */

static void shim_sync_pointers(lb_t * lb, lbmi_t * h) {
  double * f = NULL;
  double * fprime = NULL;
  SHIM_CHECK(lb, lbmi_lb_pointers(h, &f, &fprime));
  if (f == last_f && fprime == last_fprime) {          /* nothing swapped */
    ptr_owed_ = 0;
    return;
  }
  if (shim_.mode == LBMI_MODE_FUSED && lb == shim_.lb && h == shim_.h) {
    /* In `fused` nothing of the reference works on f between lb_collide and
     * lb_propagation, and every reader after that comes through this file:
     * the five launches that take f from lb->target sit behind symbols bound
     * here, and reach their originals through shim_note(., 0), which stores
     * the pair first. A step then costs no 4.5 us store kernel (a fifth of
     * the step of a 64^3 lattice). */
    ptr_owed_ = 1;
    return;
  }
  /* (lb_model_swapf does this with two blocking copies, propagation.c:240-248;
   * here a one-thread kernel on the stream of the step writes the two
   * members of the device struct: foreign kernels, launched on the same
   * default stream afterwards, see the new pair) */
  SHIM_CHECK(lb, lbmi_lb_pointers_store(h, &lb->target->f, &lb->target->fprime));
  last_f = f;
  last_fprime = fprime;
  ptr_owed_ = 0;
}

static void shim_pointers_settle(void) {
  double * f = NULL;
  double * fprime = NULL;
  if (!ptr_owed_ || shim_.h == NULL || shim_.lb == NULL) return;
  ptr_owed_ = 0;
  SHIM_CHECK(shim_.lb, lbmi_lb_pointers(shim_.h, &f, &fprime));
  if (f == last_f && fprime == last_fprime) return;
  SHIM_CHECK(shim_.lb, lbmi_lb_pointers_store(shim_.h, &shim_.lb->target->f,
					      &shim_.lb->target->fprime));
  last_f = f;
  last_fprime = fprime;
}

static lbmi_t * shim_handle(lb_t * lb) {

  if (shim_.h != NULL && shim_.lb == lb) return shim_.h;
  if (shim_.h != NULL) lbmi_free(shim_.h);

  {
    lbmi_options_t opts;
    int cartsz[3], coords[3];
    int slabdim = X;
    int general = 0;
    double * f = NULL;
    double * fprime = NULL;
    /* LBMI_MODE unset (the default): the run starts in LBMI_MODE_FUSED -- halo
     * swap and propagation both deferred into the next collision, one pass
     * over f per step -- and stays there as long as nothing works on f between
     * lb_collide and lb_propagation. Everything in the reference that does
     * comes through this file and moves the handle to `halo`
     * (LBMI_MODE_FUSED_HALO: lb_collide and lb_halo leave exactly the
     * reference's state, only the propagation is deferred) when it has work
     * to do: wall_set_wall_distributions / wall_bbl with links,
     * bounce_back_on_links with colloids, open boundaries (their impose step,
     * ludwig.c:823-832), Lees-Edwards planes (lb_le_apply_boundary_conditions
     * goes through lb_memcpy both ways every step, model_le.c:72-83). Every
     * reader of f AFTER lb_propagation that the reference has -- lb_memcpy,
     * the statistics, lb_io_write, phi_lb_to_field -- is bound and flushes
     * first. LBMI_MODE = halo | eager | fused overrides: halo and eager start
     * (and stay) there; fused is the default with its name spelt out. */
    const char * mode = getenv("LBMI_MODE");

    lbmi_options_default(&opts);
    opts.nvel = lb->model.nvel;
    opts.ndist = lb->ndist;
    cs_nlocal(lb->cs, opts.nlocal);
    cs_nhalo(lb->cs, &opts.nhalo);
    cs_cartsz(lb->cs, cartsz);
    cs_cart_coords(lb->cs, coords);
    if (shim_slab_dim(lb, &slabdim)) {
      /* slabs (or one rank) */
      opts.cartdim = slabdim;
      opts.cartsz = cartsz[slabdim];
      opts.cartrank = coords[slabdim];
    }
    else {
      /* a grid of ranks (shim_supported has checked their numbering) */
      general = 1;
      opts.cartdim = LBMI_CART_GENERAL;
      for (int d = 0; d < 3; d++) {
	opts.cartgrid[d] = cartsz[d];
	opts.cartcoords[d] = coords[d];
      }
      opts.cartsz = cartsz[X]*cartsz[Y]*cartsz[Z];
      opts.cartrank = (coords[X]*cartsz[Y] + coords[Y])*cartsz[Z] + coords[Z];
    }
    opts.device = -1;                            /* ludwig.c:467-492 chose it */
    opts.halo_scheme = LBMI_HALO_FULL;           /* halo_swap_packed semantics */
    opts.mode = LBMI_MODE_FUSED;                 /* ndist 1 or 2 */
    if (mode && mode[0] == 'e') opts.mode = LBMI_MODE_EAGER;
    if (mode && mode[0] == 'h') opts.mode = LBMI_MODE_FUSED_HALO;
    if (mode && mode[0] != 'e' && mode[0] != 'f' && mode[0] != 'h') {
      pe_fatal(lb->pe, "liblbmi: LBMI_MODE=%s (halo, eager or fused)\n", mode);
    }

    SHIM_CHECK(lb, lbmi_create(&opts, &shim_.h));
    /* Ludwig launches all its kernels on the default stream
     * (tdpLaunchKernel(..., 0, 0, ...)): run ours there too, so that the
     * kernels on either side of each call are ordered without extra syncs */
    SHIM_CHECK(lb, lbmi_set_stream(shim_.h, NULL));
        shim_.lb = lb;
    shim_.mode = opts.mode;
        shim_.wall = NULL;
    shim_.wall_nlink = 0;
    shim_.param_valid = 0;
    shim_.colloids = 0;
    shim_.ncollide = 0;
    shim_.nlazy = 0;
    shim_.nlazy_rho = 0;
    shim_.automode = (mode == NULL);

    shim_device_f(lb, &f, &fprime);
    SHIM_CHECK(lb, lbmi_lb_bind(shim_.h, f, fprime));
    last_f = f;
    last_fprime = fprime;

    if (cartsz[slabdim] > 1 || general) {
      /* ncclUniqueId from rank 0 of the Cartesian communicator */
      char id[LBMI_UNIQUE_ID_BYTES];
      MPI_Comm comm;
      int rank;
      cs_cart_comm(lb->cs, &comm);
      MPI_Comm_rank(comm, &rank);
      if (rank == 0) SHIM_CHECK(lb, lbmi_comm_unique_id(id));
      MPI_Bcast(id, LBMI_UNIQUE_ID_BYTES, MPI_BYTE, 0, comm);
      SHIM_CHECK(lb, lbmi_comm_init(shim_.h, id));
      if ((slabdim == Z || general) && shim_.mode == LBMI_MODE_FUSED) {
	/* slabs along Z, a grid of ranks: the library runs fused as halo
	 * (lbmi_create) */
	shim_.mode = LBMI_MODE_FUSED_HALO;
      }
    }
  }

  return shim_.h;
}

/* lb_collide_param_commit (model.c:342-349) is the reference's only upload of
 * lb->param to the device copy that lb->target->param points at, and the
 * original lb_collide is its only caller (collision.c:157). Foreign kernels
 * read it: wall_setu_kernel takes nvel, wv, cv, rho0 from it every step
 * (wall.c:943-945), and with a zero nvel its lb_f_set lands up to
 * 18 nsite doubles BEFORE f. So the bound lb_collide commits as the original
 * does -- when the host struct has changed, not every step: the copy to the
 * symbol is a blocking one. */

static void shim_param_commit(lb_t * lb) {
  if (shim_.param_valid &&
      memcmp(&shim_.param_committed, lb->param, sizeof(lb_collide_param_t)) == 0) {
    return;
  }
  lb_collide_param_commit(lb);
  memcpy(&shim_.param_committed, lb->param, sizeof(lb_collide_param_t));
  if (!shim_.param_valid) {
    /* once: what the device sees through lb->target->param is the model */
    lb_collide_param_t * pdev = NULL;
    int nvel_dev = -1;
    tdpAssert(tdpMemcpy(&pdev, &lb->target->param, sizeof(lb_collide_param_t *),
			tdpMemcpyDeviceToHost));
    tdpAssert(tdpMemcpy(&nvel_dev, &pdev->nvel, sizeof(int),
			tdpMemcpyDeviceToHost));
    if (nvel_dev != lb->model.nvel) {
      pe_fatal(lb->pe, "liblbmi: lb->target->param->nvel = %d on the device, "
	       "model nvel = %d\n", nvel_dev, lb->model.nvel);
    }
  }
  shim_.param_valid = 1;
}

/* The hydro arrays have been, or are about to be, written by code outside
 * the library (the reference's own kernels): drop what the handle remembers
 * about their contents ("known to hold zeros", include/lbmi.h). */

static void shim_hydro_foreign(hydro_t * hydro) {
  if (shim_.h == NULL || hydro == NULL) return;
  if (hydro->nsite != shim_.lb->nsite) return;   /* never seen by the library */
  SHIM_CHECK(shim_.lb, lbmi_hydro_field_dirty(shim_.h, shim_field_data(hydro->force)));
  SHIM_CHECK(shim_.lb, lbmi_hydro_field_dirty(shim_.h, shim_field_data(hydro->u)));
  SHIM_CHECK(shim_.lb, lbmi_hydro_field_dirty(shim_.h, shim_field_data(hydro->rho)));
}

static void shim_needs_canonical_f(lb_t * lb, const char * who);
static int shim_fuse_collide_ok(lb_t * lb, hydro_t * hydro, noise_t * noise,
				fe_t * fe, visc_t * visc);

/*****************************************************************************
 *
 *  lb_collide
 *
 *****************************************************************************/

int lb_collide(lb_t * lb, hydro_t * hydro, map_t * map, noise_t * noise,
	       fe_t * fe, visc_t * visc) {

  if (hydro == NULL) return 0;                   /* collision.c:149 */

  assert(lb);
  assert(map);

  /* the free-energy sector noted on the way here: taken along below if this
   * collision qualifies, run now if not */
  if (fuse_.stage != FUSE_NONE &&
      !(fuse_.stage == FUSE_UZERO &&
	shim_fuse_collide_ok(lb, hydro, noise, fe, visc))) {
    shim_fuse_flush();
  }

  /* Not covered by liblbmi: fluctuations, stress relaxation with a free
   * energy other than the symmetric one; two
   * distributions only with the symmetric free energy (as the reference,
   * collision.c:160) */
  /* (isothermal fluctuations: the D3Q19 collisions have them, lbmi_noise_set;
   * with one distribution and the stress relaxed, or -- where the reference
   * itself cannot, noise.h:18 -- on D3Q27: the original) */
  if (!shim_supported(lb) ||
      (noise->on[NOISE_RHO] && (lb->model.nvel != 19 ||
				(lb->ndist == 1 && fe && fe->use_stress_relaxation) ||
				noise->nsites != lb->nsite)) ||
      (visc != NULL && lb->ndist != 1) ||
      (fe && fe->use_stress_relaxation && fe->id != FE_SYMMETRIC) ||
      (lb->ndist == 2 && (fe == NULL || fe->id != FE_SYMMETRIC))) {
    /* (with Lees-Edwards planes the hydro arrays and every field carry
     * buffer planes, lees_edw_nsites: their components are hydro->nsite
     * apart, the distributions' lb->nsite: lbmi_hydro_t::nsite and
     * lbmi_fe_symm_t::nsite below say so) */
    if (shim_.h && shim_.lb == lb) {
      SHIM_CHECK(lb, lbmi_lb_flush(shim_.h));
      shim_sync_pointers(lb, shim_.h);
      /* the original reads hydro->force as somebody outside the library
       * left it and writes hydro->rho, u itself: nothing the library
       * remembers about their contents holds after this call */
      shim_hydro_foreign(hydro);
    }
    fuse_.armed = 0;
    fuse_.cand = 0;
    return lb_collide_ref(lb, hydro, map, noise, fe, visc);
  }

  {
    lbmi_t * h = shim_handle(lb);
    lbmi_hydro_t hy;
    physics_t * phys = NULL;
    double rho0, eta, zeta, fbody[3];
    int scheme = LBMI_RELAXATION_M10;
    char * status = NULL;

    /* lb_collision_relaxation_times_set (collision.c:1181-1264) and the
     * constant part of lb_collision_parameters_commit (:1928-1980) */
    physics_ref(&phys);
    physics_rho0(phys, &rho0);
    physics_eta_shear(phys, &eta);
    physics_eta_bulk(phys, &zeta);
    physics_fbody(phys, fbody);
    {
      /* pulsatile part of the body force, collision.c:1954-1965 */
      PI_DOUBLE(pi);
      double amp[3], freq;
      double t = physics_control_timestep(phys);
      physics_fpulse(phys, amp);
      physics_fpulse_frequency(phys, &freq);
      for (int ia = 0; ia < 3; ia++) fbody[ia] += amp[ia]*sin(2.0*pi*freq*t);
    }
    if (lb->nrelax == LB_RELAXATION_BGK) scheme = LBMI_RELAXATION_BGK;
    if (lb->nrelax == LB_RELAXATION_TRT) scheme = LBMI_RELAXATION_TRT;
        lb_collision_relaxation_times_set(lb);       /* lb->param, host (collision.c:155) */
    shim_param_commit(lb);                       /* ... and device (collision.c:157) */
    SHIM_CHECK(lb, lbmi_set_relaxation(h, scheme, rho0, eta, zeta));
    SHIM_CHECK(lb, lbmi_set_body_force(h, fbody));

    if (noise->on[NOISE_RHO]) {
      /* collision.c:476-518: the generator states live in noise->target->state;
       * the temperature is the global one (collision.c:1951) */
      double kt = 0.0;
      unsigned int * state = (unsigned int *)
	shim_cached(noise, &noise->target->state, sizeof(unsigned int *));
      physics_kt(phys, &kt);
      SHIM_CHECK(lb, lbmi_noise_set(h, state, noise->nsites, kt,
				    lb->param->isghost == LB_GHOST_ON));
    }
    else {
      SHIM_CHECK(lb, lbmi_noise_set(h, NULL, 0, 0.0, 0));
    }

    /* work on f between lb_collide and lb_propagation that is known here
     * (the rest announces itself: shim_needs_canonical_f) */
    if (shim_.mode == LBMI_MODE_FUSED && shim_.automode) {
      if (hydro->le && lees_edw_nplane_total(hydro->le) > 0) {
	shim_needs_canonical_f(lb, "lb_le_apply_boundary_conditions");
      }
      if (shim_openbc_ > 0) shim_needs_canonical_f(lb, "an open boundary");
    }

    status = (char *) shim_cached(map, &map->target->status, sizeof(char *));
    hy.force  = shim_field_data(hydro->force);
    hy.status = status;
    hy.rho    = shim_field_data(hydro->rho);
    hy.u      = shim_field_data(hydro->u);
    /* a viscosity model has left the local viscosity in hydro->eta
     * (collision.c:386-404, 1947) */
    hy.eta    = visc ? shim_field_data(hydro->eta) : NULL;
    hy.nsite  = hydro->nsite;                    /* component stride of force, u */

    /* hydro_f_zero below tells the library that hydro->force holds zeros,
     * and a force field of zeros is not read. That holds until somebody
     * writes to it, and everybody who does inside the reference --
     * phi_force_calculation, fe_lc_droplet_bodyforce, psi_force_*,
     * nernst_planck_driver (all need a free energy, ludwig.c:643-738) and
     * subgrid_force_from_particles (colloids, ludwig.c:2071, 2149) -- is
     * outside the library: with a free energy or colloids the force is taken
     * to have been written every step. */
    if (fuse_.stage == FUSE_UZERO) {
      /* (folded step: nobody has written to the force since hydro_f_zero) */
    }
    else if (fe != NULL || shim_.colloids || shim_.ncollide == 0) {
      /* (the first collision: colloids show at the first
       * bounce_back_on_links, after it) */
      SHIM_CHECK(lb, lbmi_hydro_field_dirty(h, hy.force));
    }
    shim_.ncollide += 1;

    /* hydro->rho, u on demand. The reference's collision stores them at every
     * site every step (collision.c:563-596); who reads them on the device
     * before the next collision overwrites them? In ludwig.c's loop:
     *   phi_cahn_hilliard / ch_solver -> advection (advection.c:517-951),
     *   phi_force_colloid (:158-302), phi_grad_mu (:76-270), lc_droplet
     *   (:787), blue_phase_beris_edwards (:515, 1043), leslie_ericksen
     *   (:133, 303), nernst_planck, psi_force_*        all need a free energy;
     *   visc_arrhenius (:169) and the other models     a visc_t;
     *   subgrid, stats_calibration, build              colloids (and go through
     *                                                  hydro_memcpy, bound);
     *   lb_bc_inflow/outflow_rhou update + impose      open boundaries;
     *   Lees-Edwards buffer planes of u                le planes.
     * Without any of them -- each is known here: fe, visc, what
     * bounce_back_on_links and the open-boundary constructors have seen,
     * hydro->le -- the only readers left are hydro_memcpy (statistics, output)
     * and hydro_u_halo, both bound: they get rho, u formed first
     * (lbmi_lb_hydro_sync), equal to rounding. So that is the default;
     * LBMI_HYDRO=store keeps the reference's stores in every run,
     * LBMI_HYDRO=lazy is the default spelt out. 32 B/site per step. */
    {
      static int wanted = -1;
      int lazy;
      if (wanted < 0) {
	const char * e = getenv("LBMI_HYDRO");
	wanted = (e == NULL || e[0] == 'l');
	if (e != NULL && e[0] != 'l' && e[0] != 's') {
	  pe_fatal(lb->pe, "liblbmi: LBMI_HYDRO=%s (lazy or store)\n", e);
	}
      }
      lazy = (wanted && fe == NULL && visc == NULL && !shim_.colloids &&
	      lb->ndist == 1 && shim_openbc_ == 0 &&
	      !(hydro->le && lees_edw_nplane_total(hydro->le) > 0));
      shim_.nlazy += lazy;
      /* hydro->rho alone has no reader on the device anywhere in the
       * reference but the open boundaries (lb_bc_inflow_rhou.c:271-502,
       * lb_bc_outflow_rhou.c:392-504; everything else takes it from the host
       * copy, through hydro_memcpy): where u must be stored -- a free energy
       * advects with it, a viscosity model reads it -- the density still need
       * not be, 8 B/site per step */
      if (!lazy && wanted && lb->ndist == 1 && shim_openbc_ == 0 &&
	  !(hydro->le && lees_edw_nplane_total(hydro->le) > 0)) {
	lazy = 2;
	shim_.nlazy_rho += 1;
      }
      SHIM_CHECK(lb, lbmi_tune(h, "hydro_lazy", lazy));
    }

    if (fuse_.stage == FUSE_UZERO) {
      /* hydro_f_zero .. lb_collide of this step in one call */
      fe_symm_t * fs = (fe_symm_t *) fe;
      fe_symm_param_t param;
      double mobility = 0.0;
      int order = 0;
      size_t nsites = (size_t) fuse_.phi->nsites;
      double * phid = shim_field_data(fuse_.phi);
      fe_symm_param(fs, &param);
      physics_mobility(phys, &mobility);
      advection_order(&order);
      if (fuse_.sites < nsites) {
	if (fuse_.phinew) tdpAssert(tdpFree(fuse_.phinew));
	if (fuse_.uprev) tdpAssert(tdpFree(fuse_.uprev));
	tdpAssert(tdpMalloc((void **) &fuse_.phinew, sizeof(double)*nsites));
	tdpAssert(tdpMalloc((void **) &fuse_.uprev, 3*sizeof(double)*nsites));
	fuse_.sites = nsites;
      }
      {
	static int told = 0;
	if (!told) pe_info(lb->pe, "liblbmi: free-energy sector folded into lb_collide "
			   "(LBMI_FE=1: call by call)\n");
	told = 1;
      }
      SHIM_CHECK(lb, lbmi_fe_scheme_set(h, fuse_.fgrad->d2 == grad_3d_7pt_fluid_d2 ? 7 : 27,
					order));
      {
	/* the velocities of the previous collision are read at the neighbours
	 * while the new ones are written, and phi likewise: each from the
	 * array that holds the latest values into the other one */
	double * ua = hy.u;
	const double * usrc = fuse_.u_in_b ? fuse_.uprev : ua;
	const double * psrc = fuse_.phi_in_q ? fuse_.phinew : phid;
	double * pdst = fuse_.phi_in_q ? phid : fuse_.phinew;
	hy.u = fuse_.u_in_b ? ua : fuse_.uprev;
	SHIM_CHECK(lb, lbmi_symmetric_lb_collide(h, &hy, usrc, param.a, param.b,
						 param.kappa, mobility, psrc, pdst));
	fuse_.u_in_b = !fuse_.u_in_b;
	fuse_.phi_in_q = !fuse_.phi_in_q;
      }
      shim_note(S_FIELD_HALO, 1);
      shim_note(S_FIELD_GRAD_COMPUTE, 1);
      shim_note(S_PHI_FORCE_CALCULATION, 1);
      shim_note(S_PHI_CAHN_HILLIARD, 1);
      shim_note(S_HYDRO_U_ZERO, 1);
      fuse_.stage = FUSE_NONE;
      fuse_.nfused += 1;
    }
    else if (lb->ndist == 2 || (fe && fe->use_stress_relaxation)) {
      /* lb_collision_binary (collision.c:610-1027), or the single-fluid
       * collision with the symmetric stress relaxed (:413-429) */
      fe_symm_t * fs = (fe_symm_t *) fe;
      fe_symm_param_t param;
      lbmi_fe_symm_t bin;
      fe_symm_param(fs, &param);
      bin.a = param.a;
      bin.b = param.b;
      bin.kappa = param.kappa;
      physics_mobility(phys, &bin.mobility);
      bin.nsite = fs->phi->nsites;                /* component stride of grad phi */
      bin.phi = shim_field_data(fs->phi);
      shim_grad_arrays(fs->dphi, (double **) &bin.grad, (double **) &bin.delsq);
      if (lb->ndist == 2) {
	SHIM_CHECK(lb, lbmi_lb_collide_binary(h, &hy, &bin));
      }
      else {
	SHIM_CHECK(lb, lbmi_lb_collide_fe(h, &hy, &bin));
      }
    }
    else {
      SHIM_CHECK(lb, lbmi_lb_collide(h, &hy));
    }
    shim_note(S_LB_COLLIDE, 1);
    shim_sync_pointers(lb, h);                   /* FUSED swaps here */

    /* did this step go through the whole sequence in the case covered? Then
     * the next one may be folded */
    if (!fuse_.armed && fuse_.cand == 3 && shim_fuse_wanted() &&
	fuse_.seen_fgrad != NULL && fuse_.seen_fgrad->field == fuse_.phi) {
      fuse_.fgrad = fuse_.seen_fgrad;
      fuse_.armed = shim_fuse_collide_ok(lb, hydro, noise, fe, visc);
    }
    fuse_.cand = 0;
  }

  return 0;
}

/*****************************************************************************
 *
 *  phi_lb_to_field  (phi_lb_coupler.c:39-64)
 *
 *****************************************************************************/

int phi_lb_to_field(field_t * phi, lb_t * lb) {
  shim_fuse_flush();

  assert(phi);
  assert(lb);

  if (!shim_supported(lb) || lb->ndist != 2) return phi_lb_to_field_ref(phi, lb);

  shim_note(S_PHI_LB_TO_FIELD, 1);
  SHIM_CHECK(lb, lbmi_lb_phi_to_field(shim_handle(lb), shim_field_data(phi)));

  return 0;
}

/*****************************************************************************
 *
 *  wall_bbl  (wall.c:960-989): bounce-back on the reference's links
 *
 *  The links are taken from the HOST arrays wall->linki, linkj, linkp, linku
 *  (and linkk, linkq, links with slip) that wall_init_boundaries, wall_init_uw
 *  and wall_init_boundaries_slip fill (wall.c:399-451, 864-890, 489-593):
 *  liblbmi checks every record and keeps its own device copies, once per
 *  wall_t. It does not rely on wall->target->link*, whose device allocation
 *  goes through "int tmp; tdpMalloc((void **) &tmp, ...)" (wall.c:412-426,
 *  516-528). The momentum goes where the reference's kernel puts it,
 *  wall->target->fnet, so wall_momentum() works unchanged.
 *
 *****************************************************************************/

static void shim_wall_links(wall_t * wall, lbmi_t * h) {

  if (shim_.wall == wall && shim_.wall_nlink == wall->nlink) return;

  SHIM_CHECK(wall->lb, lbmi_wall_links_set(h, wall->nlink, wall->linki,
					   wall->linkj, wall->linkp,
					   wall->linku));
  if (wall->param->slip.active) {                         /* wall.c:971 */
    SHIM_CHECK(wall->lb,
	       lbmi_wall_slip_links_set(h, wall->linkk,
					(const signed char *) wall->linkq,
					(const signed char *) wall->links,
					wall->param->slip.s));
  }
  SHIM_CHECK(wall->lb,
	     lbmi_wall_fnet_bind(h, (double *) ((char *) wall->target
						+ offsetof(wall_t, fnet))));
  shim_.wall = wall;
  shim_.wall_nlink = wall->nlink;
}

/* Kernels of the reference that read or write lb->target->f between
 * lb_collide and lb_propagation (wall_setu_kernel wall.c:930-950, the
 * colloid bounce-back bbl.c:272-360) need the post-collision state with its
 * halo in the reference's order: EAGER and FUSED_HALO have exactly that,
 * FUSED has not (halo swap deferred, blocked order). A run in LBMI_MODE=fused
 * that gets here with work to do continues in FUSED_HALO. */

static void shim_needs_canonical_f(lb_t * lb, const char * who) {
  if (shim_.h == NULL || shim_.lb != lb) return;
  if (shim_.mode != LBMI_MODE_FUSED) return;
  pe_info(lb->pe, "liblbmi: %s acts on the distributions between lb_collide "
	  "and lb_propagation: execution mode fused -> halo\n", who);
  SHIM_CHECK(lb, lbmi_lb_mode_set(shim_.h, LBMI_MODE_FUSED_HALO));
  shim_.mode = LBMI_MODE_FUSED_HALO;
  ptr_owed_ = 1;                                 /* (whatever was left owing) */
  shim_pointers_settle();
  shim_sync_pointers(lb, shim_.h);
}

int wall_bbl(wall_t * wall) {
  shim_fuse_flush_lb();

  assert(wall);
  assert(wall->target);

  if (wall->nlink == 0) return 0;                /* wall.c:967 */

  if (!shim_supported(wall->lb)) {
    return wall_bbl_ref(wall);
  }

  {
    lbmi_t * h = shim_handle(wall->lb);
    char * status = NULL;

    shim_needs_canonical_f(wall->lb, "wall_bbl");
    shim_wall_links(wall, h);
    /* the kernels test map->status[i] for MAP_COLLOID (wall.c:1046, 1146) */
    status = (char *) shim_cached(wall->map, &wall->map->target->status,
				  sizeof(char *));
    SHIM_CHECK(wall->lb, lbmi_wall_status_set(h, status));
    SHIM_CHECK(wall->lb, lbmi_wall_velocity_set(h, wall->param->ubot,
						wall->param->utop));
    shim_note(S_WALL_BBL, 1);
    SHIM_CHECK(wall->lb, lbmi_wall_bbl(h));
  }

  return 0;
}

/*****************************************************************************
 *
 *  wall_set_wall_distributions (wall.c:900-950), bounce_back_on_links
 *  (bbl.c:147-200): the originals, on distributions that are what they expect
 *
 *****************************************************************************/

int wall_set_wall_distributions(wall_t * wall) {
  shim_fuse_flush_lb();

  assert(wall);

  if (wall->nlink == 0) return 0;                /* wall.c:907 */
  shim_needs_canonical_f(wall->lb, "wall_set_wall_distributions");

  return wall_set_wall_distributions_ref(wall);
}

int bounce_back_on_links(bbl_t * bbl, lb_t * lb, wall_t * wall,
			 colloids_info_t * cinfo) {
  shim_fuse_flush_lb();
  int ntotal = 0;

  assert(lb);
  assert(cinfo);

  colloids_info_ntotal(cinfo, &ntotal);
  if (ntotal == 0) return 0;                     /* bbl.c:160 */
  if (ntotal > 0) {
    if (!shim_.colloids && shim_.h != NULL && shim_.lb == lb) {
      /* the collision of this step may have left rho, u on demand: from now
       * on they are stored (lb_collide), and this once they are formed */
      SHIM_CHECK(lb, lbmi_lb_hydro_sync(shim_.h));
    }
    shim_.colloids = 1;
    shim_needs_canonical_f(lb, "bounce_back_on_links");
  }

  return bounce_back_on_links_ref(bbl, lb, wall, cinfo);
}

/*****************************************************************************
 *
 *  lb_bc_inflow_rhou_create, lb_bc_outflow_rhou_create (lb_bc_open_rt.c:119,
 *  160): the originals. Their update / impose steps read hydro->rho, u and
 *  set distributions between lb_halo and lb_propagation (ludwig.c:600-605,
 *  823-832) through function tables this file cannot bind: a run that has
 *  created one keeps the reference's state at that point (halo) and stores
 *  rho, u in every collision.
 *
 *****************************************************************************/

int lb_bc_inflow_rhou_create(pe_t * pe, cs_t * cs,
			     const lb_bc_inflow_opts_t * options,
			     lb_bc_inflow_rhou_t ** inflow) {
  shim_openbc_ += 1;
  return lb_bc_inflow_rhou_create_ref(pe, cs, options, inflow);
}

int lb_bc_outflow_rhou_create(pe_t * pe, cs_t * cs,
			      const lb_bc_outflow_opts_t * options,
			      lb_bc_outflow_rhou_t ** outflow) {
  shim_openbc_ += 1;
  return lb_bc_outflow_rhou_create_ref(pe, cs, options, outflow);
}

/*****************************************************************************
 *
 *  lb_free, field_free, map_free, wall_free: the originals, after dropping
 *  what this file remembers under the object's address
 *
 *****************************************************************************/

int lb_free(lb_t * lb) {
  shim_fuse_flush();
  fuse_.armed = 0;
  fuse_.cand = 0;
  assert(lb);
  if (shim_.lb == lb) {
    /* the handle borrows lb->target->f / fprime: it goes first */
    if (shim_.h) lbmi_free(shim_.h);
    shim_ended_ = shim_;                         /* for the report at exit */
    memset(&shim_, 0, sizeof(shim_));
    last_f = NULL;
    last_fprime = NULL;
    ptr_owed_ = 0;
  }
  return lb_free_ref(lb);
}

int field_free(field_t * obj) {
  shim_fuse_flush();
  if (obj == fuse_.phi) {
    fuse_.armed = 0;
    fuse_.cand = 0;
    fuse_.phi = NULL;
  }
  if (shim_.h != NULL && obj != NULL) {
    /* "known to hold zeros" is kept by device address: the next allocation
     * may get this one */
    for (int n = 0; n < shim_ncache_; n++) {
      if (shim_cache_[n].obj == obj) {
	lbmi_hydro_field_dirty(shim_.h, (const double *) shim_cache_[n].data);
      }
    }
  }
  shim_forget(obj);
  return field_free_ref(obj);
}

void field_grad_free(field_grad_t * obj) {
  shim_fuse_flush();
  if (obj == fuse_.fgrad || obj == fuse_.seen_fgrad) {
    fuse_.armed = 0;
    fuse_.fgrad = NULL;
    fuse_.seen_fgrad = NULL;
  }
  if (obj) {
    shim_forget(&obj->grad);
    shim_forget(&obj->delsq);
  }
  field_grad_free_ref(obj);
}

int map_free(map_t * obj) {
  shim_forget(obj);
  return map_free_ref(obj);
}

int noise_free(noise_t * obj) {
  shim_forget(obj);
  if (shim_.h) lbmi_noise_set(shim_.h, NULL, 0, 0.0, 0);
  return noise_free_ref(obj);
}

int wall_free(wall_t * wall) {
  if (shim_.wall == wall) {
    if (shim_.h) lbmi_wall_links_set(shim_.h, 0, NULL, NULL, NULL, NULL);
    shim_.wall = NULL;
    shim_.wall_nlink = 0;
  }
  return wall_free_ref(wall);
}

/*****************************************************************************
 *
 *  lb_halo, lb_halo_swap
 *
 *****************************************************************************/

int lb_halo_swap(lb_t * lb, lb_halo_enum_t flag) {
  shim_fuse_flush_lb();

  assert(lb);

  if (!shim_supported(lb) || flag != LB_HALO_TARGET) {
    return lb_halo_swap_ref(lb, flag);
  }

  shim_note(S_LB_HALO_SWAP, 1);
  SHIM_CHECK(lb, lbmi_lb_halo(shim_handle(lb)));

  return 0;
}

int lb_halo(lb_t * lb) {

  assert(lb);

  return lb_halo_swap(lb, lb->haloscheme);
}

/*****************************************************************************
 *
 *  lb_propagation
 *
 *****************************************************************************/

int lb_propagation(lb_t * lb) {
  shim_fuse_flush_lb();

  assert(lb);

  if (!shim_supported(lb)) return lb_propagation_ref(lb);

  {
    lbmi_t * h = shim_handle(lb);
    shim_note(S_LB_PROPAGATION, 1);
    SHIM_CHECK(lb, lbmi_lb_propagation(h));
    shim_sync_pointers(lb, h);                   /* lb_model_swapf */
  }

  return 0;
}

/*****************************************************************************
 *
 *  lb_memcpy
 *
 *  Device -> host must see the canonical state: flush a deferred halo swap
 *  and propagation first (no-op in EAGER mode).
 *
 *****************************************************************************/

int lb_memcpy(lb_t * lb, tdpMemcpyKind flag) {
  shim_fuse_flush();

  assert(lb);

  /* ludwig.c:507 copies the initial distributions to the device before the
   * first step: the handle comes into being here, so that the free-energy
   * sector of the first step and the report of step 0 are already bound */
  if (flag == tdpMemcpyHostToDevice && shim_supported(lb)) (void) shim_handle(lb);

  if (shim_.h && shim_.lb == lb) {
    SHIM_CHECK(lb, lbmi_lb_flush(shim_.h));
    SHIM_CHECK(lb, lbmi_synchronize(shim_.h));
    shim_sync_pointers(lb, shim_.h);
  }

  {
    int ifail = lb_memcpy_ref(lb, flag);
    if (flag == tdpMemcpyHostToDevice && shim_.h && shim_.lb == lb) {
      /* f rewritten behind the library's back (the first copy of ludwig.c:507,
       * the Lees-Edwards reprojection model_le.c:72-83, a restart): planes a
       * slab has already sent ahead for its next fused step are stale */
      SHIM_CHECK(lb, lbmi_lb_dirty(shim_.h));
    }
    return ifail;
  }
}


/*****************************************************************************
 *
 *  lb_io_write, lb_io_read  (model.c:1568-1649), row f3
 *
 *  The reference copies all of f to the host, packs the records there and
 *  writes through MPI-IO. In its MPI-IO mode with one file (i/o grid 1_1_1)
 *  and binary records, an X slab is one contiguous byte range of that file:
 *  the records are packed on the device and written from there, the files
 *  are the same byte for byte. Text records (distribution_io_format ascii)
 *  are the library's as well: packed on the device, formatted on the host as
 *  lb_write_buf_ascii formats them. So is the old-style i/o of a run whose
 *  input names no i/o mode (io_harness.c, through lb->io_info) while it keeps
 *  one file of binary records: the same record stream under the old name
 *  with the old text metadata beside it. Several files (distribution_io_grid
 *  N_1_1 over X slabs, MPI-IO mode): each group of slabs one file of its
 *  planes. Anything else (an i/o grid across the slabs, several old-style
 *  files, the old text records) goes to the original.
 *
 *****************************************************************************/

/* Returns 0 (the original does it), or 1 + the format for lbmi_io_format_set */

static int shim_io_supported(lb_t * lb, const io_metadata_t * meta, int reading) {
  int dim = X;
  int fmt = 0;
  if (!shim_supported(lb)) return 0;
  /* (a slab along Y or Z is not one byte range of the file) */
  if (!shim_slab_dim(lb, &dim) || dim != X) return 0;
  /* several files: along the slab direction only (a group of slabs is then a
   * block of planes, one byte range per slab of its file) */
  if (meta->options.iogrid[Y] != 1 || meta->options.iogrid[Z] != 1) return 0;
  if (meta->options.iogrid[X] != 1 && meta->options.mode != IO_MODE_MPIIO) return 0;
  if (meta->subfile.nfile != meta->options.iogrid[X]) return 0;
  if (meta->options.mode == IO_MODE_MPIIO) {
    if (meta->options.iorformat == IO_RECORD_ASCII) fmt = LBMI_IO_ASCII;
    else if (meta->options.iorformat != IO_RECORD_BINARY) return 0;
    return 1 + fmt;
  }
  /* Old-style (model.c:1583-1587, 1629-1633): what lb_io_info_set installed
   * decides, not the metadata. One file of binary records, read by global
   * position: the same record stream under the old name. */
  {
    const io_info_t * info = lb->io_info;
    if (info == NULL || info->io_comm == NULL) return 0;
    if (info->io_comm->n_io != 1) return 0;
    if (reading) {
      if (info->read_data == NULL || info->read_data != info->read_binary) return 0;
      if (!info->single_file_read || !info->processor_independent) return 0;
    }
    else {
      if (info->write_data == NULL || info->write_data != info->write_binary) return 0;
      if (info->args.output.mode == IO_MODE_MULTIPLE) return 0;
      if (info->args.output.report) return 0;          /* its own timing line */
      if (info->bytesize != sizeof(double)*(size_t) lb->nvel*lb->ndist) return 0;
      if (strcmp(info->metadata_stub, "dist") != 0) return 0;
    }
  }
  return 1 + LBMI_IO_SINGLE;
}

/* the file of this rank's i/o group (io_subfile_t) and the periodicity the
 * metadata prints */

static void shim_io_file(lb_t * lb, const io_metadata_t * meta) {
  lbmi_io_file_t file = {0};
  file.nfile = meta->subfile.nfile;
  file.index = meta->subfile.index;
  file.file_nx = meta->subfile.sizes[X];
  file.file_x0 = meta->subfile.offset[X];
  cs_periodic(lb->cs, file.periodic);
  SHIM_CHECK(lb, lbmi_io_file_set(shim_handle(lb), &file));
}

int lb_io_write(lb_t * lb, int timestep, io_event_t * event) {
  shim_fuse_flush();

  int fmt = 0;

  assert(lb);
  assert(event);

  fmt = shim_io_supported(lb, &lb->output, 0);
  if (fmt == 0) {
    return lb_io_write_ref(lb, timestep, event);   /* via lb_memcpy: flushes */
  }
  fmt -= 1;

  {
    int ntotal[3], noffset[3];
    cs_ntotal(lb->cs, ntotal);
    cs_nlocal_offset(lb->cs, noffset);
    if (!(fmt & LBMI_IO_SINGLE)) {
      io_event_record(event, IO_EVENT_AGGR);
      io_event_record(event, IO_EVENT_WRITE);
    }
    /* the metadata file(s) (rank at offset 0) and this rank's byte range */
    shim_note(S_LB_IO_WRITE, 1);
    /* binary records, distribution_io_format ascii (model.c:1438-1462), or
     * the old-style files */
    SHIM_CHECK(lb, lbmi_io_format_set(shim_handle(lb), fmt));
    shim_io_file(lb, &lb->output);
    SHIM_CHECK(lb, lbmi_lb_io_write(shim_handle(lb), ".", timestep, ntotal[X],
				    noffset[X]));
    shim_sync_pointers(lb, shim_.h);                /* a flush may have swapped */
    lb->output.iswriten = 1;
    if (fmt & LBMI_IO_SINGLE) {
      lb->io_info->metadata_written = 1;
    }
    else {
      io_event_report(event, &lb->output, "dist");
    }
  }

  return 0;
}

int lb_io_read(lb_t * lb, int timestep, io_event_t * event) {
  shim_fuse_flush();

  int fmt = 0;

  assert(lb);
  assert(event);

  fmt = shim_io_supported(lb, &lb->input, 1);
  if (fmt == 0) {
    if (shim_.h && shim_.lb == lb) {
      /* the original fills the HOST copy; nothing deferred may survive it */
      SHIM_CHECK(lb, lbmi_lb_flush(shim_.h));
      shim_sync_pointers(lb, shim_.h);
    }
    return lb_io_read_ref(lb, timestep, event);
  }
  fmt -= 1;

  {
    int ntotal[3], noffset[3];
    cs_ntotal(lb->cs, ntotal);
    cs_nlocal_offset(lb->cs, noffset);
    shim_note(S_LB_IO_READ, 1);
    SHIM_CHECK(lb, lbmi_io_format_set(shim_handle(lb), fmt));
    shim_io_file(lb, &lb->input);
    SHIM_CHECK(lb, lbmi_lb_io_read(shim_handle(lb), ".", timestep, ntotal[X],
				   noffset[X]));
    shim_sync_pointers(lb, shim_.h);
    /* ludwig.c:333-340 goes on with the host copy (it is copied to the
     * device again there): keep it in step */
    lb_memcpy_ref(lb, tdpMemcpyDeviceToHost);
  }

  return 0;
}

/*****************************************************************************
 *
 *  Rows f1 / f2 of the scope table: what runs around the LB step with the
 *  symmetric free energy. These need the handle of the lb_t, i.e. they take
 *  effect once the first lb_collide / lb_halo has created it; until then
 *  (initialisation) the originals run.
 *
 *  phi_force_calculation and phi_cahn_hilliard are NOT bound here: their
 *  replacements (lbmi_symmetric_force, lbmi_cahn_hilliard, or the single
 *  pass lbmi_symmetric_step[_periodic]) are valid under conditions only the
 *  maintainer can assert -- symmetric free energy, stress-divergence force,
 *  no walls / colloids / Lees-Edwards planes, no order-parameter noise, no
 *  external chemical-potential gradient, pch->info.conserve == 0 -- and
 *  change the ownership of phi (phi -> phi_out). INTEGRATION.md, section 7.
 *
 *****************************************************************************/

static lbmi_t * shim_handle_if_any(cs_t * cs) {
  int nlocal[3], mine[3];
  if (shim_.h == NULL || shim_.lb == NULL) return NULL;
  /* the same coordinate system as the lb_t the handle was made for */
  cs_nlocal(cs, nlocal);
  cs_nlocal(shim_.lb->cs, mine);
  if (nlocal[X] != mine[X] || nlocal[Y] != mine[Y] || nlocal[Z] != mine[Z]) {
    return NULL;
  }
  return shim_.h;
}

/* hydro_free (hydro.c:127): what the binding still keeps in second arrays
 * goes back first, and nothing of this object is remembered */

int hydro_free(hydro_t * hydro) {
  if (hydro != NULL && hydro == fuse_.hydro) {
    shim_fuse_flush();
    fuse_.armed = 0;
    fuse_.cand = 0;
    fuse_.hydro = NULL;
  }
  return hydro_free_ref(hydro);
}

int hydro_memcpy(hydro_t * hydro, tdpMemcpyKind flag) {
  shim_fuse_flush();
  assert(hydro);
  if (shim_.h != NULL) {
    /* device -> host: rho, u still owed by a lazy collision are formed first;
     * host -> device (subgrid_force_from_particles subgrid.c:200, a restart):
     * settled as well, or they would later be formed over what arrives now */
    SHIM_CHECK(shim_.lb, lbmi_lb_hydro_sync(shim_.h));
    SHIM_CHECK(shim_.lb, lbmi_synchronize(shim_.h));
  }
  {
    int ifail = hydro_memcpy_ref(hydro, flag);
    /* ... and the arrays have been written outside the library: nothing it
     * remembers about their contents (a force field of zeros) holds */
    if (flag == tdpMemcpyHostToDevice) shim_hydro_foreign(hydro);
    return ifail;
  }
}

static int shim_hydro_u_zero_now(hydro_t * hydro, const double uzero[3]);

int hydro_u_zero(hydro_t * hydro, const double uzero[3]) {
  assert(hydro);
  if (fuse_.stage == FUSE_CH && hydro == fuse_.hydro &&
      uzero[X] == 0.0 && uzero[Y] == 0.0 && uzero[Z] == 0.0) {
    /* (the collision that follows writes u at every interior site, and the
     * Cahn-Hilliard update noted before it still wants the old values) */
    fuse_.stage = FUSE_UZERO;
    return 0;
  }
  shim_fuse_flush();
  return shim_hydro_u_zero_now(hydro, uzero);
}

static int shim_hydro_u_zero_now(hydro_t * hydro, const double uzero[3]) {
  lbmi_t * h = NULL;
  assert(hydro);
  h = shim_handle_if_any(hydro->cs);
  if (h == NULL || hydro->nsite != shim_.lb->nsite) return hydro_u_zero_ref(hydro, uzero);
  shim_note(S_HYDRO_U_ZERO, 1);
  SHIM_CHECK(shim_.lb, lbmi_hydro_field_set(h, shim_field_data(hydro->u), 3, uzero));
  return 0;
}

int hydro_f_zero(hydro_t * hydro, const double fzero[3]) {
  lbmi_t * h = NULL;
  assert(hydro);
  shim_fuse_flush_lb();                          /* (neither phi nor u) */
  h = shim_handle_if_any(hydro->cs);
  if (h == NULL || hydro->nsite != shim_.lb->nsite) return hydro_f_zero_ref(hydro, fzero);
  shim_note(S_HYDRO_F_ZERO, 1);
  SHIM_CHECK(shim_.lb, lbmi_hydro_field_set(h, shim_field_data(hydro->force), 3, fzero));
  return 0;
}

/* field_halo: the device scheme, no Lees-Edwards planes, halo width within
 * the halo of the lattice (hydro_u_halo comes through here as well,
 * hydro.c:190-197) */

static int shim_field_halo_now(field_t * field);
static int shim_fuse_special_step(void);

int field_halo(field_t * field) {
  assert(field);
  if (fuse_.armed && fuse_.stage == FUSE_NONE && field == fuse_.phi &&
      shim_.h != NULL && shim_.mode == LBMI_MODE_FUSED && !shim_fuse_special_step()) {
    fuse_.stage = FUSE_HALO;
    return 0;
  }
  shim_fuse_flush();
  return shim_field_halo_now(field);
}

static int shim_field_halo_now(field_t * field) {
  lbmi_t * h = NULL;
  int nhalo = 0;
  int nlocal[3];
  assert(field);
  h = shim_handle_if_any(field->cs);
  cs_nhalo(field->cs, &nhalo);
  cs_nlocal(field->cs, nlocal);
  if (h == NULL || field->opts.haloscheme != FIELD_HALO_TARGET ||
      (field->le && lees_edw_nplane_total(field->le) > 0) ||
      field->nsites != shim_.lb->nsite ||
      field->nhcomm < 1 || field->nhcomm > nhalo || field->nf > 27 ||
      /* (a quasi-two-dimensional system: the swap is wider than the box) */
      field->nhcomm > nlocal[X] || field->nhcomm > nlocal[Y] ||
      field->nhcomm > nlocal[Z]) {
    return field_halo_ref(field);
  }
  shim_note(S_FIELD_HALO, 1);
  SHIM_CHECK(shim_.lb, lbmi_field_halo_n(h, field->nf, field->nhcomm,
					 shim_field_data(field)));
  return 0;
}

/* field_grad_compute for a scalar with the fluid-only 7- or 27-point
 * stencils at level 2 (grad and delsq); anything else is the original */

static int shim_field_grad_compute_now(field_grad_t * fgrad);

int field_grad_compute(field_grad_t * fgrad) {
  assert(fgrad);
  if (fuse_.stage == FUSE_HALO && fgrad == fuse_.fgrad) {
    fuse_.stage = FUSE_GRAD;
    return 0;
  }
  shim_fuse_flush();
  return shim_field_grad_compute_now(fgrad);
}

static int shim_field_grad_compute_now(field_grad_t * fgrad) {
  lbmi_t * h = NULL;
  double * grad = NULL;
  double * delsq = NULL;
  int npt = 0;
  assert(fgrad);
  assert(fgrad->d2);
  h = shim_handle_if_any(fgrad->field->cs);
  if (fgrad->d2 == grad_3d_7pt_fluid_d2) npt = 7;
  if (fgrad->d2 == grad_3d_27pt_fluid_d2) npt = 27;
  if (h == NULL || npt == 0 || fgrad->nf != 1 || fgrad->level != 2 ||
      (fgrad->field->le && lees_edw_nplane_total(fgrad->field->le) > 0)) {
    return field_grad_compute_ref(fgrad);
  }
  shim_note(S_FIELD_GRAD_COMPUTE, 1);
  if (fgrad->nf == 1) fuse_.seen_fgrad = fgrad;   /* (a candidate for the sequence) */
  shim_grad_arrays(fgrad, &grad, &delsq);
  if (npt == 7) {
    SHIM_CHECK(shim_.lb, lbmi_field_grad_7pt(h, shim_field_data(fgrad->field),
					     grad, delsq));
  }
  else {
    SHIM_CHECK(shim_.lb, lbmi_field_grad_27pt(h, shim_field_data(fgrad->field),
					      grad, delsq));
  }
  return 0;
}

/*****************************************************************************
 *
 *  stats_distribution_print, stats_distribution_momentum
 *  (stats_distribution.c:55-139, 201-350): row a17
 *
 *  The reference walks over the HOST copy of f for the density statistics and
 *  runs distribution_gm_kernel -- one compare-and-swap lock per block around a
 *  serial Kahan loop -- for the momentum. Here the density of every interior
 *  site comes from the device (summed in p order, as lb_0th_moment) and is
 *  added up on the host in the reference's site order, so the line printed
 *  is the reference's to the last digit (its variance is a difference of
 *  nearly equal sums: the order of the additions shows in it); the momentum
 *  is the library's shuffle + LDS + two-stage Kahan reduction. Both flush a
 *  deferred propagation first: they do not depend on an lb_memcpy in front.
 *
 *****************************************************************************/

int stats_distribution_print(lb_t * lb, map_t * map) {
  shim_fuse_flush();

  assert(lb);
  assert(map);

  if (!shim_supported(lb) || shim_.h == NULL || shim_.lb != lb) {
    return stats_distribution_print_ref(lb, map);
  }

  {
    int nlocal[3];
    size_t n = 0;
    double stat_local[5] = {0.0, 0.0, 0.0, +DBL_MAX, -DBL_MAX};
    double stat_total[5];
    double rhomean, rhovar;
    double * rho = NULL;
    MPI_Comm comm;

    cs_nlocal(lb->cs, nlocal);
    pe_mpi_comm(lb->pe, &comm);
    rho = (double *) malloc(sizeof(double)*(size_t) nlocal[X]*nlocal[Y]*nlocal[Z]);
    if (rho == NULL) pe_fatal(lb->pe, "liblbmi: malloc(rho) failed\n");
    shim_note(S_STATS_DISTRIBUTION_PRINT, 1);
    SHIM_CHECK(lb, lbmi_lb_density(shim_.h, rho));
    shim_sync_pointers(lb, shim_.h);              /* a flush may have swapped */

    for (int ic = 1; ic <= nlocal[X]; ic++) {     /* stats_distribution.c:73-88 */
      for (int jc = 1; jc <= nlocal[Y]; jc++) {
	for (int kc = 1; kc <= nlocal[Z]; kc++, n++) {
	  int status = MAP_FLUID;
	  map_status(map, cs_index(lb->cs, ic, jc, kc), &status);
	  if (status != MAP_FLUID) continue;
	  stat_local[0] += 1.0;
	  stat_local[1] += rho[n];
	  stat_local[2] += rho[n]*rho[n];
	  stat_local[3] = dmin(rho[n], stat_local[3]);
	  stat_local[4] = dmax(rho[n], stat_local[4]);
	}
      }
    }
    free(rho);

    MPI_Reduce(stat_local, stat_total, 3, MPI_DOUBLE, MPI_SUM, 0, comm);
    MPI_Reduce(stat_local + 3, stat_total + 3, 1, MPI_DOUBLE, MPI_MIN, 0, comm);
    MPI_Reduce(stat_local + 4, stat_total + 4, 1, MPI_DOUBLE, MPI_MAX, 0, comm);

    rhomean = stat_total[1]/stat_total[0];
    rhovar  = (stat_total[2]/stat_total[0]) - rhomean*rhomean;

    pe_info(lb->pe, "\nScalars - total mean variance min max\n");
    pe_info(lb->pe, "[rho] %14.2f %14.11f %14.7e %14.11f %14.11f\n",
	    stat_total[1], rhomean, fabs(rhovar), stat_total[3], stat_total[4]);
  }

  return 0;
}

int stats_distribution_momentum(lb_t * lb, map_t * map, double g[3]) {
  shim_fuse_flush();

  assert(lb);
  assert(map);
  assert(g);

  if (!shim_supported(lb) || shim_.h == NULL || shim_.lb != lb) {
    return stats_distribution_momentum_ref(lb, map, g);
  }

  {
    double out[9];
    double glocal[3];
    char * status = (char *) shim_cached(map, &map->target->status, sizeof(char *));
    MPI_Comm comm;

    pe_mpi_comm(lb->pe, &comm);
    shim_note(S_STATS_DISTRIBUTION_MOMENTUM, 1);
    SHIM_CHECK(lb, lbmi_lb_moments(shim_.h, status, out));
    shim_sync_pointers(lb, shim_.h);
    /* the rank's compensated sums; across ranks a plain sum (the reference
     * merges the compensation terms as well, stats_distribution.c:245-262) */
    glocal[X] = out[5]; glocal[Y] = out[6]; glocal[Z] = out[7];
    g[X] = 0.0; g[Y] = 0.0; g[Z] = 0.0;
    MPI_Reduce(glocal, g, 3, MPI_DOUBLE, MPI_SUM, 0, comm);
  }

  return 0;
}

/*****************************************************************************
 *
 *  phi_force_calculation (phi_force.c:74-136), phi_cahn_hilliard
 *  (phi_cahn_hilliard.c:206-284): row f2 (LBMI_FE=0 leaves them to the original)
 *
 *  One kernel each in place of pth_stress_compute + pth_force_fluid_driver
 *  (a 9-component stress array in between) and of advection_x + the flux
 *  kernels + the update (four flux arrays in between). Only in the case the
 *  replacements were written and tested for -- anything else is the original:
 *  symmetric free energy, force by stress divergence, finite-difference order
 *  parameter (one distribution), no walls, no porous medium, no colloids, no
 *  Lees-Edwards planes, no order-parameter noise, no external chemical-
 *  potential gradient, the plain forward step (conserve == 0), advection
 *  order <= 4, gradients by the 7- or 27-point fluid stencil.
 *
 *****************************************************************************/

static int shim_fe_wanted(void) {
  static int wanted = -1;
  if (wanted < 0) {
    /* on unless LBMI_FE=0: every call checks the one case it was written
     * for and runs the original otherwise. LBMI_FE=1: bound call by call,
     * never folded into the collision. */
    const char * e = getenv("LBMI_FE");
    wanted = !(e != NULL && e[0] == '0');
    if (wanted && e != NULL && e[0] == '1') wanted = 2;
  }
  return wanted;
}

static int shim_fuse_wanted(void) {
  return shim_fe_wanted() == 1;
}

/* a step on which ludwig.c reports or writes something (ludwig.c:868-960) */

static int shim_fuse_special_step(void) {
  return is_statistics_step() || is_measurement_step() || is_config_step() ||
    is_colloid_io_step() || is_phi_output_step() || is_vel_output_step() ||
    is_psi_output_step() || is_fed_output_step() ||
    is_shear_measurement_step() || is_shear_output_step();
}

/* lb_collide with the sector noted in front of it: the plain periodic
 * one-rank two-phase fluid, nothing else at work on f, u or phi */

static int shim_fuse_collide_ok(lb_t * lb, hydro_t * hydro, noise_t * noise,
				fe_t * fe, visc_t * visc) {
  int cartsz[3], periodic[3], nlocal[3];
  int nhalo = 0;
  if (!shim_fuse_wanted() || !shim_supported(lb)) return 0;
  if (shim_.h == NULL || shim_.lb != lb || shim_.mode != LBMI_MODE_FUSED) return 0;
  if (lb->ndist != 1 || visc != NULL || shim_.colloids || shim_openbc_ != 0) return 0;
  if (noise != NULL && noise->on[NOISE_RHO]) return 0;
  if (fe == NULL || fe != fuse_.fe || fe->id != FE_SYMMETRIC ||
      fe->use_stress_relaxation) return 0;
  if (hydro != fuse_.hydro || hydro->nsite != lb->nsite) return 0;
  if (hydro->le && lees_edw_nplane_total(hydro->le) > 0) return 0;
  if (fuse_.phi == NULL || (size_t) fuse_.phi->nsites != (size_t) lb->nsite) return 0;
  cs_cartsz(lb->cs, cartsz);
  cs_periodic(lb->cs, periodic);
  cs_nlocal(lb->cs, nlocal);
  cs_nhalo(lb->cs, &nhalo);
  if (nhalo < 2) return 0;
  for (int ia = 0; ia < 3; ia++) {
    if (cartsz[ia] != 1 || periodic[ia] != 1 || nlocal[ia] < 4) return 0;
  }
  return 1;
}

static int shim_phi_force_calculation_now(pe_t * pe, cs_t * cs, lees_edw_t * le,
					  wall_t * wall, pth_t * pth, fe_t * fe,
					  map_t * map, field_t * phi, hydro_t * hydro);
static int shim_phi_cahn_hilliard_now(phi_ch_t * pch, fe_t * fe, field_t * phi,
				      hydro_t * hydro, map_t * map, noise_t * noise);
static int shim_phi_force_ok(lees_edw_t * le, wall_t * wall, pth_t * pth, fe_t * fe,
			     field_t * phi, hydro_t * hydro, int * npt);
static int shim_phi_ch_ok(phi_ch_t * pch, fe_t * fe, field_t * phi, hydro_t * hydro,
			  map_t * map, noise_t * noise, int * npt, int * order);

/* What has been noted this step, run now and in order: some call came that
 * is not the next one of the sequence */

/* phi and hydro->u of the reference hold the latest interior values again */

static void shim_fuse_settle(void) {
  if (shim_.h == NULL) {
    fuse_.phi_in_q = 0;
    fuse_.u_in_b = 0;
    return;
  }
  if (fuse_.phi_in_q) {
    fuse_.phi_in_q = 0;
    if (fuse_.phi != NULL) {
      SHIM_CHECK(shim_.lb, lbmi_field_interior_copy(shim_.h, 1, fuse_.phinew,
						    shim_field_data(fuse_.phi)));
    }
  }
  if (fuse_.u_in_b) {
    fuse_.u_in_b = 0;
    if (fuse_.hydro != NULL) {
      SHIM_CHECK(shim_.lb, lbmi_field_interior_copy(shim_.h, 3, fuse_.uprev,
						    shim_field_data(fuse_.hydro->u)));
    }
  }
}

static void shim_fuse_flush(void) {
  shim_fuse_settle();
  shim_fuse_flush_lb();
}

static void shim_fuse_flush_lb(void) {
  const int stage = fuse_.stage;
  if (stage == FUSE_NONE) return;
  shim_fuse_settle();                            /* the noted calls read phi, u */
  fuse_.stage = FUSE_NONE;
  fuse_.armed = 0;                               /* a normal step arms again */
  if (stage >= FUSE_HALO) shim_field_halo_now(fuse_.phi);
  if (stage >= FUSE_GRAD) shim_field_grad_compute_now(fuse_.fgrad);
  if (stage >= FUSE_FORCE) {
    shim_phi_force_calculation_now(fuse_.pe, fuse_.cs, fuse_.le, fuse_.wall, fuse_.pth,
				   fuse_.fe, fuse_.map_force, fuse_.phi, fuse_.hydro);
  }
  if (stage >= FUSE_CH) {
    shim_phi_cahn_hilliard_now(fuse_.pch, fuse_.fe, fuse_.phi, fuse_.hydro,
			       fuse_.map_ch, fuse_.noise);
  }
  if (stage >= FUSE_UZERO) shim_hydro_u_zero_now(fuse_.hydro, fuse_.uzero);
}

static int shim_fe_symm_ok(fe_t * fe, field_t * phi, lees_edw_t * le, int * npt) {
  fe_symm_t * fs = (fe_symm_t *) fe;
  if (!shim_fe_wanted() || shim_.h == NULL || shim_.colloids) return 0;
  if (fe == NULL || fe->id != FE_SYMMETRIC || shim_.lb->ndist != 1) return 0;
  if (phi == NULL || fs->phi != phi || phi->nf != 1) return 0;
  if (le != NULL && lees_edw_nplane_total(le) > 0) return 0;
  if (shim_handle_if_any(phi->cs) == NULL) return 0;
  *npt = 0;
  if (fs->dphi->d2 == grad_3d_7pt_fluid_d2) *npt = 7;
  if (fs->dphi->d2 == grad_3d_27pt_fluid_d2) *npt = 27;
  return (*npt != 0 && fs->dphi->level == 2);
}

static int shim_phi_force_ok(lees_edw_t * le, wall_t * wall, pth_t * pth, fe_t * fe,
			     field_t * phi, hydro_t * hydro, int * npt) {
  int is_pm = 0;
  if (hydro == NULL) return 0;
  if (wall) wall_is_pm(wall, &is_pm);
  if (pth->method != FE_FORCE_METHOD_STRESS_DIVERGENCE || is_pm ||
      (wall && wall_present(wall)) || !shim_fe_symm_ok(fe, phi, le, npt)) return 0;
  return 1;
}

int phi_force_calculation(pe_t * pe, cs_t * cs, lees_edw_t * le, wall_t * wall,
			  pth_t * pth, fe_t * fe, map_t * map, field_t * phi,
			  hydro_t * hydro) {
  int npt = 0;
  assert(pth);
  if (fuse_.stage == FUSE_GRAD && phi == fuse_.phi && fe == fuse_.fe &&
      hydro == fuse_.hydro && shim_phi_force_ok(le, wall, pth, fe, phi, hydro, &npt) &&
      ((fe_symm_t *) fe)->dphi == fuse_.fgrad) {
    fuse_.pe = pe; fuse_.cs = cs; fuse_.le = le; fuse_.wall = wall; fuse_.pth = pth;
    fuse_.map_force = map;
    fuse_.stage = FUSE_FORCE;
    return 0;
  }
  shim_fuse_flush();
  return shim_phi_force_calculation_now(pe, cs, le, wall, pth, fe, map, phi, hydro);
}

static int shim_phi_force_calculation_now(pe_t * pe, cs_t * cs, lees_edw_t * le,
					  wall_t * wall, pth_t * pth, fe_t * fe,
					  map_t * map, field_t * phi, hydro_t * hydro) {
  int npt = 0;

  assert(pth);

  if (hydro == NULL) return 0;                               /* phi_force.c:86 */
  if (pth->method == FE_FORCE_METHOD_NO_FORCE) return 0;

  if (!shim_phi_force_ok(le, wall, pth, fe, phi, hydro, &npt)) {
    return phi_force_calculation_ref(pe, cs, le, wall, pth, fe, map, phi, hydro);
  }

  {
    fe_symm_t * fs = (fe_symm_t *) fe;
    fe_symm_param_t param;
    double * grad = NULL;
    double * delsq = NULL;
    fe_symm_param(fs, &param);
    shim_grad_arrays(fs->dphi, &grad, &delsq);
    {
      static int told = 0;
      if (!told) pe_info(pe, "liblbmi: phi_force_calculation bound (LBMI_FE=0: the original)\n");
      told = 1;
    }
    /* F_a -= d_b P_ab at the interior sites, from the arrays the (bound)
     * field_grad_compute has just filled */
    shim_note(S_PHI_FORCE_CALCULATION, 1);
    /* (a candidate for the sequence folded into lb_collide) */
    fuse_.cand |= 1;
    fuse_.phi = phi; fuse_.fe = fe; fuse_.hydro = hydro;
    SHIM_CHECK(shim_.lb, lbmi_symmetric_force(shim_.h, param.a, param.b,
					      param.kappa, shim_field_data(phi),
					      grad, delsq,
					      shim_field_data(hydro->force)));
  }

  return 0;
}

static int shim_phi_ch_ok(phi_ch_t * pch, fe_t * fe, field_t * phi, hydro_t * hydro,
			  map_t * map, noise_t * noise, int * npt, int * order) {
  int noise_phi = 0, ispm = 0;
  double gm[3] = {0.0, 0.0, 0.0};
  physics_t * phys = NULL;
  physics_ref(&phys);
  physics_grad_mu(phys, gm);
  advection_order(order);
  if (noise) noise_present(noise, NOISE_PHI, &noise_phi);
  if (map) map_pm(map, &ispm);
  if (hydro == NULL || noise_phi || ispm || pch->info.conserve != 0 ||
      *order < 1 || *order > 4 || gm[X] != 0.0 || gm[Y] != 0.0 || gm[Z] != 0.0 ||
      !shim_fe_symm_ok(fe, phi, pch->le, npt)) return 0;
  return 1;
}

int phi_cahn_hilliard(phi_ch_t * pch, fe_t * fe, field_t * phi,
		      hydro_t * hydro, map_t * map, noise_t * noise) {
  int npt = 0, order = 0;
  assert(pch);
  assert(fe);
  assert(phi);
  if (fuse_.stage == FUSE_FORCE && pch == fuse_.pch && phi == fuse_.phi &&
      fe == fuse_.fe && hydro == fuse_.hydro &&
      shim_phi_ch_ok(pch, fe, phi, hydro, map, noise, &npt, &order)) {
    fuse_.map_ch = map; fuse_.noise = noise;
    fuse_.stage = FUSE_CH;
    return 0;
  }
  shim_fuse_flush();
  return shim_phi_cahn_hilliard_now(pch, fe, phi, hydro, map, noise);
}

static int shim_phi_cahn_hilliard_now(phi_ch_t * pch, fe_t * fe, field_t * phi,
				      hydro_t * hydro, map_t * map, noise_t * noise) {
  static double * scratch = NULL;        /* the new phi, nsite doubles (device) */
  static size_t scratch_sites = 0;
  int npt = 0, order = 0;
  physics_t * phys = NULL;

  assert(pch);
  assert(fe);
  assert(phi);

  physics_ref(&phys);

  if (!shim_phi_ch_ok(pch, fe, phi, hydro, map, noise, &npt, &order)) {
    return phi_cahn_hilliard_ref(pch, fe, phi, hydro, map, noise);
  }

  {
    fe_symm_t * fs = (fe_symm_t *) fe;
    fe_symm_param_t param;
    double mobility = 0.0;
    double * grad = NULL;
    double * delsq = NULL;
    double * phid = shim_field_data(phi);
    double * u = shim_field_data(hydro->u);
    size_t nsites = (size_t) phi->nsites;

    fe_symm_param(fs, &param);
    physics_mobility(phys, &mobility);
    shim_grad_arrays(fs->dphi, &grad, &delsq);
    if (scratch_sites < nsites) {
      if (scratch) tdpAssert(tdpFree(scratch));
      tdpAssert(tdpMalloc((void **) &scratch, sizeof(double)*nsites));
      scratch_sites = nsites;
    }
    {
      static int told = 0;
      if (!told) pe_info(pch->pe, "liblbmi: phi_cahn_hilliard bound (LBMI_FE=0: the original)\n");
      told = 1;
    }
    SHIM_CHECK(shim_.lb, lbmi_fe_scheme_set(shim_.h, npt, order));
    /* hydro_u_halo (phi_cahn_hilliard.c:224), then advection, diffusive
     * flux and forward step in one kernel; the reference updates phi in
     * place, so the new interior goes back into its array */
    SHIM_CHECK(shim_.lb, lbmi_field_halo_n(shim_.h, 3, 1, u));
    shim_note(S_PHI_CAHN_HILLIARD, 1);
    if (phi == fuse_.phi && fe == fuse_.fe && hydro == fuse_.hydro) {
      fuse_.cand |= 2;
      fuse_.pch = pch;
    }
    SHIM_CHECK(shim_.lb, lbmi_cahn_hilliard(shim_.h, param.a, param.b,
					    param.kappa, mobility, phid, delsq,
					    u, scratch));
    SHIM_CHECK(shim_.lb, lbmi_field_interior_copy(shim_.h, 1, scratch, phid));
  }

  return 0;
}

/*****************************************************************************
 *
 *  cahn_hilliard_stats, cahn_hilliard_stats_time0
 *  (cahn_hilliard_stats.c:52-215)
 *
 *  The sum, variance and extrema of phi that ludwig.c prints at every
 *  statistics step (and phi->field_init_sum at start-up). The reference
 *  collects them with three kernels whose blocks take turns under a
 *  compare-and-swap lock: 4.3 ms per call at 32^3, about 5 s at 128^3 -- ten of
 *  the sixteen seconds the 100-step droplet run takes with everything else
 *  bound (profiles/r02_app_timing.txt). Here: one pass and a tree reduction
 *  (lbmi_field_stats), Kahan-compensated as there. Not for the "conserve"
 *  variants (a doubly compensated sum against a correction field).
 *
 *  The first call comes at start-up (ludwig.c:429), before an lb_t has
 *  passed through this file: a handle of its own, made from the coordinate
 *  system of phi, serves the calls that need no distributions.
 *
 *****************************************************************************/

static lbmi_t * shim_field_handle(cs_t * cs) {
  static lbmi_t * aux = NULL;
  static int aux_nlocal[3] = {0, 0, 0};
  lbmi_t * h = shim_handle_if_any(cs);
  int nlocal[3];
  if (h != NULL) return h;
  if (DATA_MODEL != DATA_MODEL_SOA) return NULL;
  cs_nlocal(cs, nlocal);
  if (aux != NULL && (nlocal[X] != aux_nlocal[X] || nlocal[Y] != aux_nlocal[Y] ||
		      nlocal[Z] != aux_nlocal[Z])) {
    lbmi_free(aux);
    aux = NULL;
  }
  if (aux == NULL) {
    lbmi_options_t opts;
    lbmi_options_default(&opts);
    cs_nlocal(cs, opts.nlocal);
    cs_nhalo(cs, &opts.nhalo);
    if (lbmi_create(&opts, &aux) != 0) return NULL;    /* the original will do */
    if (lbmi_set_stream(aux, NULL) != 0) {
      lbmi_free(aux);
      aux = NULL;
      return NULL;
    }
    cs_nlocal(cs, aux_nlocal);
  }
  return aux;
}

static int shim_phi_stats(phi_ch_t * pch, field_t * phi, map_t * map,
			  double stats[5]) {
  lbmi_t * h = NULL;
  double local[5];
  MPI_Comm comm;
  if (pch->info.conserve != 0 || phi->nf != 1) return 0;
  if (phi->le && lees_edw_nplane_total(phi->le) > 0) return 0;
  h = shim_field_handle(phi->cs);
  if (h == NULL) return 0;
  if (lbmi_field_stats(h, shim_field_data(phi),
		       (char *) shim_cached(map, &map->target->status, sizeof(char *)),
		       local) != 0) {
    pe_fatal(pch->pe, "liblbmi: %s (%s:%d)\n", lbmi_last_error(), __FILE__, __LINE__);
  }
  pe_mpi_comm(pch->pe, &comm);
  /* (across ranks a plain sum of the compensated local sums; the reference
   * merges the compensation terms too, cahn_hilliard_stats.c:187-199) */
  MPI_Reduce(local + 0, stats + 0, 3, MPI_DOUBLE, MPI_SUM, 0, comm);
  MPI_Reduce(local + 3, stats + 3, 1, MPI_DOUBLE, MPI_MIN, 0, comm);
  MPI_Reduce(local + 4, stats + 4, 1, MPI_DOUBLE, MPI_MAX, 0, comm);
  return 1;
}

int cahn_hilliard_stats_time0(phi_ch_t * pch, field_t * phi, map_t * map) {
  shim_fuse_flush();
  double stats[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  MPI_Comm comm;
  assert(pch);
  assert(phi);
  assert(map);
  if (!shim_phi_stats(pch, phi, map, stats)) {
    return cahn_hilliard_stats_time0_ref(pch, phi, map);
  }
  shim_note(S_CAHN_HILLIARD_STATS_TIME0, 1);
  pe_mpi_comm(pch->pe, &comm);
  phi->field_init_sum = stats[1];
  MPI_Bcast(&phi->field_init_sum, 1, MPI_DOUBLE, 0, comm);
  return 0;
}

int cahn_hilliard_stats(phi_ch_t * pch, field_t * phi, map_t * map) {
  shim_fuse_flush();
  double stats[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  assert(pch);
  assert(phi);
  assert(map);
  if (!shim_phi_stats(pch, phi, map, stats)) {
    return cahn_hilliard_stats_ref(pch, phi, map);
  }
  shim_note(S_CAHN_HILLIARD_STATS, 1);
  {
    double rvol = 1.0/stats[0];
    double fbar = rvol*stats[1];                 /* mean */
    double fvar = rvol*stats[2] - fbar*fbar;     /* variance */
    pe_info(pch->pe, "[phi] %14.7e %14.7e%14.7e %14.7e%14.7e\n",
	    stats[1], fbar, fvar, stats[3], stats[4]);
  }
  return 0;
}
