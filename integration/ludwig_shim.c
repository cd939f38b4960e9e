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
 *      field_halo()      field.h:96        (field.c; FIELD_HALO_TARGET only)
  *      field_grad_compute() field_grad.h:49 (3d_7pt_fluid / 3d_27pt_fluid d2)
 *      wall_set_wall_distributions() wall.h:100, bounce_back_on_links()
 *                        bbl.h:26: the originals, after making sure that the
 *                        distributions they work on are the reference's
 *      lb_free(), field_free(), map_free(), wall_free(): the originals, after
 *                        forgetting what this file remembers of the object
 *
 *  by unpacking lb_t / hydro_t / map_t and calling the C-ABI of
 *  include/lbmi.h. The other contents of collision.c / model.c /
 *  propagation.c (relaxation setters, moments, I/O, ...) stay as they are:
 *  the three reference files are compiled with
 *
 *      -Dlb_collide=lb_collide_ref -Dlb_halo=lb_halo_ref
 *      -Dlb_halo_swap=lb_halo_swap_ref -Dlb_propagation=lb_propagation_ref
 *      -Dlb_memcpy=lb_memcpy_ref -Dlb_io_write=lb_io_write_ref
 *      -Dlb_io_read=lb_io_read_ref
 *
  *      -Dlb_free=lb_free_ref
 *
 *  (and wall.c with -Dwall_bbl=wall_bbl_ref -Dwall_set_wall_distributions=
 *  wall_set_wall_distributions_ref -Dwall_free=wall_free_ref, bbl.c with
 *  -Dbounce_back_on_links=bounce_back_on_links_ref, map.c with -Dmap_free=
 *  map_free_ref, phi_lb_coupler.c with
 *  -Dphi_lb_to_field=phi_lb_to_field_ref, hydro.c with -Dhydro_u_zero=
 *  hydro_u_zero_ref -Dhydro_f_zero=hydro_f_zero_ref, field.c with -Dfield_halo=
 *  field_halo_ref -Dfield_free=field_free_ref, field_grad.c with
 *  -Dfield_grad_compute=field_grad_compute_ref) so that their originals remain
 *  available as fall-backs (colloids, Lees-Edwards, host halo
 *  schemes, noise), and this file is compiled
 *  with the same -D_D3Q19_|-D_D3Q27_ -DADDR_SOA as the rest of libludwig.a
 *  and linked with -llbmi. See INTEGRATION.md.
 *
 *  This file is compile-checked against the reference headers by
 *  __graft_entry__.build() when /root/reference is present (gcc
 *  -fsyntax-only); it cannot be linked or run in this repository because
 *  the reference tree is not part of it.
 *
 *****************************************************************************/

#include <assert.h>
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
int phi_lb_to_field_ref(field_t * phi, lb_t * lb);
int hydro_u_zero_ref(hydro_t * hydro, const double uzero[3]);
int hydro_f_zero_ref(hydro_t * hydro, const double fzero[3]);
int field_halo_ref(field_t * field);
int field_grad_compute_ref(field_grad_t * fgrad);

/* One liblbmi handle per lb_t (Ludwig has one lb_t per rank) */

typedef struct shim_s {
  lb_t * lb;
  lbmi_t * h;
  int mode;                       /* lbmi_mode_t in use */
  wall_t * wall;                  /* whose links the handle holds a copy of */
  int wall_nlink;
  int colloids;                   /* bounce_back_on_links has seen colloids */
  int param_valid;                /* param_committed is what the device has */
  lb_collide_param_t param_committed;
} shim_t;

static shim_t shim_;              /* zero: no handle, LBMI_MODE_EAGER */

#define SHIM_CHECK(lb, call)						\
  do {									\
    int ifail_ = (call);						\
    if (ifail_ != 0) {							\
      pe_fatal((lb)->pe, "liblbmi: %s (%s:%d)\n", lbmi_last_error(),	\
	       __FILE__, __LINE__);					\
    }									\
  } while (0)

/* Can liblbmi take this lb_t? SoA build, device halo scheme, decomposition
 * along X only. Anything else uses the originals. */

static int shim_supported(lb_t * lb) {
  int cartsz[3];
  if (DATA_MODEL != DATA_MODEL_SOA) return 0;
  if (lb->ndist != 1 && lb->ndist != 2) return 0;
  if (lb->model.nvel != 19 && lb->model.nvel != 27) return 0;
  if (lb->haloscheme != LB_HALO_TARGET) return 0;
  cs_cartsz(lb->cs, cartsz);
  if (cartsz[Y] != 1 || cartsz[Z] != 1) return 0;
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

static double * shim_field_data(field_t * field) {
  return (double *) shim_cached(field, &field->target->data, sizeof(double *));
}

/* After any call that swaps f and fprime, make lb->target->f/fprime point
 * at the current arrays, so that foreign kernels (wall.c:930-950,
 * bbl.c:294-360, stats_distribution.c:322) keep working. */

static double * last_f = NULL;         /* what lb->target holds now */
static double * last_fprime = NULL;

static void shim_sync_pointers(lb_t * lb, lbmi_t * h) {
  double * f = NULL;
  double * fprime = NULL;
  SHIM_CHECK(lb, lbmi_lb_pointers(h, &f, &fprime));
  if (f == last_f && fprime == last_fprime) return;    /* nothing swapped */
  tdpAssert(tdpMemcpy(&lb->target->f, &f, sizeof(double *),
		      tdpMemcpyHostToDevice));
  tdpAssert(tdpMemcpy(&lb->target->fprime, &fprime, sizeof(double *),
		      tdpMemcpyHostToDevice));
  last_f = f;
  last_fprime = fprime;
}

static lbmi_t * shim_handle(lb_t * lb) {

  if (shim_.h != NULL && shim_.lb == lb) return shim_.h;
  if (shim_.h != NULL) lbmi_free(shim_.h);

  {
    lbmi_options_t opts;
    int cartsz[3], coords[3];
    double * f = NULL;
    double * fprime = NULL;
        /* LBMI_MODE = halo (the default: after lb_collide and lb_halo f is the
     * reference's, so walls, colloids and anything else that acts between
     * lb_halo and lb_propagation find what they expect; only the propagation
     * is deferred into the next collision, and every reader this file knows
     * of -- lb_memcpy, the statistics, lb_io_write, phi_lb_to_field -- flushes
     * it first), eager (f as the reference after every call, three passes
     * over f per step) or fused (halo swap and propagation both deferred:
     * nothing may touch f between lb_collide and lb_propagation; a run that
     * turns out to have wall links or colloids drops to halo at the first
     * wall_set_wall_distributions / bounce_back_on_links that would see it) */
    const char * mode = getenv("LBMI_MODE");

    lbmi_options_default(&opts);
    opts.nvel = lb->model.nvel;
    opts.ndist = lb->ndist;
    cs_nlocal(lb->cs, opts.nlocal);
    cs_nhalo(lb->cs, &opts.nhalo);
    cs_cartsz(lb->cs, cartsz);
    cs_cart_coords(lb->cs, coords);
    opts.cartsz = cartsz[X];
    opts.cartrank = coords[X];
    opts.device = -1;                            /* ludwig.c:467-492 chose it */
    opts.halo_scheme = LBMI_HALO_FULL;           /* halo_swap_packed semantics */
        opts.mode = LBMI_MODE_FUSED_HALO;              /* ndist 1 or 2 */
    if (mode && mode[0] == 'e') opts.mode = LBMI_MODE_EAGER;
    if (mode && mode[0] == 'f' && lb->ndist == 1) opts.mode = LBMI_MODE_FUSED;
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

    shim_device_f(lb, &f, &fprime);
    SHIM_CHECK(lb, lbmi_lb_bind(shim_.h, f, fprime));
    last_f = f;
    last_fprime = fprime;

    if (cartsz[X] > 1) {
      /* ncclUniqueId from rank 0 of the Cartesian communicator */
      char id[LBMI_UNIQUE_ID_BYTES];
      MPI_Comm comm;
      int rank;
      cs_cart_comm(lb->cs, &comm);
      MPI_Comm_rank(comm, &rank);
      if (rank == 0) SHIM_CHECK(lb, lbmi_comm_unique_id(id));
      MPI_Bcast(id, LBMI_UNIQUE_ID_BYTES, MPI_BYTE, 0, comm);
      SHIM_CHECK(lb, lbmi_comm_init(shim_.h, id));
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

  /* Not covered by liblbmi: fluctuations, stress relaxation with a free
   * energy other than the symmetric one; two
   * distributions only with the symmetric free energy (as the reference,
   * collision.c:160) */
  if (!shim_supported(lb) || noise->on[NOISE_RHO] ||
      (visc != NULL && lb->ndist != 1) ||
      (fe && fe->use_stress_relaxation && fe->id != FE_SYMMETRIC) ||
      (lb->ndist == 2 && (fe == NULL || fe->id != FE_SYMMETRIC))) {
    if (shim_.h && shim_.lb == lb) {
      SHIM_CHECK(lb, lbmi_lb_flush(shim_.h));
      shim_sync_pointers(lb, shim_.h);
    }
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

    status = (char *) shim_cached(map, &map->target->status, sizeof(char *));
    hy.force  = shim_field_data(hydro->force);
    hy.status = status;
    hy.rho    = shim_field_data(hydro->rho);
    hy.u      = shim_field_data(hydro->u);
    /* a viscosity model has left the local viscosity in hydro->eta
     * (collision.c:386-404, 1947) */
    hy.eta    = visc ? shim_field_data(hydro->eta) : NULL;

    /* hydro_f_zero below tells the library that hydro->force holds zeros,
     * and a force field of zeros is not read. That holds until somebody
     * writes to it, and everybody who does inside the reference --
     * phi_force_calculation, fe_lc_droplet_bodyforce, psi_force_*,
     * nernst_planck_driver (all need a free energy, ludwig.c:643-738) and
     * subgrid_force_from_particles (colloids, ludwig.c:2071, 2149) -- is
     * outside the library: with a free energy or colloids the force is taken
     * to have been written every step. */
    if (fe != NULL || shim_.colloids) {
      SHIM_CHECK(lb, lbmi_hydro_field_dirty(h, hy.force));
    }

    if (lb->ndist == 2 || (fe && fe->use_stress_relaxation)) {
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
      bin.phi = shim_field_data(fs->phi);
      tdpAssert(tdpMemcpy(&bin.grad, &fs->dphi->target->grad, sizeof(double *),
			  tdpMemcpyDeviceToHost));
      tdpAssert(tdpMemcpy(&bin.delsq, &fs->dphi->target->delsq,
			  sizeof(double *), tdpMemcpyDeviceToHost));
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
    shim_sync_pointers(lb, h);                   /* FUSED swaps here */
  }

  return 0;
}

/*****************************************************************************
 *
 *  phi_lb_to_field  (phi_lb_coupler.c:39-64)
 *
 *****************************************************************************/

int phi_lb_to_field(field_t * phi, lb_t * lb) {

  assert(phi);
  assert(lb);

  if (!shim_supported(lb) || lb->ndist != 2) return phi_lb_to_field_ref(phi, lb);

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
  pe_info(lb->pe, "liblbmi: %s acts on the distributions between lb_halo and "
	  "lb_propagation: LBMI_MODE=fused -> halo\n", who);
  SHIM_CHECK(lb, lbmi_lb_mode_set(shim_.h, LBMI_MODE_FUSED_HALO));
  shim_.mode = LBMI_MODE_FUSED_HALO;
  shim_sync_pointers(lb, shim_.h);
}

int wall_bbl(wall_t * wall) {

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

  assert(wall);

  if (wall->nlink == 0) return 0;                /* wall.c:907 */
  shim_needs_canonical_f(wall->lb, "wall_set_wall_distributions");

  return wall_set_wall_distributions_ref(wall);
}

int bounce_back_on_links(bbl_t * bbl, lb_t * lb, wall_t * wall,
			 colloids_info_t * cinfo) {
  int ntotal = 0;

  assert(lb);
  assert(cinfo);

  colloids_info_ntotal(cinfo, &ntotal);
  if (ntotal > 0) {
    shim_.colloids = 1;
    shim_needs_canonical_f(lb, "bounce_back_on_links");
  }

  return bounce_back_on_links_ref(bbl, lb, wall, cinfo);
}

/*****************************************************************************
 *
 *  lb_free, field_free, map_free, wall_free: the originals, after dropping
 *  what this file remembers under the object's address
 *
 *****************************************************************************/

int lb_free(lb_t * lb) {
  assert(lb);
  if (shim_.lb == lb) {
    /* the handle borrows lb->target->f / fprime: it goes first */
    if (shim_.h) lbmi_free(shim_.h);
    memset(&shim_, 0, sizeof(shim_));
    last_f = NULL;
    last_fprime = NULL;
  }
  return lb_free_ref(lb);
}

int field_free(field_t * obj) {
  shim_forget(obj);
  return field_free_ref(obj);
}

int map_free(map_t * obj) {
  shim_forget(obj);
  return map_free_ref(obj);
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

  assert(lb);

  if (!shim_supported(lb) || flag != LB_HALO_TARGET) {
    return lb_halo_swap_ref(lb, flag);
  }

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

  assert(lb);

  if (!shim_supported(lb)) return lb_propagation_ref(lb);

  {
    lbmi_t * h = shim_handle(lb);
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

  assert(lb);

  if (shim_.h && shim_.lb == lb) {
    SHIM_CHECK(lb, lbmi_lb_flush(shim_.h));
    SHIM_CHECK(lb, lbmi_synchronize(shim_.h));
    shim_sync_pointers(lb, shim_.h);
  }

  return lb_memcpy_ref(lb, flag);
}


/*****************************************************************************
 *
 *  lb_io_write, lb_io_read  (model.c:1568-1649), row f3
 *
 *  The reference copies all of f to the host, packs the records there and
 *  writes through MPI-IO. In its MPI-IO mode with one file (i/o grid 1_1_1)
 *  and binary records, an X slab is one contiguous byte range of that file:
 *  the records are packed on the device and written from there, the files
 *  are the same byte for byte. Anything else (old-style i/o, several files,
 *  ASCII records) goes to the original.
 *
 *****************************************************************************/

static int shim_io_supported(lb_t * lb, const io_metadata_t * meta) {
  if (!shim_supported(lb)) return 0;
  if (meta->options.mode != IO_MODE_MPIIO) return 0;
  if (meta->options.iorformat != IO_RECORD_BINARY) return 0;
  if (meta->options.iogrid[X] != 1 || meta->options.iogrid[Y] != 1 ||
      meta->options.iogrid[Z] != 1) return 0;
  return 1;
}

int lb_io_write(lb_t * lb, int timestep, io_event_t * event) {

  assert(lb);
  assert(event);

  if (!shim_io_supported(lb, &lb->output)) {
    return lb_io_write_ref(lb, timestep, event);   /* via lb_memcpy: flushes */
  }

  {
    int ntotal[3], noffset[3];
    cs_ntotal(lb->cs, ntotal);
    cs_nlocal_offset(lb->cs, noffset);
    io_event_record(event, IO_EVENT_AGGR);
    io_event_record(event, IO_EVENT_WRITE);
    /* the metadata file (rank at offset 0) and this rank's byte range */
    SHIM_CHECK(lb, lbmi_lb_io_write(shim_handle(lb), ".", timestep, ntotal[X],
				    noffset[X]));
    shim_sync_pointers(lb, shim_.h);                /* a flush may have swapped */
    lb->output.iswriten = 1;
    io_event_report(event, &lb->output, "dist");
  }

  return 0;
}

int lb_io_read(lb_t * lb, int timestep, io_event_t * event) {

  assert(lb);
  assert(event);

  if (!shim_io_supported(lb, &lb->input)) {
    if (shim_.h && shim_.lb == lb) {
      /* the original fills the HOST copy; nothing deferred may survive it */
      SHIM_CHECK(lb, lbmi_lb_flush(shim_.h));
      shim_sync_pointers(lb, shim_.h);
    }
    return lb_io_read_ref(lb, timestep, event);
  }

  {
    int ntotal[3], noffset[3];
    cs_ntotal(lb->cs, ntotal);
    cs_nlocal_offset(lb->cs, noffset);
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

int hydro_u_zero(hydro_t * hydro, const double uzero[3]) {
  lbmi_t * h = NULL;
  assert(hydro);
  h = shim_handle_if_any(hydro->cs);
  if (h == NULL) return hydro_u_zero_ref(hydro, uzero);
  SHIM_CHECK(shim_.lb, lbmi_hydro_field_set(h, shim_field_data(hydro->u), 3, uzero));
  return 0;
}

int hydro_f_zero(hydro_t * hydro, const double fzero[3]) {
  lbmi_t * h = NULL;
  assert(hydro);
  h = shim_handle_if_any(hydro->cs);
  if (h == NULL) return hydro_f_zero_ref(hydro, fzero);
  SHIM_CHECK(shim_.lb, lbmi_hydro_field_set(h, shim_field_data(hydro->force), 3, fzero));
  return 0;
}

/* field_halo: the device scheme, no Lees-Edwards planes, halo width within
 * the halo of the lattice (hydro_u_halo comes through here as well,
 * hydro.c:190-197) */

int field_halo(field_t * field) {
  lbmi_t * h = NULL;
  int nhalo = 0;
  assert(field);
  h = shim_handle_if_any(field->cs);
  cs_nhalo(field->cs, &nhalo);
  if (h == NULL || field->opts.haloscheme != FIELD_HALO_TARGET ||
      (field->le && lees_edw_nplane_total(field->le) > 0) ||
      field->nhcomm < 1 || field->nhcomm > nhalo || field->nf > 27) {
    return field_halo_ref(field);
  }
  SHIM_CHECK(shim_.lb, lbmi_field_halo_n(h, field->nf, field->nhcomm,
					 shim_field_data(field)));
  return 0;
}

/* field_grad_compute for a scalar with the fluid-only 7- or 27-point
 * stencils at level 2 (grad and delsq); anything else is the original */

int field_grad_compute(field_grad_t * fgrad) {
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
  tdpAssert(tdpMemcpy(&grad, &fgrad->target->grad, sizeof(double *),
		      tdpMemcpyDeviceToHost));
  tdpAssert(tdpMemcpy(&delsq, &fgrad->target->delsq, sizeof(double *),
		      tdpMemcpyDeviceToHost));
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
