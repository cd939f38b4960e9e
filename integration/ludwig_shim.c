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
 *  (and wall.c with -Dwall_bbl=wall_bbl_ref, phi_lb_coupler.c with
 *  -Dphi_lb_to_field=phi_lb_to_field_ref, hydro.c with -Dhydro_u_zero=
 *  hydro_u_zero_ref -Dhydro_f_zero=hydro_f_zero_ref, field.c with -Dfield_halo=
 *  field_halo_ref, field_grad.c with -Dfield_grad_compute=
 *  field_grad_compute_ref) so that their originals remain
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
} shim_t;

static shim_t shim_ = {NULL, NULL, LBMI_MODE_EAGER};

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
 * blocking copy that drains the stream: remember them (objects of these
 * types live as long as the run). */

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
    /* LBMI_MODE = eager (default: f as the reference after every call),
     * halo (collision and halo as observable as in eager, propagation
     * deferred: walls and colloids are fine) or fused (halo swap and
     * propagation deferred: nothing may touch f between lb_collide and
     * lb_propagation) */
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
    opts.mode = LBMI_MODE_EAGER;
    if (mode && mode[0] == 'f' && lb->ndist == 1) opts.mode = LBMI_MODE_FUSED;
    if (mode && mode[0] == 'h') opts.mode = LBMI_MODE_FUSED_HALO;  /* ndist 1 or 2 */

    SHIM_CHECK(lb, lbmi_create(&opts, &shim_.h));
    /* Ludwig launches all its kernels on the default stream
     * (tdpLaunchKernel(..., 0, 0, ...)): run ours there too, so that the
     * kernels on either side of each call are ordered without extra syncs */
    SHIM_CHECK(lb, lbmi_set_stream(shim_.h, NULL));
    shim_.lb = lb;
    shim_.mode = opts.mode;

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
    lb_collision_relaxation_times_set(lb);       /* keeps lb->param current */
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
 *  wall_bbl  (wall.c:960-989): bounce-back on the reference's own links
 *
 *  The link arrays and the momentum accumulator are members of the DEVICE
 *  copy of wall_t; the kernel works on them where they are.
 *
 *****************************************************************************/

int wall_bbl(wall_t * wall) {

  assert(wall);
  assert(wall->target);

  if (wall->nlink == 0) return 0;                /* wall.c:967 */

  if (!shim_supported(wall->lb)) {
    return wall_bbl_ref(wall);
  }
  if (shim_.mode == LBMI_MODE_FUSED) {
    pe_fatal(wall->pe, "liblbmi: LBMI_MODE=fused cannot be used with walls "
	     "(bounce-back acts between lb_halo and lb_propagation): use "
	     "LBMI_MODE=halo\n");
  }

  {
    int * link[4] = {NULL, NULL, NULL, NULL};
    double * fnet = (double *) ((char *) wall->target + offsetof(wall_t, fnet));
    tdpAssert(tdpMemcpy(&link[0], &wall->target->linki, sizeof(int *),
			tdpMemcpyDeviceToHost));
    tdpAssert(tdpMemcpy(&link[1], &wall->target->linkj, sizeof(int *),
			tdpMemcpyDeviceToHost));
    tdpAssert(tdpMemcpy(&link[2], &wall->target->linkp, sizeof(int *),
			tdpMemcpyDeviceToHost));
    tdpAssert(tdpMemcpy(&link[3], &wall->target->linku, sizeof(int *),
			tdpMemcpyDeviceToHost));
    {
      /* the kernels test map->status[i] for MAP_COLLOID (wall.c:1046, 1146) */
      char * status = NULL;
      tdpAssert(tdpMemcpy(&status, &wall->map->target->status, sizeof(char *),
			  tdpMemcpyDeviceToHost));
      SHIM_CHECK(wall->lb, lbmi_wall_status_set(shim_handle(wall->lb), status));
    }
    if (wall->param->slip.active) {                       /* wall.c:971 */
      int * linkk = NULL;
      int8_t * linkq = NULL;
      int8_t * links = NULL;
      tdpAssert(tdpMemcpy(&linkk, &wall->target->linkk, sizeof(int *),
			  tdpMemcpyDeviceToHost));
      tdpAssert(tdpMemcpy(&linkq, &wall->target->linkq, sizeof(int8_t *),
			  tdpMemcpyDeviceToHost));
      tdpAssert(tdpMemcpy(&links, &wall->target->links, sizeof(int8_t *),
			  tdpMemcpyDeviceToHost));
      SHIM_CHECK(wall->lb,
		 lbmi_wall_bbl_slip_arrays(shim_handle(wall->lb), wall->nlink,
					   link[0], link[1], link[2], linkk,
					   (const signed char *) linkq,
					   (const signed char *) links,
					   wall->param->slip.s, fnet));
    }
    else {
      SHIM_CHECK(wall->lb, lbmi_wall_bbl_arrays(shim_handle(wall->lb),
						 wall->nlink, link[0], link[1],
						 link[2], link[3],
						 wall->param->ubot,
						 wall->param->utop, fnet));
    }
  }

  return 0;
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
