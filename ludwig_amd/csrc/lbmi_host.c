/*****************************************************************************
 *
 *  lbmi_host.c
 *
 *  ANSI C (C99) host side of liblbmi: the handle (the part of lb_t and
 *  halo_swap_t the hot path needs), parameter handling, the eager and the
 *  fused (deferred) execution of lb_collide / lb_halo / lb_propagation,
 *  and the RCCL ring for the slab decomposition. All compute is done by the
 *  HIP kernels of lbmi_kernels.hip; there is no CPU fallback.
 *
 *  Reference counterparts are cited at each function; see include/lbmi.h.
 *
 *****************************************************************************/

#define __HIP_PLATFORM_AMD__ 1

#include <errno.h>
#include <fcntl.h>
#include <pthread.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include "lbmi.h"
#include "lbmi_kernels.h"

enum {X = 0, Y = 1, Z = 2};
enum {LBMI_NEVENT = 4096, LBMI_NDETAIL = 256};
enum {LBMI_XOPS_MAX = 4*LBMI_NVEL_MAX};

struct lbmi_s {
  lbmi_options_t opts;
  lbmi_kparam_t kp;
  int device;

  int8_t cv[LBMI_NVEL_MAX][3];
  lbmi_halo_sel_t sel_full[3];       /* all populations, both sides */
  lbmi_halo_sel_t sel_reduced[3];    /* c_d = +1 low side, c_d = -1 high */

  hipStream_t stream;                /* compute (own_stream, or the caller's) */
  hipStream_t own_stream;
  hipStream_t comm_stream;           /* halo exchange (multi-GPU overlap) */
  hipStream_t bnd_stream;            /* boundary planes of a slab (FUSED) */
  hipEvent_t ev_bnd;                 /* boundary planes of fprime written */
  int x_concurrent;                  /* 1: boundary planes beside the interior */
  int eager_oop;                     /* EAGER lb_collide out of place + swap (default) */
  hipEvent_t ev_ready;               /* boundary planes of f written */
  hipEvent_t ev_halo;                /* x halo planes of f filled */

  /* lb_t-like state */
  double * f;
  double * fprime;
  int owns_f;
  int pending_halo;                  /* FUSED: lb_halo recorded */
  int pending_prop;                  /* FUSED: lb_propagation recorded */
  int layout_swapped;                /* INPLACE: population p lives in slot opp(p) */
  int early_prop;                    /* INPLACE: k_aa_odd already propagated this step */
  int halo_seen;                     /* INPLACE: lb_halo called while early_prop */
  int halo_done;                     /* FUSED_HALO: the pending halo is in f already */
  int nt_store_mode;                 /* -1 auto, else the lbmi_tune value */
  int use_blocked;                   /* FUSED, 1 GPU: keep the deferred state in
					the blocked order (lbmi_tune "blocked") */
  int blocked;                       /* f is in the blocked order right now */

    /* hydro arrays: what is known to hold zeros (lbmi_hydro_field_set), and
   * rho, u of the last collision owed to the caller (lbmi_tune "hydro_lazy") */
  const void * known_zero[4];
  int hydro_lazy;
  unsigned int * noise_state;        /* lbmi_noise_set: the reference's generator state */
  long long noise_stride;
  double noise_kt;
  int noise_ghosts;
  int hydro_stale;
  lbmi_hydro_dev_t lazy_h;
  double lazy_fbody[3];

  /* moments workspace */
  double * mom_work;
  double * mom_out;

  /* the ring of X slabs: RCCL (one process per GPU), or an in-process ring
   * of handles on one device (lbmi_ring_t: tests, rehearsals) */
  ncclComm_t comm;
  lbmi_ring_t * ring;
  unsigned long long peers_enabled;  /* devices this one has enabled peer access to */
  int have_comm;
  double * sendlo, * sendhi, * recvlo, * recvhi;   /* staging, any halo swap */
  size_t xbuf_doubles;
  int xdim;                          /* the decomposed direction: X (0) unless
					opts.cartdim says Y (1) or Z (2); with
					LBMI_CART_GENERAL the first decomposed one */
  int csz[3];                        /* ranks along X, Y, Z */
  int ccoord[3];                     /* this rank's coordinates */
  int multi;                         /* more than one direction decomposed */
  int x_packed;                      /* 1: pack/unpack through buffers */
  /* FUSED step: buffers of its own (a field halo between two steps must not
   * disturb what the boundary launch left for the next exchange) */
  double * fx[4];                    /* sendlo, sendhi, recvlo, recvhi */
  int x_direct;                      /* boundary launch works on fx directly */
  int xsend_valid;                   /* fx send buffers hold the planes of f */
  int halo_fold;                     /* FUSED_HALO, one rank: the step's kernel computes
					the halo shell of its result (lbmi_tune) */
  int halo_fresh;                    /* ... and it did: f has its halo already */

  /* free-energy sector: gradient stencil (7 | 27), advection order (1..4) */
  int grad_npt;
  int adv_order;
  int io_ascii;                      /* distribution files: text records */
  lbmi_io_file_t io_file;            /* i/o grid {nfile,1,1}: this rank's file; periodicity */
  int io_file_set;
  double * fe_force;                 /* lbmi_symmetric_lb_step off the fused
					route: the thermodynamic force */

  /* walls: links (device), their host copy, momentum accounting */
  int nlink;
  int * link_dev[4];                 /* i, j, p, u */
  int * link_host[4];
  double * wall_part;                /* per-block momentum partials */
  int wall_part_nblk;
  double * wall_fnet;                /* device accumulator, 3 doubles */
  double wall_ubot[3];
  double wall_utop[3];
  const char * wall_status;          /* device map for the MAP_COLLOID test */
  double * wall_fnet_ext;            /* lbmi_wall_fnet_bind: the caller's accumulator */
  int * wall_err;                    /* pinned, mapped: index + 1 of a link record
					the kernels refused (0: none) */
  const void * wall_seen[2];         /* caller-owned link arrays already checked */
  int wall_seen_nlink;
  /* lbmi_lb_run: two steady-state steps captured in a hipGraph */
  int use_graph;
  hipGraphExec_t run_graph;
  hipEvent_t ev_graph;
  struct {
    const double * f;
    const double * fprime;
    lbmi_hydro_dev_t h;
    lbmi_kparam_t kp;
    int nt_store_mode;
    int hydro_lazy;
    hipStream_t stream;
  } run_key;
  int slip_active;                   /* wall_slip_t */
  double slip_s[19];
  int * slip_k_dev;                  /* linkk, linkq, links (device) */
  int8_t * slip_q_dev;
  int8_t * slip_s_dev;
  int * slip_host[3];                /* host copies, as ints */
  double rho0;

  /* kernel timing */
  int timing;                        /* 0 off, k: every k-th launch */
  int timing_count;
  int timing_now;
  int nev;
  hipEvent_t ev0[LBMI_NEVENT];
  hipEvent_t ev1[LBMI_NEVENT];
  int ev_created;
  double ms_accum;
  int launches_accum;
  /* slab step in detail: interior launch, exchange, boundary launch */
  hipEvent_t evd[6][LBMI_NDETAIL];
  int nd;
};

static __thread char lbmi_errbuf[512] = "no error";

static int lbmi_fail(int code, const char * fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(lbmi_errbuf, sizeof(lbmi_errbuf), fmt, ap);
  va_end(ap);
  return code;
}

const char * lbmi_last_error(void) {
  return lbmi_errbuf;
}

#define HIPCHECK(call)							\
  do {									\
    hipError_t e_ = (call);						\
    if (e_ != hipSuccess) {						\
      return lbmi_fail(LBMI_ERR_HIP, "%s:%d: %s: %s", __FILE__, __LINE__, \
		       #call, hipGetErrorString(e_));			\
    }									\
  } while (0)

#define KCHECK(call)							\
  do {									\
    int e_ = (call);							\
    if (e_ != 0) {							\
      return lbmi_fail(LBMI_ERR_HIP, "%s:%d: %s: %s", __FILE__, __LINE__, \
		       #call, hipGetErrorString((hipError_t) e_));	\
    }									\
  } while (0)

#define NCCLCHECK(call)							\
  do {									\
    ncclResult_t r_ = (call);						\
    if (r_ != ncclSuccess) {						\
      return lbmi_fail(LBMI_ERR_RCCL, "%s:%d: %s: %s", __FILE__, __LINE__, \
		       #call, ncclGetErrorString(r_));			\
    }									\
  } while (0)

/*****************************************************************************
 *
 *  lbmi_options_default  (lb_data_options_default, lb_data_options.c:24-38)
 *
 *****************************************************************************/

int lbmi_options_default(lbmi_options_t * opts) {

  if (opts == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "opts is NULL");

  memset(opts, 0, sizeof(lbmi_options_t));
  opts->nvel = 19;
  opts->ndist = 1;
  opts->nlocal[X] = 64; opts->nlocal[Y] = 64; opts->nlocal[Z] = 64;
  opts->nhalo = 1;
  opts->device = -1;
  opts->mode = LBMI_MODE_EAGER;
  opts->halo_scheme = LBMI_HALO_FULL;
  opts->cartsz = 1;
  opts->cartrank = 0;

  return 0;
}

int lbmi_model(int nvel, int8_t * cv, double * wv, double * na, double * ma) {
  if (nvel != 19 && nvel != 27) {
    return lbmi_fail(LBMI_ERR_UNSUPPORTED, "nvel = %d (19 or 27)", nvel);
  }
  if (!cv || !wv || !na || !ma) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  return lbmi_k_model(nvel, cv, wv, na, ma);
}

/* Populations needed per halo side (lb_halo_size, model.c:1192-1219): the
 * low halo plane of direction d is read by pulls with c_d = +1. */

static void lbmi_halo_selections(lbmi_t * lb) {
  for (int d = 0; d < 3; d++) {
    lbmi_halo_sel_t * sf = &lb->sel_full[d];
    lbmi_halo_sel_t * sr = &lb->sel_reduced[d];
    memset(sf, 0, sizeof(*sf));
    memset(sr, 0, sizeof(*sr));
    for (int p = 0; p < lb->kp.nvel; p++) {
      sf->lo[sf->nlo++] = (int8_t) p;
      sf->hi[sf->nhi++] = (int8_t) p;
      if (lb->cv[p][d] == +1) sr->lo[sr->nlo++] = (int8_t) p;
      if (lb->cv[p][d] == -1) sr->hi[sr->nhi++] = (int8_t) p;
    }
  }
}

/*****************************************************************************
 *
 *  lbmi_create  (lb_data_create + lb_init, model.c:56-331;
 *                halo_swap_create, halo_swap.c:109-266)
 *
 *****************************************************************************/

/* The decomposition, whichever way the options give it: cartdim 0, 1, 2 =
 * cartsz slabs along that direction; LBMI_CART_GENERAL = the grid of
 * cartgrid with this rank at cartcoords (ranks numbered as MPI_Cart_create
 * numbers them without reordering: Z fastest). */

static int lbmi_cart_rank(const int grid[3], const int coords[3]) {
  return (coords[X]*grid[Y] + coords[Y])*grid[Z] + coords[Z];
}

static int lbmi_cart_check(const lbmi_options_t * opts) {
  if (opts->cartdim != LBMI_CART_GENERAL) {
    for (int d = 0; d < 3; d++) {
      if (opts->cartgrid[d] != 0 || opts->cartcoords[d] != 0) {
	return lbmi_fail(LBMI_ERR_ARGUMENT, "cartgrid / cartcoords are for "
			 "cartdim = %d only (must be zero)", LBMI_CART_GENERAL);
      }
    }
    return 0;
  }
  for (int d = 0; d < 3; d++) {
    if (opts->cartgrid[d] < 1 || opts->cartcoords[d] < 0 ||
	opts->cartcoords[d] >= opts->cartgrid[d]) {
      return lbmi_fail(LBMI_ERR_ARGUMENT, "cartgrid[%d] = %d, cartcoords[%d] = %d",
		       d, opts->cartgrid[d], d, opts->cartcoords[d]);
    }
  }
  if (opts->cartgrid[X]*opts->cartgrid[Y]*opts->cartgrid[Z] != opts->cartsz ||
      lbmi_cart_rank(opts->cartgrid, opts->cartcoords) != opts->cartrank) {
    return lbmi_fail(LBMI_ERR_ARGUMENT, "cartsz / cartrank = %d / %d do not "
		     "belong to the grid %d x %d x %d at (%d, %d, %d)",
		     opts->cartsz, opts->cartrank, opts->cartgrid[X],
		     opts->cartgrid[Y], opts->cartgrid[Z], opts->cartcoords[X],
		     opts->cartcoords[Y], opts->cartcoords[Z]);
  }
  return 0;
}

static void lbmi_cart_of(const lbmi_options_t * opts, int csz[3], int ccoord[3],
			 int * xdim, int * multi) {
  int ndec = 0;
  for (int d = 0; d < 3; d++) {
    csz[d] = 1;
    ccoord[d] = 0;
  }
  if (opts->cartdim == LBMI_CART_GENERAL) {
    *xdim = -1;
    for (int d = 0; d < 3; d++) {
      csz[d] = opts->cartgrid[d];
      ccoord[d] = opts->cartcoords[d];
      if (csz[d] > 1) {
	ndec += 1;
	if (*xdim < 0) *xdim = d;
      }
    }
    if (*xdim < 0) *xdim = X;
  }
  else {
    *xdim = opts->cartdim;
    csz[*xdim] = opts->cartsz;
    ccoord[*xdim] = opts->cartrank;
  }
  *multi = (ndec > 1);
}

/* the ranks below and above along direction d (periodic) */

static void lbmi_cart_nbr(const int csz[3], const int ccoord[3], int d,
			  int * prev, int * next) {
  int c[3] = {ccoord[X], ccoord[Y], ccoord[Z]};
  c[d] = (ccoord[d] + csz[d] - 1) % csz[d];
  *prev = lbmi_cart_rank(csz, c);
  c[d] = (ccoord[d] + 1) % csz[d];
  *next = lbmi_cart_rank(csz, c);
}

static void lbmi_cart_init(lbmi_t * lb) {
  lbmi_cart_of(&lb->opts, lb->csz, lb->ccoord, &lb->xdim, &lb->multi);
}

/* Is direction d exchanged with neighbours (else: wrapped on this rank)? A
 * one-rank ring along the slab direction counts: its planes travel. */

static int lbmi_dec(const lbmi_t * lb, int d) {
  if (lb->multi) return lb->csz[d] > 1;
  return d == lb->xdim && (lb->opts.cartsz > 1 || lb->have_comm);
}

int lbmi_create(const lbmi_options_t * opts, lbmi_t ** handle) {

  int prio_low = 0, prio_high = 0;

  lbmi_t * lb = NULL;
  int ndevice = 0;
  hipDeviceProp_t prop;

  if (opts == NULL || handle == NULL) {
    return lbmi_fail(LBMI_ERR_ARGUMENT, "lbmi_create: NULL argument");
  }
  *handle = NULL;

  if (opts->nvel != 19 && opts->nvel != 27) {
    return lbmi_fail(LBMI_ERR_UNSUPPORTED, "nvel = %d: only d3q19 and d3q27",
		     opts->nvel);
  }
  if (opts->ndist != 1 && opts->ndist != 2) {
    return lbmi_fail(LBMI_ERR_UNSUPPORTED, "ndist = %d: 1 or 2", opts->ndist);
  }
  if (opts->ndist == 2 && opts->mode == LBMI_MODE_INPLACE) {
    /* the two-distribution (symmetric_lb) step: two arrays */
    return lbmi_fail(LBMI_ERR_UNSUPPORTED, "ndist = 2 needs LBMI_MODE_EAGER, LBMI_MODE_FUSED_HALO or LBMI_MODE_FUSED");
  }
  if (opts->nhalo < 1) return lbmi_fail(LBMI_ERR_ARGUMENT, "nhalo < 1");
  for (int d = 0; d < 3; d++) {
    if (opts->nlocal[d] < 1) {
      return lbmi_fail(LBMI_ERR_ARGUMENT, "nlocal[%d] = %d", d, opts->nlocal[d]);
    }
  }
  if (opts->cartsz < 1 || opts->cartrank < 0 || opts->cartrank >= opts->cartsz) {
    return lbmi_fail(LBMI_ERR_ARGUMENT, "cartsz/cartrank = %d/%d",
		     opts->cartsz, opts->cartrank);
  }
  if (opts->cartdim < 0 || opts->cartdim > LBMI_CART_GENERAL) {
    return lbmi_fail(LBMI_ERR_ARGUMENT, "cartdim = %d (0 X, 1 Y, 2 Z, 3 general)", opts->cartdim);
  }
  {
    int ifail = lbmi_cart_check(opts);
    if (ifail) return ifail;
  }
  if (opts->mode != LBMI_MODE_EAGER && opts->mode != LBMI_MODE_FUSED &&
      opts->mode != LBMI_MODE_INPLACE && opts->mode != LBMI_MODE_FUSED_HALO) {
    return lbmi_fail(LBMI_ERR_ARGUMENT, "mode = %d", opts->mode);
  }
  if (opts->halo_scheme != LBMI_HALO_FULL &&
      opts->halo_scheme != LBMI_HALO_REDUCED) {
    return lbmi_fail(LBMI_ERR_ARGUMENT, "halo_scheme = %d", opts->halo_scheme);
  }

  {
    long long nsite = 1;
    long long nall[3];
    for (int d = 0; d < 3; d++) {
      nall[d] = (long long) opts->nlocal[d] + 2*opts->nhalo;
      nsite *= nall[d];
    }
    /* site indices are 32-bit in the kernels (the reference uses int
     * throughout, LB_ADDR); population offsets are 64-bit */
    if (nsite >= 2147483647LL) {
      return lbmi_fail(LBMI_ERR_UNSUPPORTED, "nsite = %lld exceeds 2^31-1",
		       nsite);
    }
  }

  if (hipGetDeviceCount(&ndevice) != hipSuccess || ndevice < 1) {
    return lbmi_fail(LBMI_ERR_NODEVICE,
		     "no HIP device: liblbmi has no CPU fallback");
  }

  lb = (lbmi_t *) calloc(1, sizeof(lbmi_t));
  if (lb == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "calloc failed");

  lb->opts = *opts;
  lbmi_cart_init(lb);
  if (lb->multi && (opts->mode == LBMI_MODE_FUSED || opts->mode == LBMI_MODE_INPLACE)) {
    /* more than one direction decomposed: the exchange is the sequence of
     * passes X, Y, Z where lb_halo is called; the propagation alone is folded
     * into the next collision */
    lb->opts.mode = LBMI_MODE_FUSED_HALO;
  }
  if (opts->cartsz > 1 && lb->xdim == Z && !lb->multi &&
      (opts->mode == LBMI_MODE_FUSED || opts->mode == LBMI_MODE_INPLACE)) {
    /* Slabs along Z: a boundary plane is one value out of every row of the
     * array, and a launch that redoes those two planes against the exchange
     * buffers (what Y slabs do, lbmi_fused_step) costs more in 64-byte
     * sectors touched for 8 bytes than the overlap buys: measured on a
     * 256 x 256 x 32 slab 0.273 ms per step against 0.195 ms with the exchange
     * where lb_halo is called and the propagation alone folded into the next
     * collision (profiles/r03_slab_directions.txt). So that is what FUSED
     * means on Z slabs. */
    lb->opts.mode = LBMI_MODE_FUSED_HALO;
  }
  if (opts->device >= 0) {
    if (opts->device >= ndevice) {
      free(lb);
      return lbmi_fail(LBMI_ERR_NODEVICE, "device %d of %d", opts->device,
		       ndevice);
    }
    lb->device = opts->device;
  }
  else {
    if (hipGetDevice(&lb->device) != hipSuccess) lb->device = 0;
  }
  if (hipSetDevice(lb->device) != hipSuccess) {
    free(lb);
    return lbmi_fail(LBMI_ERR_HIP, "hipSetDevice(%d) failed", lb->device);
  }
  if (hipGetDeviceProperties(&prop, lb->device) == hipSuccess) {
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
      free(lb);
      return lbmi_fail(LBMI_ERR_NODEVICE, "device arch %s: liblbmi is built "
		       "for gfx950 (MI355X) only", prop.gcnArchName);
    }
  }

  /* Kernel parameter block */
  lb->kp.nvel = opts->nvel;
  lb->kp.scheme = LBMI_RELAXATION_M10;
  lb->kp.nhalo = opts->nhalo;
  for (int d = 0; d < 3; d++) {
    lb->kp.nlocal[d] = opts->nlocal[d];
    lb->kp.nall[d] = opts->nlocal[d] + 2*opts->nhalo;
  }
  lb->kp.stry = lb->kp.nall[Z];
  lb->kp.strx = lb->kp.nall[Y]*lb->kp.nall[Z];
  lb->kp.nsite = (long long) lb->kp.nall[X]*lb->kp.strx;
  /* Reciprocals biased upwards by 2^-40 so that (int)((double) i*rstr) is
   * exactly i/str for every 0 <= i < 2^31 (the product is >= q when
   * i = q*str, and < q + 1 when i = q*str + str - 1 since q*str < 2^40) */
  lb->kp.rstrx = (1.0/lb->kp.strx)*(1.0 + 1.0/1099511627776.0);
  lb->kp.rstry = (1.0/lb->kp.stry)*(1.0 + 1.0/1099511627776.0);

  {
    double wv[LBMI_NVEL_MAX], na[LBMI_NVEL_MAX];
    double * ma = (double *) malloc(sizeof(double)*LBMI_NVEL_MAX*LBMI_NVEL_MAX);
    if (ma == NULL) { free(lb); return lbmi_fail(LBMI_ERR_ARGUMENT, "malloc"); }
    lbmi_k_model(opts->nvel, &lb->cv[0][0], wv, na, ma);
    free(ma);
  }
  lbmi_halo_selections(lb);

  /* Launch tuning of the fused kernel, measured on MI355X (profiles/,
   * DESIGN.md): XCDs interleaved in groups of 16 blocks, and 64 KiB of
   * (unused) dynamic LDS per 256-thread block so that 2 blocks = 8 waves
   * are resident per CU: with more waves in flight the streamed lines of
   * the 2 x nvel arrays overflow the 4 MiB L2 of an XCD before the
   * neighbouring wave has used its share of them. */
  lb->kp.xcd_group = 32;
  lb->kp.lds_cap = 65536;
  lb->x_packed = 1;
  lb->x_direct = 1;
  lb->halo_fold = 1;
  lb->x_concurrent = 1;
  lb->eager_oop = 1;
  lb->use_blocked = 1;               /* profiles/r01_blocked_order.txt */
  lb->kp.fe_tiled = 1;
  lb->kp.fe_stripes = 0;             /* profiles/r03_rejected.txt, 6 */
  lb->kp.fe_xcd_group = 8;
  lb->nt_store_mode = -1;            /* idem: nontemporal stores when f, fprime
					exceed the Infinity Cache */
  lb->grad_npt = 7;
  lb->adv_order = 1;

  /* Defaults of the reference: rho0 = 1, eta = zeta = 1/6 (physics.c:33-56) */
  lbmi_set_relaxation(lb, LBMI_RELAXATION_M10, 1.0, 1.0/6.0, 1.0/6.0);

  /* lbmi_free() releases whatever exists so far (members start as NULL) */
  /* the exchange and the boundary planes sit on the critical path of a slab
   * step while the long interior launch fills the chip: highest priority */
  if (hipDeviceGetStreamPriorityRange(&prio_low, &prio_high) != hipSuccess) {
    prio_low = 0;
    prio_high = 0;
  }
  if (hipStreamCreateWithFlags(&lb->own_stream, hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithPriority(&lb->comm_stream, hipStreamNonBlocking, prio_high) != hipSuccess ||
      hipStreamCreateWithPriority(&lb->bnd_stream, hipStreamNonBlocking, prio_high) != hipSuccess ||
      hipEventCreateWithFlags(&lb->ev_bnd, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&lb->ev_ready, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&lb->ev_halo, hipEventDisableTiming) != hipSuccess) {
    lbmi_free(lb);
    return lbmi_fail(LBMI_ERR_HIP, "stream/event creation failed");
  }
  lb->stream = lb->own_stream;

  if (hipMalloc((void **) &lb->mom_work,
		sizeof(double)*12*(size_t) lbmi_k_moments_nblk()) != hipSuccess ||
      hipMalloc((void **) &lb->mom_out, sizeof(double)*16) != hipSuccess) {
    lbmi_free(lb);
    return lbmi_fail(LBMI_ERR_HIP, "hipMalloc (moments workspace) failed");
  }
  if (hipHostMalloc((void **) &lb->wall_err, sizeof(int), hipHostMallocMapped) != hipSuccess) {
    lbmi_free(lb);
    return lbmi_fail(LBMI_ERR_HIP, "hipHostMalloc (link error flag) failed");
  }
  *lb->wall_err = 0;

  *handle = lb;

  return 0;
}

/*****************************************************************************
 *
 *  lbmi_free  (lb_free, model.c:181-213; halo_swap_free)
 *
 *****************************************************************************/

static void lbmi_wall_release(lbmi_t * lb);
static void lbmi_run_graph_release(lbmi_t * lb);

int lbmi_free(lbmi_t * lb) {

  if (lb == NULL) return 0;

  hipSetDevice(lb->device);
  if (lb->own_stream) hipStreamSynchronize(lb->stream);
  if (lb->comm_stream) hipStreamSynchronize(lb->comm_stream);
  if (lb->bnd_stream) hipStreamSynchronize(lb->bnd_stream);

  lbmi_comm_free(lb);

  if (lb->owns_f) {
    hipFree(lb->f);
    hipFree(lb->fprime);
  }
  if (lb->mom_work) hipFree(lb->mom_work);
  if (lb->mom_out) hipFree(lb->mom_out);
  if (lb->fe_force) hipFree(lb->fe_force);
  lbmi_wall_release(lb);
  if (lb->wall_err) hipHostFree(lb->wall_err);
  lbmi_run_graph_release(lb);
  if (lb->ev_graph) hipEventDestroy(lb->ev_graph);
  if (lb->ev_created) {
    for (int n = 0; n < LBMI_NEVENT; n++) {
      if (lb->ev0[n]) hipEventDestroy(lb->ev0[n]);
      if (lb->ev1[n]) hipEventDestroy(lb->ev1[n]);
    }
    for (int k = 0; k < 6; k++) {
      for (int n = 0; n < LBMI_NDETAIL; n++) {
	if (lb->evd[k][n]) hipEventDestroy(lb->evd[k][n]);
      }
    }
  }
  if (lb->ev_ready) hipEventDestroy(lb->ev_ready);
  if (lb->ev_halo) hipEventDestroy(lb->ev_halo);
  if (lb->comm_stream) hipStreamDestroy(lb->comm_stream);
  if (lb->ev_bnd) hipEventDestroy(lb->ev_bnd);
  if (lb->bnd_stream) hipStreamDestroy(lb->bnd_stream);
  if (lb->own_stream) hipStreamDestroy(lb->own_stream);
  free(lb);

  return 0;
}

int lbmi_nsite(const lbmi_t * lb, size_t * nsite) {
  if (lb == NULL || nsite == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  *nsite = (size_t) lb->kp.nsite;
  return 0;
}

int lbmi_nall(const lbmi_t * lb, int nall[3]) {
  if (lb == NULL || nall == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  for (int d = 0; d < 3; d++) nall[d] = lb->kp.nall[d];
  return 0;
}

/*****************************************************************************
 *
 *  lbmi_set_relaxation
 *
 *  lb_collision_relaxation_times_set (collision.c:1181-1264) and the rates
 *  the single-fluid kernel actually uses: lb_relaxation_time_shear_v
 *  (:1287-1300), _bulk_v (:1339-1373), _ghosts_v (:1443-1538).
 *
 *****************************************************************************/

int lbmi_set_relaxation(lbmi_t * lb, int scheme, double rho0,
			double eta_shear, double eta_bulk) {

  const double cs2 = (1.0/3.0);
  double rtau, rtau_bulk;

  if (lb == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "lb is NULL");
  if (!(rho0 > 0.0)) return lbmi_fail(LBMI_ERR_ARGUMENT, "rho0 = %g", rho0);

  rtau = 1.0/(0.5 + eta_shear/(rho0*cs2));
  rtau_bulk = 1.0/(0.5 + eta_bulk/(rho0*cs2));
  lb->rho0 = rho0;
  lb->kp.rho0 = rho0;
  lb->kp.bulk_ratio = eta_bulk/eta_shear;

  switch (scheme) {
  case LBMI_RELAXATION_M10:
    lb->kp.rtau_shear = rtau;
    lb->kp.rtau_bulk = rtau_bulk;
    lb->kp.rtau_even = 1.0;
    lb->kp.rtau_odd = 1.0;
    break;
  case LBMI_RELAXATION_BGK:
    lb->kp.rtau_shear = rtau;
    lb->kp.rtau_bulk = rtau;       /* no separate bulk viscosity */
    lb->kp.rtau_even = rtau;
    lb->kp.rtau_odd = rtau;
    break;
  case LBMI_RELAXATION_TRT:
    if (lb->kp.nvel != 19) {
      /* the reference leaves rtau_ghost[] uninitialised for d3q27
       * (collision.c:1487-1534): reject rather than imitate */
      return lbmi_fail(LBMI_ERR_UNSUPPORTED, "TRT is defined for d3q19 only");
    }
    {
      double tau = eta_shear/(rho0*cs2);
      double rodd = 0.5 + 2.0*tau/(tau + 3.0/8.0);
      if (rodd > 2.0) rodd = 2.0;
      lb->kp.rtau_shear = rtau;
      lb->kp.rtau_bulk = rtau_bulk;
      lb->kp.rtau_even = rtau;     /* modes 10, 14, 18 */
      lb->kp.rtau_odd = rodd;      /* modes 11-13, 15-17 */
    }
    break;
  default:
    return lbmi_fail(LBMI_ERR_ARGUMENT, "relaxation scheme %d", scheme);
  }

  lb->kp.scheme = scheme;

  return 0;
}

int lbmi_set_body_force(lbmi_t * lb, const double fbody[3]) {
  if (lb == NULL || fbody == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  for (int d = 0; d < 3; d++) lb->kp.fbody[d] = fbody[d];
  return 0;
}

int lbmi_relaxation_rates(const lbmi_t * lb, double rtau[4]) {
  if (lb == NULL || rtau == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  rtau[0] = lb->kp.rtau_shear;
  rtau[1] = lb->kp.rtau_bulk;
  rtau[2] = lb->kp.rtau_even;
  rtau[3] = lb->kp.rtau_odd;
  return 0;
}

/*****************************************************************************
 *
 *  Stateless kernels
 *
 *****************************************************************************/

static lbmi_hydro_dev_t lbmi_hydro_dev(const lbmi_hydro_t * hydro) {
  lbmi_hydro_dev_t h;
  memset(&h, 0, sizeof(h));          /* padding too: the struct is compared */
  if (hydro) {
    h.stride = hydro->nsite;
    h.force = hydro->force;
    h.status = hydro->status;
    h.rho = hydro->rho;
    h.u = hydro->u;
    h.eta = hydro->eta;
  }
  return h;
}

/* Arrays known to hold zeros everywhere: set by lbmi_hydro_field_set with
 * zero values, dropped when something of this library writes to the array
 * or the caller says so (lbmi_hydro_field_dirty). A force array on the list
 * is not read by the collision, and zeroing it again is not a launch. */

static void lbmi_run_graph_release(lbmi_t * lb);

static int lbmi_known_zero(const lbmi_t * lb, const void * p) {
  if (p == NULL) return 0;
  for (int n = 0; n < 4; n++) {
    if (lb->known_zero[n] == p) return 1;
  }
  return 0;
}

static void lbmi_known_zero_drop(lbmi_t * lb, const void * p) {
  for (int n = 0; n < 4; n++) {
    if (p != NULL && lb->known_zero[n] == p) {
      lb->known_zero[n] = NULL;
      /* a captured lbmi_lb_run (tune "graph") skipped this array */
      lbmi_run_graph_release(lb);
    }
  }
}

static void lbmi_known_zero_add(lbmi_t * lb, const void * p) {
  if (lbmi_known_zero(lb, p)) return;
  for (int n = 0; n < 4; n++) {
    if (lb->known_zero[n] == NULL) {
      lb->known_zero[n] = p;
      lbmi_run_graph_release(lb);    /* captured launches still read it */
      return;
    }
  }
  lb->known_zero[0] = p;             /* full: forget the oldest entry */
  lbmi_run_graph_release(lb);
}

/* hydro_lazy: the rho and u a collision would have stored are formed from the
 * post-collision state it left, when somebody wants them (lbmi_lb_hydro_sync,
 * or anything that is about to change that state: a flush, a propagation
 * that runs at once, a copy into f). The next collision supersedes them. */

static int lbmi_hydro_materialise(lbmi_t * lb) {
  if (!lb->hydro_stale) return 0;
  lb->hydro_stale = 0;
  if (lb->f == NULL) return 0;
  {
    lbmi_kparam_t kp = lb->kp;
    for (int ia = 0; ia < 3; ia++) kp.fbody[ia] = lb->lazy_fbody[ia];
    KCHECK(lbmi_k_hydro_from_f(&kp, lb->f, &lb->lazy_h, lb->blocked, lb->stream));
  }
  lbmi_known_zero_drop(lb, lb->lazy_h.rho);
  lbmi_known_zero_drop(lb, lb->lazy_h.u);
  return 0;
}

/* A call of this library is about to read `reads` (u, rho) or overwrite
 * `writes` (the force): what a lazy collision still owes must exist first --
 * u for the reader, and for the writer because the owed u is formed with the
 * force the collision used. A force array that gets written is no longer
 * known to be zero. */

static int lbmi_hydro_touch(lbmi_t * lb, const double * reads,
			    const double * writes) {
  if (lb->hydro_stale &&
      ((reads != NULL && (reads == lb->lazy_h.u || reads == lb->lazy_h.rho)) ||
       (writes != NULL && (writes == lb->lazy_h.force ||
			   writes == lb->lazy_h.u || writes == lb->lazy_h.rho)))) {
    int ifail = lbmi_hydro_materialise(lb);
    if (ifail) return ifail;
  }
  lbmi_known_zero_drop(lb, writes);
  return 0;
}

int lbmi_lb_hydro_sync(lbmi_t * lb) {
  if (lb == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  HIPCHECK(hipSetDevice(lb->device));
  return lbmi_hydro_materialise(lb);
}

int lbmi_hydro_field_dirty(lbmi_t * lb, const double * field) {
  if (lb == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  lbmi_known_zero_drop(lb, field);
  return 0;
}

/* Event-pair timing of a kernel launch on the compute stream */

static int lbmi_time_begin(lbmi_t * lb) {
  lb->timing_now = 0;
  if (!lb->timing) return 0;
  /* every timing-th launch: an event record costs the stream a few
   * microseconds between kernels */
  if ((lb->timing_count++ % lb->timing) != 0) return 0;
  lb->timing_now = 1;
  if (lb->nev == LBMI_NEVENT) {
    double ms; int n;
    int ifail = lbmi_timing_read(lb, &ms, &n);   /* drains into accumulators */
    if (ifail) return ifail;
    lb->ms_accum = ms; lb->launches_accum = n;
  }
  HIPCHECK(hipEventRecord(lb->ev0[lb->nev], lb->stream));
  return 0;
}

static int lbmi_time_end(lbmi_t * lb) {
  if (!lb->timing_now) return 0;
  lb->timing_now = 0;
  HIPCHECK(hipEventRecord(lb->ev1[lb->nev], lb->stream));
  lb->nev += 1;
  return 0;
}

int lbmi_timing(lbmi_t * lb, int on) {
  if (lb == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  if (on > 0 && !lb->ev_created) {
    /* here, not at the first timed launch: creating the events costs a
     * fraction of a millisecond that must not land in a timed region */
    HIPCHECK(hipSetDevice(lb->device));
    lb->ev_created = 1;               /* lbmi_free destroys the non-NULL ones */
    for (int n = 0; n < LBMI_NEVENT; n++) {
      HIPCHECK(hipEventCreate(&lb->ev0[n]));
      HIPCHECK(hipEventCreate(&lb->ev1[n]));
    }
    for (int k = 0; k < 6; k++) {
      for (int n = 0; n < LBMI_NDETAIL; n++) HIPCHECK(hipEventCreate(&lb->evd[k][n]));
    }
  }
  lb->timing = (on > 0) ? on : 0;
  lb->timing_count = 0;
  lb->timing_now = 0;
  lb->nev = 0;
  lb->nd = 0;
  lb->ms_accum = 0.0;
  lb->launches_accum = 0;
  return 0;
}

int lbmi_timing_read(lbmi_t * lb, double * ms_total, int * nlaunch) {
  if (lb == NULL || !ms_total || !nlaunch) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  HIPCHECK(hipSetDevice(lb->device));
  HIPCHECK(hipStreamSynchronize(lb->stream));
  {
    double ms = lb->ms_accum;
    int n = lb->launches_accum;
    for (int k = 0; k < lb->nev; k++) {
      float t = 0.0f;
      HIPCHECK(hipEventElapsedTime(&t, lb->ev0[k], lb->ev1[k]));
      ms += t;
    }
    n += lb->nev;
    lb->nev = 0;
    lb->ms_accum = 0.0;
    lb->launches_accum = 0;
    *ms_total = ms;
    *nlaunch = n;
  }
  return 0;
}

/* The sampled slab steps (cartsz > 1 or a ring; the first LBMI_NDETAIL of
 * them) in detail: average milliseconds of the interior launch, of the
 * exchange (pack if any, messages, unpack if any) and of the boundary
 * launch, each on its own stream. Resets the samples. */

int lbmi_timing_read_detail(lbmi_t * lb, double ms[3], int * nsample) {
  if (lb == NULL || ms == NULL || nsample == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  HIPCHECK(hipSetDevice(lb->device));
  HIPCHECK(hipStreamSynchronize(lb->comm_stream));
  HIPCHECK(hipStreamSynchronize(lb->bnd_stream));
  HIPCHECK(hipStreamSynchronize(lb->stream));
  ms[0] = ms[1] = ms[2] = 0.0;
  *nsample = lb->nd;
  for (int n = 0; n < lb->nd; n++) {
    for (int k = 0; k < 3; k++) {
      float t = 0.0f;
      /* an empty interior launch (one or two planes per slab) records both
       * events back to back */
      HIPCHECK(hipEventElapsedTime(&t, lb->evd[2*k][n], lb->evd[2*k + 1][n]));
      ms[k] += t;
    }
  }
  if (lb->nd > 0) {
    for (int k = 0; k < 3; k++) ms[k] /= lb->nd;
  }
  lb->nd = 0;
  return 0;
}

int lbmi_tune(lbmi_t * lb, const char * key, int value) {
  if (lb == NULL || key == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  if (strcmp(key, "xcd_group") == 0) {
    if (value < 0 || value > 65536) return lbmi_fail(LBMI_ERR_ARGUMENT, "xcd_group");
    lb->kp.xcd_group = value;
    return 0;
  }
  if (strcmp(key, "fe_xcd_group") == 0) {
    if (value < 0 || value > 65536) return lbmi_fail(LBMI_ERR_ARGUMENT, "fe_xcd_group");
    lb->kp.fe_xcd_group = value;
    return 0;
  }
  if (strcmp(key, "x_packed") == 0) {
    lb->x_packed = (value != 0);
    return 0;
  }
  if (strcmp(key, "nt_store") == 0) {
    lb->nt_store_mode = (value < 0) ? -1 : (value & 3);   /* bit 0: f, bit 1: rho, u */
    return 0;
  }
    if (strcmp(key, "hydro_lazy") == 0) {
    if (!value) {
      int ifail;
      HIPCHECK(hipSetDevice(lb->device));
      ifail = lbmi_hydro_materialise(lb);
      if (ifail) return ifail;
    }
    if (value < 0 || value > 2) return lbmi_fail(LBMI_ERR_ARGUMENT, "hydro_lazy = %d (0, 1, 2)", value);
    if (lb->hydro_lazy != value) lbmi_run_graph_release(lb);
    lb->hydro_lazy = value;          /* 1: rho and u on demand, 2: rho only */
    return 0;
  }
  if (strcmp(key, "fe_stripes") == 0) {
    lb->kp.fe_stripes = (value != 0);
    return 0;
  }
  if (strcmp(key, "fe_tiled") == 0) {
    lb->kp.fe_tiled = (value != 0);
    return 0;
  }
  if (strcmp(key, "x_direct") == 0) {
    lb->x_direct = (value != 0);
    lb->xsend_valid = 0; lb->halo_fresh = 0;
    return 0;
  }
  if (strcmp(key, "halo_fold") == 0) {
    lb->halo_fold = (value != 0);
    return 0;
  }
  if (strcmp(key, "eager_oop") == 0) {
    lb->eager_oop = (value != 0);
    return 0;
  }
  if (strcmp(key, "x_concurrent") == 0) {
    lb->x_concurrent = (value != 0);
    return 0;
  }
  if (strcmp(key, "graph") == 0) {
    /* lbmi_lb_run on one GPU in FUSED mode: pairs of steps as one hipGraph
     * launch (fewer, cheaper launches for lattices whose step is short) */
    lb->use_graph = (value != 0);
    return 0;
  }
  if (strcmp(key, "blocked") == 0) {
    /* takes effect at the next fused step; a state already blocked is
     * converted back there or at the next flush */
    lb->use_blocked = (value != 0);
    return 0;
  }
  if (strcmp(key, "lds_cap") == 0) {
    if (value < 0 || value > 163840) {
      return lbmi_fail(LBMI_ERR_ARGUMENT, "lds_cap = %d (0..163840)", value);
    }
    lb->kp.lds_cap = value;
    return 0;
  }
  return lbmi_fail(LBMI_ERR_ARGUMENT, "unknown tuning key %s", key);
}

int lbmi_collide(lbmi_t * lb, double * f, const lbmi_hydro_t * hydro) {
  lbmi_hydro_dev_t h = lbmi_hydro_dev(hydro);
  if (lb == NULL || f == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  HIPCHECK(hipSetDevice(lb->device));
  KCHECK(lbmi_k_collide(&lb->kp, f, &h, lb->stream));
  return 0;
}

int lbmi_propagate(lbmi_t * lb, const double * f, double * fprime) {
  if (lb == NULL || f == NULL || fprime == NULL || f == fprime) {
    return lbmi_fail(LBMI_ERR_ARGUMENT, "lbmi_propagate: bad pointers");
  }
  HIPCHECK(hipSetDevice(lb->device));
  KCHECK(lbmi_k_propagate(&lb->kp, f, fprime, lb->stream));
  return 0;
}

/* Which directions can be wrapped by index arithmetic on this rank */

static int lbmi_wrapmask(const lbmi_t * lb) {
  int mask = 1 | 2 | 4;
  /* a decomposed direction has its halo planes filled by the neighbours */
  for (int d = 0; d < 3; d++) {
    if (lbmi_dec(lb, d)) mask &= ~(1 << d);
  }
  return mask;
}

int lbmi_propagate_collide(lbmi_t * lb, const double * f, double * fprime,
			   const lbmi_hydro_t * hydro, int wrap) {
  lbmi_hydro_dev_t h = lbmi_hydro_dev(hydro);
  int ifail;
  if (lb == NULL || f == NULL || fprime == NULL || f == fprime) {
    return lbmi_fail(LBMI_ERR_ARGUMENT, "lbmi_propagate_collide: bad pointers");
  }
  HIPCHECK(hipSetDevice(lb->device));
  ifail = lbmi_time_begin(lb);
  if (ifail) return ifail;
  KCHECK(lbmi_k_propagate_collide(&lb->kp, f, fprime, &h,
				  wrap ? lbmi_wrapmask(lb) : 0, 0,
				  lb->kp.nhalo,
				  lb->kp.nhalo + lb->kp.nlocal[X] - 1, 0, -1, NULL,
				  lb->stream));
  return lbmi_time_end(lb);
}

/*****************************************************************************
 *
 *  Halo swap
 *
 *  halo_swap_packed (halo_swap.c:709-1063): passes X, Y, Z in order; each
 *  covers the full extent of the other two directions so that edges and
 *  corners are completed by the later passes. On one rank a pass is a
 *  device-side periodic copy; with cartsz > 1 the X pass goes through
 *  packed device buffers and ncclSend/ncclRecv (RCCL over xGMI) instead of
 *  the reference's pinned-host staging + MPI (halo_swap.c:762-881).
 *
 *****************************************************************************/

/* The point-to-point operations of ONE X exchange of this rank, in the
 * order they are issued. Pure host arithmetic: the product path
 * (lbmi_x_sendrecv below) and lbmi_x_schedule (what tests drive with another
 * transport) both execute exactly this list.
 *
 * Towards one peer the order is the same on both sides -- what leaves
 * upwards (the "lo" components: they fill a LOW halo) before what leaves
 * downwards -- and that is what matches sends to receives, also when
 * prev == next (2 ranks: my 1st send, sendhi, meets the peer's 1st receive,
 * recvlo) or prev == next == self (1 rank).
 *
 * packed: one message per direction through staging buffers ([k][plane
 * site]): sendhi = last interior plane, components lo -> next's recvlo;
 * sendlo = first interior plane, components hi -> prev's recvhi.
 * Zero-copy: X is the slowest index, so the boundary plane of ONE component
 * is a contiguous run of strx doubles of the array itself. */

static int lbmi_x_ops(int prev, int next, const lbmi_halo_sel_t * sel,
		      long long psz, long long ns, int nh, int nlocalx,
		      int packed, int layer, lbmi_xop_t * ops) {
  int n = 0;

  if (packed) {
    const long long nlo = psz*sel->nlo;      /* arrives in / leaves for low halos */
    const long long nhi = psz*sel->nhi;
    const lbmi_xop_t four[4] = {
      {LBMI_XOP_SEND, next, LBMI_XBUF_SENDHI, 0, nlo},
      {LBMI_XOP_RECV, prev, LBMI_XBUF_RECVLO, 0, nlo},
      {LBMI_XOP_SEND, prev, LBMI_XBUF_SENDLO, 0, nhi},
      {LBMI_XOP_RECV, next, LBMI_XBUF_RECVHI, 0, nhi}};
    for (int k = 0; k < 4; k++) {
      if (four[k].count > 0) ops[n++] = four[k];
    }
    return n;
  }

  {
    const long long last = (long long) (nh + nlocalx - 1 - layer)*psz;
    const long long first = (long long) (nh + layer)*psz;
    const long long halo_lo = (long long) (nh - 1 - layer)*psz;
    const long long halo_hi = (long long) (nh + nlocalx + layer)*psz;
    for (int k = 0; k < sel->nlo; k++) {
      const long long c = ns*sel->lo[k];
      const lbmi_xop_t snd = {LBMI_XOP_SEND, next, LBMI_XBUF_DATA, c + last, psz};
      const lbmi_xop_t rcv = {LBMI_XOP_RECV, prev, LBMI_XBUF_DATA, c + halo_lo, psz};
      ops[n++] = snd;
      ops[n++] = rcv;
    }
    for (int k = 0; k < sel->nhi; k++) {
      const long long c = ns*sel->hi[k];
      const lbmi_xop_t snd = {LBMI_XOP_SEND, prev, LBMI_XBUF_DATA, c + first, psz};
      const lbmi_xop_t rcv = {LBMI_XOP_RECV, next, LBMI_XBUF_DATA, c + halo_hi, psz};
      ops[n++] = snd;
      ops[n++] = rcv;
    }
  }
  return n;
}

static void lbmi_sel_make(const int8_t cv[][3], int nvel, int reduced,
			  int dim, lbmi_halo_sel_t * sx) {
  memset(sx, 0, sizeof(*sx));
  for (int p = 0; p < nvel; p++) {
    if (!reduced || cv[p][dim] == +1) sx->lo[sx->nlo++] = (int8_t) p;
    if (!reduced || cv[p][dim] == -1) sx->hi[sx->nhi++] = (int8_t) p;
  }
}

int lbmi_x_schedule(const lbmi_options_t * opts, int scheme, int packed,
		    lbmi_xop_t * ops, int maxops, int * nops) {
  int csz[3], ccoord[3], xdim = X, multi = 0;
  if (opts == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  if (opts->cartdim < 0 || opts->cartdim > LBMI_CART_GENERAL || lbmi_cart_check(opts) != 0) {
    return lbmi_fail(LBMI_ERR_ARGUMENT, "lbmi_x_schedule: bad options");
  }
  lbmi_cart_of(opts, csz, ccoord, &xdim, &multi);
  return lbmi_x_schedule_dim(opts, xdim, scheme, packed, ops, maxops, nops);
}

int lbmi_x_schedule_dim(const lbmi_options_t * opts, int dim, int scheme,
			int packed, lbmi_xop_t * ops, int maxops, int * nops) {
  int8_t cv[LBMI_NVEL_MAX][3];
  int csz[3], ccoord[3], xdim = X, multi = 0, prev = 0, next = 0;
  double wv[LBMI_NVEL_MAX], na[LBMI_NVEL_MAX];
  double * ma = NULL;
  lbmi_halo_sel_t sx;
  lbmi_xop_t tmp[LBMI_XOPS_MAX];
  long long psz, ns;
  int n;

  if (opts == NULL || ops == NULL || nops == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  if (opts->nvel != 19 && opts->nvel != 27) return lbmi_fail(LBMI_ERR_UNSUPPORTED, "nvel = %d", opts->nvel);
  if (scheme != LBMI_HALO_FULL && scheme != LBMI_HALO_REDUCED) {
    return lbmi_fail(LBMI_ERR_ARGUMENT, "halo scheme %d", scheme);
  }
  if (opts->cartsz < 1 || opts->cartrank < 0 || opts->cartrank >= opts->cartsz ||
      opts->nhalo < 1 || opts->nlocal[X] < 1 || opts->nlocal[Y] < 1 || opts->nlocal[Z] < 1 ||
      opts->cartdim < 0 || opts->cartdim > LBMI_CART_GENERAL || dim < X || dim > Z ||
      lbmi_cart_check(opts) != 0) {
    return lbmi_fail(LBMI_ERR_ARGUMENT, "lbmi_x_schedule: bad options");
  }
  lbmi_cart_of(opts, csz, ccoord, &xdim, &multi);
  if (multi ? (csz[dim] < 2) : (dim != xdim)) {
    return lbmi_fail(LBMI_ERR_ARGUMENT, "lbmi_x_schedule: direction %d is not "
		     "decomposed", dim);
  }
  if (!packed && dim != X) {
    return lbmi_fail(LBMI_ERR_ARGUMENT, "lbmi_x_schedule: the planes of Y and Z "
		     "slabs are not contiguous: packed messages only");
  }
  lbmi_cart_nbr(csz, ccoord, dim, &prev, &next);
  ma = (double *) malloc(sizeof(double)*LBMI_NVEL_MAX*LBMI_NVEL_MAX);
  if (ma == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "malloc");
  lbmi_k_model(opts->nvel, &cv[0][0], wv, na, ma);
  free(ma);
  lbmi_sel_make(cv, opts->nvel, scheme == LBMI_HALO_REDUCED, dim, &sx);
  ns = 1;
  for (int d = 0; d < 3; d++) ns *= (long long) (opts->nlocal[d] + 2*opts->nhalo);
  psz = ns/(opts->nlocal[dim] + 2*opts->nhalo);              /* sites of a plane */
  n = lbmi_x_ops(prev, next, &sx, psz, ns, opts->nhalo,
		 opts->nlocal[dim], packed, 0, tmp);
  if (n > maxops) return lbmi_fail(LBMI_ERR_ARGUMENT, "%d operations, room for %d", n, maxops);
  memcpy(ops, tmp, sizeof(lbmi_xop_t)*(size_t) n);
  *nops = n;
  return 0;
}

/*****************************************************************************
 *
 *  The peer ring: the second transport of the X exchange. N handles of ONE
 *  process, one host thread per handle, each on a device of its own (one
 *  process driving the GPUs of a node over xGMI peer copies,
 *  hipMemcpyPeerAsync) -- or several of them, or all, on the same device (a
 *  rehearsal of the slab path on one GPU, where RCCL refuses two ranks). It
 *  replaces halo_swap.c:762-881 like the RCCL path does and needs no RCCL
 *  bootstrap (no unique id, no out-of-band exchange): bench.py --transport
 *  peer, a comparator for what the RCCL kernels cost beside the interior
 *  launch.
 *
 *  A send posts (pointer, device, count, "data ready" event) into the FIFO of
 *  its directed pair; the matching receive -- the next one issued towards
 *  that peer, RCCL's rule -- waits for the post, makes its stream wait for
 *  the event and copies device to device (the receiver pulls); the end of the
 *  group makes the sender's stream wait until its buffers have been read.
 *  Events are recorded on streams of the device they were created on: `ready`
 *  belongs to the sender's device, `done` to the receiver's. Everything else
 *  of a slab step (kernels, schedule, overlap, streams) is the code that runs
 *  over RCCL.
 *
 *****************************************************************************/

enum {RING_SLOTS = 128, LBMI_RING_MAX = 64};
enum {RING_FREE = 0, RING_POSTED = 1, RING_CONSUMED = 2};

typedef struct ring_msg_s {
  const double * ptr;
  int device;                        /* where ptr lives (the sender's) */
  size_t count;
  hipEvent_t ready;                  /* recorded by the sender: data complete */
  hipEvent_t done;                   /* recorded by the receiver: data read */
  int state;
} ring_msg_t;

typedef struct ring_fifo_s {
  ring_msg_t slot[RING_SLOTS];
  unsigned posted;                   /* messages posted so far (sender) */
  unsigned taken;                    /* messages consumed so far (receiver) */
  unsigned released;                 /* messages released so far (sender) */
} ring_fifo_t;

struct lbmi_ring_s {
  int nranks;
  int nattached;
  int failed;
  int device[LBMI_RING_MAX];         /* of each attached rank, -1: not yet */
  pthread_mutex_t mu;
  pthread_cond_t cv;
  ring_fifo_t * fifo;                /* [src*nranks + dst] */
};

int lbmi_ring_create(int nranks, lbmi_ring_t ** ring) {
  lbmi_ring_t * r = NULL;
  if (ring == NULL || nranks < 1 || nranks > LBMI_RING_MAX) {
    return lbmi_fail(LBMI_ERR_ARGUMENT, "lbmi_ring_create: 1 .. %d ranks", LBMI_RING_MAX);
  }
  *ring = NULL;
  r = (lbmi_ring_t *) calloc(1, sizeof(lbmi_ring_t));
  if (r == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "calloc");
  r->nranks = nranks;
  r->fifo = (ring_fifo_t *) calloc((size_t) nranks*nranks, sizeof(ring_fifo_t));
  if (r->fifo == NULL) { free(r); return lbmi_fail(LBMI_ERR_ARGUMENT, "calloc"); }
  for (int n = 0; n < LBMI_RING_MAX; n++) r->device[n] = -1;
  pthread_mutex_init(&r->mu, NULL);
  pthread_cond_init(&r->cv, NULL);
  *ring = r;
  return 0;
}

int lbmi_ring_free(lbmi_ring_t * r) {
  if (r == NULL) return 0;
  if (r->nattached > 0) return lbmi_fail(LBMI_ERR_STATE, "%d handles still use the ring", r->nattached);
  for (int n = 0; n < r->nranks*r->nranks; n++) {
    for (int k = 0; k < RING_SLOTS; k++) {
      if (r->fifo[n].slot[k].ready) hipEventDestroy(r->fifo[n].slot[k].ready);
      if (r->fifo[n].slot[k].done) hipEventDestroy(r->fifo[n].slot[k].done);
    }
  }
  pthread_cond_destroy(&r->cv);
  pthread_mutex_destroy(&r->mu);
  free(r->fifo);
  free(r);
  return 0;
}

/* A rank that gives up outside an exchange (its driver failed) releases the
 * others: their exchanges return LBMI_ERR_STATE from then on */

int lbmi_ring_abort(lbmi_ring_t * r) {
  if (r == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  pthread_mutex_lock(&r->mu);
  r->failed = 1;
  pthread_cond_broadcast(&r->cv);
  pthread_mutex_unlock(&r->mu);
  return 0;
}

/* A rank that fails inside an exchange must not leave the others waiting */

static int ring_fail(lbmi_ring_t * r, int code, const char * what) {
  r->failed = 1;
  pthread_cond_broadcast(&r->cv);
  pthread_mutex_unlock(&r->mu);
  return lbmi_fail(code, "in-process ring: %s", what);
}

static int ring_sendrecv(lbmi_ring_t * r, int me, int mydev,
			 unsigned long long * peers_enabled,
			 const lbmi_xop_t * ops, int nops,
			 double * const base[5], hipStream_t st) {
  unsigned first_mine[LBMI_XOPS_MAX];   /* per op: index of my send in its fifo */

  pthread_mutex_lock(&r->mu);
  /* sends never wait for the peer */
  for (int n = 0; n < nops; n++) {
    ring_fifo_t * q;
    ring_msg_t * m;
    if (ops[n].kind != LBMI_XOP_SEND) continue;
    q = &r->fifo[me*r->nranks + ops[n].peer];
    if (q->posted - q->released >= RING_SLOTS) return ring_fail(r, LBMI_ERR_STATE, "too many sends in flight");
    m = &q->slot[q->posted % RING_SLOTS];
    /* (this thread's current device is the sender's: the event is its) */
    if (m->ready == NULL &&
	hipEventCreateWithFlags(&m->ready, hipEventDisableTiming) != hipSuccess) {
      return ring_fail(r, LBMI_ERR_HIP, "hipEventCreate");
    }
    m->ptr = base[ops[n].buffer] + ops[n].offset;
    m->device = mydev;
    m->count = (size_t) ops[n].count;
    if (hipEventRecord(m->ready, st) != hipSuccess) return ring_fail(r, LBMI_ERR_HIP, "hipEventRecord");
    m->state = RING_POSTED;
    first_mine[n] = q->posted;
    q->posted += 1;
  }
  pthread_cond_broadcast(&r->cv);

  /* receives: the k-th from a peer takes the k-th message that peer posted */
  for (int n = 0; n < nops; n++) {
    ring_fifo_t * q;
    ring_msg_t * m;
    if (ops[n].kind != LBMI_XOP_RECV) continue;
    q = &r->fifo[ops[n].peer*r->nranks + me];
    while (q->posted == q->taken && !r->failed) pthread_cond_wait(&r->cv, &r->mu);
    if (r->failed) { pthread_mutex_unlock(&r->mu); return lbmi_fail(LBMI_ERR_STATE, "in-process ring: another rank failed"); }
    m = &q->slot[q->taken % RING_SLOTS];
    if (m->count != (size_t) ops[n].count) return ring_fail(r, LBMI_ERR_STATE, "a receive met a send of another length");
    /* the receiver pulls, on its own stream: from its own device, or over
     * xGMI from the sender's (peer access enabled at lbmi_comm_init_ring) */
    if (m->done == NULL &&
	hipEventCreateWithFlags(&m->done, hipEventDisableTiming) != hipSuccess) {
      return ring_fail(r, LBMI_ERR_HIP, "hipEventCreate");
    }
    if (m->device != mydev && m->device >= 0 && m->device < 64 &&
	!(*peers_enabled & (1ULL << m->device))) {
      /* a neighbour that attached after this rank: its device becomes
       * reachable now (lbmi_comm_init_ring could only enable those that were
       * already there) */
      hipError_t e = hipDeviceEnablePeerAccess(m->device, 0);
      if (e != hipSuccess) (void) hipGetLastError();   /* already enabled, or
							  not possible: the copy
							  below is staged then */
      *peers_enabled |= (1ULL << m->device);
    }
    if (hipStreamWaitEvent(st, m->ready, 0) != hipSuccess ||
	((m->device == mydev)
	 ? hipMemcpyAsync(base[ops[n].buffer] + ops[n].offset, m->ptr,
			  sizeof(double)*m->count, hipMemcpyDeviceToDevice, st)
	 : hipMemcpyPeerAsync(base[ops[n].buffer] + ops[n].offset, mydev, m->ptr,
			      m->device, sizeof(double)*m->count, st)) != hipSuccess ||
	hipEventRecord(m->done, st) != hipSuccess) {
      return ring_fail(r, LBMI_ERR_HIP, "device-to-device copy");
    }
    m->state = RING_CONSUMED;
    q->taken += 1;
    pthread_cond_broadcast(&r->cv);
  }

  /* the send buffers may be reused in stream order once they have been read */
  for (int n = 0; n < nops; n++) {
    ring_fifo_t * q;
    ring_msg_t * m;
    if (ops[n].kind != LBMI_XOP_SEND) continue;
    q = &r->fifo[me*r->nranks + ops[n].peer];
    while (q->taken <= first_mine[n] && !r->failed) pthread_cond_wait(&r->cv, &r->mu);
    if (r->failed) { pthread_mutex_unlock(&r->mu); return lbmi_fail(LBMI_ERR_STATE, "in-process ring: another rank failed"); }
    m = &q->slot[first_mine[n] % RING_SLOTS];
    if (hipStreamWaitEvent(st, m->done, 0) != hipSuccess) return ring_fail(r, LBMI_ERR_HIP, "hipStreamWaitEvent");
    m->state = RING_FREE;
    q->released += 1;
  }
  pthread_mutex_unlock(&r->mu);
  return 0;
}

/* Execute a schedule: RCCL (ncclSend/ncclRecv in one group, on st) or the
 * in-process ring. base: sendlo, sendhi, recvlo, recvhi, the array. */

static int lbmi_x_sendrecv(lbmi_t * lb, const lbmi_xop_t * ops, int nops,
			   double * const base[5], hipStream_t st) {
  if (lb->ring) {
    return ring_sendrecv(lb->ring, lb->opts.cartrank, lb->device,
			 &lb->peers_enabled, ops, nops, base, st);
  }
  NCCLCHECK(ncclGroupStart());
  for (int n = 0; n < nops; n++) {
    double * p = base[ops[n].buffer] + ops[n].offset;
    if (ops[n].kind == LBMI_XOP_SEND) {
      NCCLCHECK(ncclSend(p, (size_t) ops[n].count, ncclDouble, ops[n].peer, lb->comm, st));
    }
    else {
      NCCLCHECK(ncclRecv(p, (size_t) ops[n].count, ncclDouble, ops[n].peer, lb->comm, st));
    }
  }
  NCCLCHECK(ncclGroupEnd());
  return 0;
}

/* One X exchange of `data` (nswap layer `layer`): the reference's pinned-host
 * staging + MPI (halo_swap.c:762-881) as device-to-device messages.
 * buf: the four staging buffers to use. unpack = 0 leaves what arrived in
 * the receive buffers; packed_already: the send buffers are up to date. */

static int lbmi_x_exchange_dim(lbmi_t * lb, int dim, const lbmi_halo_sel_t * sel,
			       double * data, int blocked, int layer,
			       double * const buf[4], int packed_already,
			       int unpack, hipStream_t st);

static int lbmi_x_exchange_buf(lbmi_t * lb, const lbmi_halo_sel_t * sel,
			       double * data, int blocked, int layer,
			       double * const buf[4], int packed_already,
			       int unpack, hipStream_t st) {
  return lbmi_x_exchange_dim(lb, lb->xdim, sel, data, blocked, layer, buf,
			     packed_already, unpack, st);
}

static int lbmi_x_exchange_dim(lbmi_t * lb, int dim, const lbmi_halo_sel_t * sel,
			       double * data, int blocked, int layer,
			       double * const buf[4], int packed_already,
			       int unpack, hipStream_t st) {
  lbmi_xop_t ops[LBMI_XOPS_MAX];
  double * base[5] = {buf[0], buf[1], buf[2], buf[3], data};
  int prev = 0, next = 0;
  const long long psz = lb->kp.nsite/lb->kp.nall[dim];     /* sites of a plane */
  /* (only the planes of X slabs are contiguous runs of the array) */
  const int packed = (lb->x_packed || blocked || !unpack || packed_already ||
		      dim != X);
  int nops, ifail;

  if (!lb->have_comm) {
    return lbmi_fail(LBMI_ERR_STATE, "cartsz = %d but lbmi_comm_init() has "
		     "not been called", lb->opts.cartsz);
  }
  if (packed && ((size_t) psz*(size_t) sel->nlo > lb->xbuf_doubles ||
		 (size_t) psz*(size_t) sel->nhi > lb->xbuf_doubles)) {
    return lbmi_fail(LBMI_ERR_STATE, "halo buffers too small");
  }
  lbmi_cart_nbr(lb->csz, lb->ccoord, dim, &prev, &next);
  nops = lbmi_x_ops(prev, next, sel, psz,
		    lb->kp.nsite, lb->kp.nhalo, lb->kp.nlocal[dim], packed, layer,
		    ops);
  if (packed && !packed_already) {
    KCHECK(lbmi_k_halo_pack(&lb->kp, dim, sel, data, buf[0], buf[1], blocked,
			    layer, st));
  }
  ifail = lbmi_x_sendrecv(lb, ops, nops, base, st);
  if (ifail) return ifail;
  if (packed && unpack) {
    KCHECK(lbmi_k_halo_unpack(&lb->kp, dim, sel, data, buf[2], buf[3], blocked,
			      layer, st));
  }
  return 0;
}

static int lbmi_x_exchange_along(lbmi_t * lb, int dim, const lbmi_halo_sel_t * sel,
				 double * data, int layer, hipStream_t st) {
  double * const buf[4] = {lb->sendlo, lb->sendhi, lb->recvlo, lb->recvhi};
  return lbmi_x_exchange_dim(lb, dim, sel, data, 0, layer, buf, 0, 1, st);
}

static int lbmi_halo_generic(lbmi_t * lb, const lbmi_halo_sel_t sel[3],
			     double * data, hipStream_t st) {
  /* X, Y, Z in this order whichever of them goes over the ring: each pass
   * covers the full extent of the others, so edges and corners complete
   * (halo_swap.c:709-1063 does the same with its Cartesian neighbours) */
  for (int d = 0; d < 3; d++) {
    if (lbmi_dec(lb, d)) {
      int ifail = lbmi_x_exchange_along(lb, d, &sel[d], data, 0, st);
      if (ifail) return ifail;
    }
    else {
      KCHECK(lbmi_k_halo_copy(&lb->kp, d, &sel[d], data, 1, st));
    }
  }
  return 0;
}

int lbmi_halo(lbmi_t * lb, double * f, int scheme) {
  if (lb == NULL || f == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  HIPCHECK(hipSetDevice(lb->device));
  if (scheme == LBMI_HALO_FULL) {
    return lbmi_halo_generic(lb, lb->sel_full, f, lb->stream);
  }
  if (scheme == LBMI_HALO_REDUCED) {
    return lbmi_halo_generic(lb, lb->sel_reduced, f, lb->stream);
  }
  return lbmi_fail(LBMI_ERR_ARGUMENT, "halo scheme %d", scheme);
}

static const lbmi_halo_sel_t * lbmi_sel(const lbmi_t * lb, int scheme) {
  if (scheme == LBMI_HALO_FULL) return lb->sel_full;
  if (scheme == LBMI_HALO_REDUCED) return lb->sel_reduced;
  return NULL;
}

int lbmi_halo_x_count(lbmi_t * lb, int scheme, size_t * nsendlo,
		      size_t * nsendhi) {
  const lbmi_halo_sel_t * sel;
  if (lb == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  sel = lbmi_sel(lb, scheme);
  if (sel == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "halo scheme %d", scheme);
  if (lb->xdim != X || lb->multi) {
    return lbmi_fail(LBMI_ERR_UNSUPPORTED, "lbmi_halo_x_*: slabs along X only "
		     "(cartdim = %d: lbmi_halo / lbmi_lb_halo do the exchange)", lb->opts.cartdim);
  }
  /* sendlo feeds the neighbour's HIGH halo (components hi), sendhi its LOW */
  if (nsendlo) *nsendlo = (size_t) lb->kp.strx*(size_t) sel[X].nhi;
  if (nsendhi) *nsendhi = (size_t) lb->kp.strx*(size_t) sel[X].nlo;
  return 0;
}

int lbmi_halo_x_pack(lbmi_t * lb, const double * f, int scheme,
		     double * sendlo, double * sendhi) {
  const lbmi_halo_sel_t * sel;
  if (lb == NULL || !f || !sendlo || !sendhi) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  sel = lbmi_sel(lb, scheme);
  if (sel == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "halo scheme %d", scheme);
  HIPCHECK(hipSetDevice(lb->device));
  KCHECK(lbmi_k_halo_pack_x(&lb->kp, &sel[X], f, sendlo, sendhi, 0, 0, lb->stream));
  return 0;
}

int lbmi_halo_x_unpack(lbmi_t * lb, double * f, int scheme,
		       const double * recvlo, const double * recvhi) {
  const lbmi_halo_sel_t * sel;
  if (lb == NULL || !f || !recvlo || !recvhi) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  sel = lbmi_sel(lb, scheme);
  if (sel == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "halo scheme %d", scheme);
  HIPCHECK(hipSetDevice(lb->device));
  KCHECK(lbmi_k_halo_unpack_x(&lb->kp, &sel[X], f, recvlo, recvhi, 0, 0, lb->stream));
  return 0;
}

int lbmi_halo_yz(lbmi_t * lb, double * f, int scheme) {
  const lbmi_halo_sel_t * sel;
  if (lb == NULL || !f) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  sel = lbmi_sel(lb, scheme);
  if (sel == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "halo scheme %d", scheme);
  HIPCHECK(hipSetDevice(lb->device));
  KCHECK(lbmi_k_halo_copy(&lb->kp, Y, &sel[Y], f, 1, lb->stream));
  KCHECK(lbmi_k_halo_copy(&lb->kp, Z, &sel[Z], f, 1, lb->stream));
  return 0;
}

int lbmi_field_halo(lbmi_t * lb, int nel, double * data) {
  lbmi_halo_sel_t sel[3];
  if (lb == NULL || data == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  if (nel < 1 || nel > LBMI_NVEL_MAX) {
    return lbmi_fail(LBMI_ERR_ARGUMENT, "nel = %d (1..%d)", nel, LBMI_NVEL_MAX);
  }
  HIPCHECK(hipSetDevice(lb->device));
  {
    int ifail = lbmi_hydro_touch(lb, data, NULL);    /* hydro_u_halo of a lazy u */
    if (ifail) return ifail;
  }
  for (int d = 0; d < 3; d++) {
    memset(&sel[d], 0, sizeof(sel[d]));
    for (int n = 0; n < nel; n++) {
      sel[d].lo[sel[d].nlo++] = (int8_t) n;
      sel[d].hi[sel[d].nhi++] = (int8_t) n;
    }
  }
  return lbmi_halo_generic(lb, sel, data, lb->stream);
}

/*****************************************************************************
 *
 *  lbmi_moments  (stats_distribution.c:55-117, 201-350)
 *
 *****************************************************************************/

int lbmi_moments(lbmi_t * lb, const double * f, const char * status,
		 double out[9]) {
  if (lb == NULL || f == NULL || out == NULL) {
    return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  }
  HIPCHECK(hipSetDevice(lb->device));
  KCHECK(lbmi_k_moments(&lb->kp, f, status, lb->mom_work, lb->mom_out,
			lb->stream));
  HIPCHECK(hipMemcpyAsync(out, lb->mom_out, 9*sizeof(double),
			  hipMemcpyDeviceToHost, lb->stream));
  HIPCHECK(hipStreamSynchronize(lb->stream));
  return 0;
}

/*****************************************************************************
 *
 *  The lb_t-like stateful surface
 *
 *****************************************************************************/

int lbmi_lb_bind(lbmi_t * lb, double * f, double * fprime) {

  if (lb == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  HIPCHECK(hipSetDevice(lb->device));

  if (lb->owns_f) {
    hipFree(lb->f);
    hipFree(lb->fprime);
    lb->owns_f = 0;
  }
  lb->f = NULL;
  lb->fprime = NULL;
  lb->hydro_stale = 0;
  /* other arrays, another allocation: what was known about addresses seen
   * with the old pair need not hold for what lives at them now */
  memset(lb->known_zero, 0, sizeof(lb->known_zero));
  lbmi_run_graph_release(lb);
  lb->xsend_valid = 0; lb->halo_fresh = 0;
  lb->pending_halo = 0;
  lb->pending_prop = 0;
  lb->layout_swapped = 0;
  lb->early_prop = 0;
  lb->halo_seen = 0;
  lb->halo_done = 0;
  lb->blocked = 0;

  if (f == NULL) {
    size_t sz = sizeof(double)*(size_t) lb->kp.nsite*(size_t) lb->kp.nvel
    *(size_t) lb->opts.ndist;
    HIPCHECK(hipMalloc((void **) &lb->f, sz));
    HIPCHECK(hipMalloc((void **) &lb->fprime, sz));
    HIPCHECK(hipMemsetAsync(lb->f, 0, sz, lb->stream));
    HIPCHECK(hipMemsetAsync(lb->fprime, 0, sz, lb->stream));
    HIPCHECK(hipStreamSynchronize(lb->stream));
    lb->owns_f = 1;
  }
  else {
    if (fprime == NULL || fprime == f) {
      return lbmi_fail(LBMI_ERR_ARGUMENT, "lbmi_lb_bind: fprime invalid");
    }
    lb->f = f;
    lb->fprime = fprime;
  }

  return 0;
}

int lbmi_lb_pointers(lbmi_t * lb, double ** f, double ** fprime) {
  if (lb == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  if (f) *f = lb->f;
  if (fprime) *fprime = lb->fprime;
  return 0;
}

/* lb_model_swapf for a caller that keeps the two pointers in DEVICE memory
 * (lb->target->f, lb->target->fprime: members of the device copy of lb_t,
 * propagation.c:240-248): the current pair is written there by a one-thread
 * kernel on the handle's stream -- ordered with the step, no blocking copy. */

int lbmi_lb_pointers_store(lbmi_t * lb, double ** f_slot, double ** fprime_slot) {
  if (lb == NULL || f_slot == NULL || fprime_slot == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  HIPCHECK(hipSetDevice(lb->device));
  KCHECK(lbmi_k_store_pointers(f_slot, fprime_slot, lb->f, lb->fprime, lb->stream));
  return 0;
}

static void lbmi_swapf(lbmi_t * lb) {        /* lb_model_swapf */
  double * tmp = lb->f;
  lb->f = lb->fprime;
  lb->fprime = tmp;
  /* another array is current: what the last boundary launch of a slab left
   * in the send buffers belongs to the old one (lbmi_fused_step sets the
   * flag again after its own swap) */
  lb->xsend_valid = 0; lb->halo_fresh = 0;
}

/* INPLACE is honoured on a single rank without a communicator; with slabs
 * the same calls run the FUSED two-array path */

static int lbmi_inplace(const lbmi_t * lb) {
  return (lb->opts.mode == LBMI_MODE_INPLACE && lb->opts.cartsz == 1 &&
	  !lb->have_comm);
}

static int lbmi_deferred(const lbmi_t * lb) {
  return (lb->opts.mode == LBMI_MODE_FUSED ||
	  lb->opts.mode == LBMI_MODE_INPLACE ||
	  lb->opts.mode == LBMI_MODE_FUSED_HALO);
}

/* The blocked order of a deferred state (lbmi_kernels.hip, faddr): single
 * rank, every pull wrapped by index, and every site the kernel touches
 * inside the whole blocks of the array */

static int lbmi_blocked_ok(const lbmi_t * lb) {
  int nfull = lbmi_k_blocked_sites(&lb->kp);
  if (!lb->use_blocked || lb->opts.mode != LBMI_MODE_FUSED) return 0;
  /* two distributions ([site/256][n*nvel + p][site%256]): one rank, no
   * fluctuations (their collision has no variant in this order) */
  if (lb->opts.ndist != 1 &&
      (lb->opts.cartsz > 1 || lb->have_comm || lb->noise_state != NULL)) return 0;
  if ((lb->xdim != X || lb->multi) && (lb->opts.cartsz > 1 || lb->have_comm)) return 0;
  if (lb->opts.cartsz == 1 && !lb->have_comm) {
    /* every pull is wrapped by index: nothing beyond the interior planes */
    int last = (lb->kp.nhalo + lb->kp.nlocal[X])*lb->kp.strx;
    return (last <= nfull);
  }
  /* Slabs pull from the x halo planes as well. Of the high one, the last
   * nhalo rows in y (and the z halo of the row before) are never pulled from
   * (y and z wrap by index): the sites past the last whole block must lie
   * in there. */
  return (lb->kp.nsite - nfull <= lb->kp.nhalo*lb->kp.stry + lb->kp.nhalo);
}

/* Back to the reference's SoA order (same state, other addresses) */

static int lbmi_unblock(lbmi_t * lb) {
  if (!lb->blocked) return 0;
  KCHECK(lbmi_k_relayout_n(&lb->kp, lb->opts.ndist, lb->f, lb->fprime, 0, lb->stream));
  lbmi_swapf(lb);
  lb->blocked = 0;
  return 0;
}

/* FUSED step: exchange of the X planes on the comm stream overlapped with
 * the interior x-planes on the compute stream, boundary planes afterwards */

static int lbmi_fused_step(lbmi_t * lb, const lbmi_hydro_dev_t * h) {

  const int nh = lb->kp.nhalo;
  const int xlo = nh;
  const int xhi = nh + lb->kp.nlocal[X] - 1;
  const int wrapmask = lbmi_wrapmask(lb);
  int xsend = 0;
  int fresh = 0;
  int ifail;

  ifail = lbmi_time_begin(lb);
  if (ifail) return ifail;

  if (lb->opts.mode == LBMI_MODE_FUSED_HALO) {
    /* the halo swap has been done (and anything may have bounced back into
     * it): pull from the array as it is, halo sites included */
    if (lb->halo_fold && lb->opts.cartsz == 1 && !lb->have_comm && h->noise == NULL) {
      /* ... and compute the halo shell of the result on the way (every shell
       * site = the collision of its periodic image): the lb_halo that follows
       * has nothing left to do unless somebody writes to f first */
      KCHECK(lbmi_k_propagate_collide_halo(&lb->kp, lb->f, lb->fprime, h, lb->stream));
      fresh = 1;
    }
    else {
      KCHECK(lbmi_k_propagate_collide(&lb->kp, lb->f, lb->fprime, h, 0, 0,
				      xlo, xhi, 0, -1, NULL, lb->stream));
    }
  }
  else if (lb->opts.cartsz == 1 && !lb->have_comm) {
    /* lay: 0 SoA -> SoA; 1 SoA -> blocked; 2 blocked -> blocked */
    int lay = lbmi_blocked_ok(lb) ? (lb->blocked ? 2 : 1) : 0;
    if (lay == 0 && lb->blocked) {
      ifail = lbmi_unblock(lb);
      if (ifail) return ifail;
    }
    if (lb->nt_store_mode >= 0) {
      lb->kp.nt_store = lb->nt_store_mode;
    }
    else {
      /* A lattice whose two arrays fit the 256 MiB Infinity Cache is served
       * from it step after step: nontemporal stores would send it to HBM
       * (64^3: 19 -> 24 us). Beyond that they help (96^3: +15 %). */
      size_t bytes = 2*sizeof(double)*(size_t) lb->kp.nsite*(size_t) lb->kp.nvel;
      lb->kp.nt_store = (bytes > ((size_t) 256 << 20)) ? 1 : 0;
    }
    KCHECK(lbmi_k_propagate_collide(&lb->kp, lb->f, lb->fprime, h, wrapmask,
				    lay, xlo, xhi, 0, -1, NULL, lb->stream));
    lb->blocked = (lay != 0);
  }
  else if (lb->xdim != X) {
    /* A slab along Y (Z slabs never get here: lbmi_create; the code serves
     * either). Its boundary planes are not runs of x planes, so
     * there is no interior launch that leaves them out: ONE launch over all
     * x planes runs while the messages travel -- what it makes of the first
     * and the last plane of the decomposed direction comes from halo planes
     * nobody has filled and is overwritten -- and then the face launch does
     * those two planes against the exchange buffers and fills the send
     * buffers of the next exchange, whose messages leave right behind it
     * (lbmi_k_propagate_collide_face). With fluctuations, or x_direct 0: the
     * exchange into the halo planes first, then one launch. */
    const int dim = lb->xdim;
    const int direct = (lb->x_direct && h->noise == NULL);
    const lbmi_xbuf_t xbuf = {lb->fx[2], lb->fx[3], lb->fx[0], lb->fx[1]};
    const int pre = (direct && lb->xsend_valid);

    if (lb->blocked) {
      ifail = lbmi_unblock(lb);
      if (ifail) return ifail;
    }
    lb->kp.nt_store = 0;
    HIPCHECK(hipEventRecord(lb->ev_ready, lb->stream));
    if (!direct) {
      HIPCHECK(hipStreamWaitEvent(lb->comm_stream, lb->ev_ready, 0));
      ifail = lbmi_x_exchange_buf(lb, &lb->sel_reduced[dim], lb->f, 0, 0,
				  lb->fx, 0, 1, lb->comm_stream);
      if (ifail) return ifail;
      HIPCHECK(hipEventRecord(lb->ev_halo, lb->comm_stream));
      HIPCHECK(hipStreamWaitEvent(lb->stream, lb->ev_halo, 0));
      KCHECK(lbmi_k_propagate_collide(&lb->kp, lb->f, lb->fprime, h, wrapmask,
				      0, xlo, xhi, 0, -1, NULL, lb->stream));
    }
    else {
      KCHECK(lbmi_k_propagate_collide(&lb->kp, lb->f, lb->fprime, h, wrapmask,
				      0, xlo, xhi, 0, -1, NULL, lb->stream));
      if (!pre) {
	HIPCHECK(hipStreamWaitEvent(lb->comm_stream, lb->ev_ready, 0));
	ifail = lbmi_x_exchange_buf(lb, &lb->sel_reduced[dim], lb->f, 0, 0,
				    lb->fx, 0, 0, lb->comm_stream);
	if (ifail) return ifail;
	HIPCHECK(hipEventRecord(lb->ev_halo, lb->comm_stream));
      }
      HIPCHECK(hipStreamWaitEvent(lb->stream, lb->ev_halo, 0));
      KCHECK(lbmi_k_propagate_collide_face(&lb->kp, dim, lb->f, lb->fprime, h,
					   wrapmask, &xbuf, lb->stream));
      HIPCHECK(hipEventRecord(lb->ev_bnd, lb->stream));
      /* the messages of the NEXT step, now */
      HIPCHECK(hipStreamWaitEvent(lb->comm_stream, lb->ev_bnd, 0));
      ifail = lbmi_x_exchange_buf(lb, &lb->sel_reduced[dim], lb->fprime, 0, 0,
				  lb->fx, 1, 0, lb->comm_stream);
      if (ifail) return ifail;
      HIPCHECK(hipEventRecord(lb->ev_halo, lb->comm_stream));
      xsend = 1;
    }
    lb->blocked = 0;
  }
  else {
    /* The interior planes need no x halo: enqueue them first so that the
     * compute stream never idles while the host is busy enqueueing the
     * exchange (an RCCL group costs ~15 us of host time). ev_ready marks
     * "previous step complete", i.e. the boundary planes of f are final. */
    /* lay: 0 SoA -> SoA; 1 SoA -> blocked; 2 blocked -> blocked. Both
     * launches of a step and the exchange see f in ONE order. */
    int lay = lbmi_blocked_ok(lb) ? (lb->blocked ? 2 : 1) : 0;
    if (lay == 0 && lb->blocked) {
      ifail = lbmi_unblock(lb);
      if (ifail) return ifail;
    }
    if (lb->nt_store_mode >= 0) {
      lb->kp.nt_store = lb->nt_store_mode;
    }
    else {
      size_t bytes = 2*sizeof(double)*(size_t) lb->kp.nsite*(size_t) lb->kp.nvel;
      lb->kp.nt_store = (bytes > ((size_t) 256 << 20)) ? 1 : 0;
    }
    {
      /* x_direct (default): the boundary launch reads what arrived straight
       * from the receive buffers and leaves what the NEXT exchange sends in
       * the send buffers, so a steady-state step is: interior launch beside
       * (messages, then ONE boundary launch) -- no pack, no unpack kernel.
       * Otherwise: pack, messages, unpack into the halo planes of f. */
      /* (a fluctuating collision has no variant against the exchange
       * buffers: pack, messages, unpack) */
      const int direct = (lb->x_direct && lb->x_packed && h->noise == NULL);
      const int detail = (lb->timing_now && lb->nd < LBMI_NDETAIL);
      const lbmi_xbuf_t xbuf = {lb->fx[2], lb->fx[3], lb->fx[0], lb->fx[1]};
      const lbmi_xbuf_t * xb = direct ? &xbuf : NULL;
      hipStream_t bst = lb->x_concurrent ? lb->bnd_stream : lb->stream;

      /* direct form, steady state: the messages of THIS step were issued at
       * the end of the previous call, as soon as its boundary launch had
       * filled the send buffers -- they have had most of an interior launch
       * to arrive, not just the start of this one */
      const int pre = (direct && lb->xsend_valid);

      HIPCHECK(hipEventRecord(lb->ev_ready, lb->stream));
      if (detail) HIPCHECK(hipEventRecord(lb->evd[0][lb->nd], lb->stream));
      KCHECK(lbmi_k_propagate_collide(&lb->kp, lb->f, lb->fprime, h, wrapmask,
				      lay, xlo + 1, xhi - 1, 0, -1, NULL,
				      lb->stream));
      if (detail) HIPCHECK(hipEventRecord(lb->evd[1][lb->nd], lb->stream));

      if (!pre) {
	/* comm stream: the boundary planes of f are final at ev_ready */
	HIPCHECK(hipStreamWaitEvent(lb->comm_stream, lb->ev_ready, 0));
	if (detail && !direct) HIPCHECK(hipEventRecord(lb->evd[2][lb->nd], lb->comm_stream));
	ifail = lbmi_x_exchange_buf(lb, &lb->sel_reduced[X], lb->f, lb->blocked,
				    0, lb->fx, 0, !direct, lb->comm_stream);
	if (ifail) return ifail;
	if (detail && !direct) HIPCHECK(hipEventRecord(lb->evd[3][lb->nd], lb->comm_stream));
	HIPCHECK(hipEventRecord(lb->ev_halo, lb->comm_stream));
      }

      /* The two boundary planes depend on the state of the previous step
       * (complete at ev_ready) and on what arrived, not on the interior
       * launch of THIS step (they write other planes of fprime): with
       * x_concurrent they run on a third stream beside it, so that a step
       * costs the interior kernel, not interior + boundary + two stream
       * hand-overs. The compute stream joins at the end: whatever follows
       * sees the whole of fprime. */
      if (lb->x_concurrent) HIPCHECK(hipStreamWaitEvent(bst, lb->ev_ready, 0));
      HIPCHECK(hipStreamWaitEvent(bst, lb->ev_halo, 0));
      if (detail) HIPCHECK(hipEventRecord(lb->evd[4][lb->nd], bst));
      KCHECK(lbmi_k_propagate_collide(&lb->kp, lb->f, lb->fprime, h, wrapmask,
				      lay, xlo, xlo, xhi, (xhi > xlo) ? xhi : xhi - 1,
				      xb, bst));
      if (detail) HIPCHECK(hipEventRecord(lb->evd[5][lb->nd], bst));
      HIPCHECK(hipEventRecord(lb->ev_bnd, bst));
      if (lb->x_concurrent) HIPCHECK(hipStreamWaitEvent(lb->stream, lb->ev_bnd, 0));

      if (direct) {
	/* the messages of the NEXT step, now: what the boundary launch has
	 * just left in the send buffers is all they carry, and the receive
	 * buffers are free once it has read them (ev_bnd). If the next call
	 * is not a fused step, they have travelled for nothing (whatever
	 * changes f in another way drops xsend_valid, and the step after
	 * packs and exchanges afresh). Every rank runs the same sequence of
	 * calls, so the messages pair up as before. */
	HIPCHECK(hipStreamWaitEvent(lb->comm_stream, lb->ev_bnd, 0));
	if (detail) HIPCHECK(hipEventRecord(lb->evd[2][lb->nd], lb->comm_stream));
	ifail = lbmi_x_exchange_buf(lb, &lb->sel_reduced[X], lb->fprime, 0, 0,
				    lb->fx, 1, 0, lb->comm_stream);
	if (ifail) return ifail;
	if (detail) HIPCHECK(hipEventRecord(lb->evd[3][lb->nd], lb->comm_stream));
	HIPCHECK(hipEventRecord(lb->ev_halo, lb->comm_stream));
      }
      if (detail) lb->nd += 1;
      xsend = direct;            /* of fprime, which becomes f below */
    }
    lb->blocked = (lay != 0);
  }

  ifail = lbmi_time_end(lb);
  if (ifail) return ifail;

  lbmi_swapf(lb);
  lb->xsend_valid = xsend;
  lb->halo_fresh = fresh;

  return 0;
}

/*****************************************************************************
 *
 *  lbmi_noise_set  (noise_t as lb_collide borrows it, collision.c:476-518)
 *
 *****************************************************************************/

int lbmi_noise_set(lbmi_t * lb, unsigned int * state, long long nsites,
		   double kt, int ghosts_on) {

  if (lb == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  if (state != NULL) {
    if (lb->opts.nvel != 19) {
      return lbmi_fail(LBMI_ERR_ARGUMENT, "fluctuations: D3Q19 only "
		       "(noise.h:18, NNOISE_MAX = 10)");
    }
    if (nsites < (long long) lb->kp.nsite) {
      return lbmi_fail(LBMI_ERR_ARGUMENT, "fluctuations: %lld generator states "
		       "for %d sites", nsites, lb->kp.nsite);
    }
    if (!(kt >= 0.0)) return lbmi_fail(LBMI_ERR_ARGUMENT, "fluctuations: kt < 0");
  }
  if (lb->noise_state != state || lb->noise_stride != nsites ||
      lb->noise_kt != kt || lb->noise_ghosts != (ghosts_on != 0)) {
    /* a captured lbmi_lb_run carries the old generator / temperature */
    lbmi_run_graph_release(lb);
  }
  lb->noise_state = state;
  lb->noise_stride = nsites;
  lb->noise_kt = kt;
  lb->noise_ghosts = (ghosts_on != 0);

  return 0;
}

/*****************************************************************************
 *
 *  lbmi_lb_collide  (lb_collide, collision.c:143-163)
 *
 *****************************************************************************/

static int lbmi_lb_collide_dev(lbmi_t * lb, const lbmi_hydro_dev_t * hp);

/* What lbmi_lb_collide hands to the kernels for a hydro object: the generator
 * of lbmi_noise_set joined, a force array known to hold zeros left out (F =
 * the body force, bit for bit). Also the key of a captured lbmi_lb_run. */

static lbmi_hydro_dev_t lbmi_hydro_effective(const lbmi_t * lb,
					     const lbmi_hydro_t * hydro) {
  lbmi_hydro_dev_t h = lbmi_hydro_dev(hydro);
  if (lb->noise_state != NULL) {
    h.noise = lb->noise_state;
    h.noise_stride = lb->noise_stride;
    h.noise_kt = lb->noise_kt;
    h.noise_ghosts = lb->noise_ghosts;
  }
  if (lbmi_known_zero(lb, h.force)) h.force = NULL;
  return h;
}

int lbmi_lb_collide(lbmi_t * lb, const lbmi_hydro_t * hydro) {

  lbmi_hydro_dev_t h;

  if (lb == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  h = lbmi_hydro_effective(lb, hydro);
  if (lb->f == NULL) return lbmi_fail(LBMI_ERR_STATE, "no distributions bound");
  if (lb->opts.ndist != 1) {
    return lbmi_fail(LBMI_ERR_STATE, "ndist = 2: lbmi_lb_collide_binary");
  }
  HIPCHECK(hipSetDevice(lb->device));

  if (lb->noise_state != NULL) {
    /* isothermal fluctuations: collide in place, or the pull of FUSED_HALO */
    if (lb->opts.nvel != 19) {
      return lbmi_fail(LBMI_ERR_STATE, "fluctuations: D3Q19 only (noise.h:18)");
    }
    if (lbmi_inplace(lb)) {
      return lbmi_fail(LBMI_ERR_STATE, "fluctuations: not in LBMI_MODE_INPLACE "
		       "(lbmi_lb_mode_set)");
    }
  }

  /* whatever an earlier collision still owed is superseded by this one */
  lb->hydro_stale = 0;
  if (lb->hydro_lazy == 1 && lbmi_deferred(lb) && !lbmi_inplace(lb) &&
      (h.rho != NULL || h.u != NULL)) {
    int ifail;
    lb->lazy_h = h;
    for (int ia = 0; ia < 3; ia++) lb->lazy_fbody[ia] = lb->kp.fbody[ia];
    h.rho = NULL;
    h.u = NULL;
    ifail = lbmi_lb_collide_dev(lb, &h);
    if (ifail == 0) lb->hydro_stale = 1;
    return ifail;
  }
  if (lb->hydro_lazy == 2 && lbmi_deferred(lb) && !lbmi_inplace(lb) && h.rho != NULL) {
    /* rho alone on demand: u is stored (somebody reads it before the next
     * collision: the advection of an order parameter, a viscosity model),
     * the density has no reader on the device in the reference but the open
     * boundaries (rho = sum f'_p needs neither the force nor u) */
    int ifail;
    lb->lazy_h = h;
    lb->lazy_h.u = NULL;
    lb->lazy_h.force = NULL;
    for (int ia = 0; ia < 3; ia++) lb->lazy_fbody[ia] = lb->kp.fbody[ia];
    h.rho = NULL;
    lbmi_known_zero_drop(lb, h.u);
    ifail = lbmi_lb_collide_dev(lb, &h);
    if (ifail == 0) lb->hydro_stale = 1;
    return ifail;
  }
  lbmi_known_zero_drop(lb, h.rho);
  lbmi_known_zero_drop(lb, h.u);
  return lbmi_lb_collide_dev(lb, &h);
}

static int lbmi_lb_collide_dev(lbmi_t * lb, const lbmi_hydro_dev_t * hp) {

  const lbmi_hydro_dev_t h = *hp;

  if (lbmi_inplace(lb)) {
    int ifail;
    if (lb->early_prop) {
      return lbmi_fail(LBMI_ERR_STATE, "lb_collide: lb_propagation of the "
		       "previous step has not been called");
    }
    if (lb->pending_prop) {
      /* P(t) C(t+1) P(t+1) in place */
      if (!lb->pending_halo) {
	return lbmi_fail(LBMI_ERR_STATE, "propagation pending without halo");
      }
      ifail = lbmi_time_begin(lb);
      if (ifail) return ifail;
      KCHECK(lbmi_k_aa_odd(&lb->kp, lb->f, &h, 7, lb->layout_swapped,
			   lb->stream));
      ifail = lbmi_time_end(lb);
      if (ifail) return ifail;
      lb->pending_prop = 0;
      lb->pending_halo = 0;
      lb->layout_swapped = 0;
      lb->early_prop = 1;
      lb->halo_seen = 0;
      return 0;
    }
    if (lb->pending_halo) {
      return lbmi_fail(LBMI_ERR_STATE, "lb_collide after lb_halo without "
		       "lb_propagation");
    }
    if (lb->layout_swapped) {
      KCHECK(lbmi_k_aa_unswap(&lb->kp, lb->f, lb->stream));
      lb->layout_swapped = 0;
    }
    /* C(t) in place, stored into swapped slots */
    ifail = lbmi_time_begin(lb);
    if (ifail) return ifail;
    KCHECK(lbmi_k_aa_even(&lb->kp, lb->f, &h, lb->stream));
    ifail = lbmi_time_end(lb);
    if (ifail) return ifail;
    lb->layout_swapped = 1;
    return 0;
  }

  if (lb->pending_prop) {
    /* FUSED: propagation(t) and, by index wrap / overlapped exchange, the
     * halo swap(t), are done inside the collision(t+1) kernel */
    if (!lb->pending_halo) {
      return lbmi_fail(LBMI_ERR_STATE, "propagation pending without halo");
    }
    lb->pending_prop = 0;
    lb->pending_halo = 0;
    lb->halo_done = 0;
    return lbmi_fused_step(lb, &h);
  }

  if (lb->pending_halo) {
    return lbmi_fail(LBMI_ERR_STATE, "lb_collide after lb_halo without "
		     "lb_propagation");
  }

  if (lb->blocked) {
    int ifail = lbmi_unblock(lb);
    if (ifail) return ifail;
  }
  lb->xsend_valid = 0; lb->halo_fresh = 0;               /* f changes */
  if (h.noise == NULL && lb->opts.ndist == 1 && lb->eager_oop) {
    /* out of place into the other array, then the two change roles (as
     * lb_propagation's swap): in-place read-modify-write is the slower way to
     * move the same bytes here (lbmi_kernels.hip, k_collide_to) */
    KCHECK(lbmi_k_collide_to(&lb->kp, lb->f, lb->fprime, &h, lb->stream));
    lbmi_swapf(lb);
    return 0;
  }
  KCHECK(lbmi_k_collide(&lb->kp, lb->f, &h, lb->stream));

  return 0;
}

/*****************************************************************************
 *
 *  Flat walls and bounce-back on links (wall.c)
 *
 *****************************************************************************/

enum {LBMI_MAP_FLUID = 0, LBMI_MAP_BOUNDARY = 1};        /* map.h:23 */

static void lbmi_slip_release(lbmi_t * lb) {
  if (lb->slip_k_dev) hipFree(lb->slip_k_dev);
  if (lb->slip_q_dev) hipFree(lb->slip_q_dev);
  if (lb->slip_s_dev) hipFree(lb->slip_s_dev);
  lb->slip_k_dev = NULL;
  lb->slip_q_dev = NULL;
  lb->slip_s_dev = NULL;
  for (int k = 0; k < 3; k++) {
    free(lb->slip_host[k]);
    lb->slip_host[k] = NULL;
  }
  lb->slip_active = 0;
}

static void lbmi_wall_release(lbmi_t * lb) {
  for (int k = 0; k < 4; k++) {
    if (lb->link_dev[k]) hipFree(lb->link_dev[k]);
    free(lb->link_host[k]);
    lb->link_dev[k] = NULL;
    lb->link_host[k] = NULL;
  }
  lbmi_slip_release(lb);
  if (lb->wall_part) hipFree(lb->wall_part);
  if (lb->wall_fnet) hipFree(lb->wall_fnet);
  lb->wall_part = NULL;
  lb->wall_part_nblk = 0;
  lb->wall_fnet = NULL;
  lb->wall_fnet_ext = NULL;
  lb->wall_seen[0] = NULL;
  lb->wall_seen[1] = NULL;
  lb->wall_seen_nlink = 0;
  lb->nlink = 0;
}

static int lbmi_wall_upload(lbmi_t * lb, int nlink);

static int lbmi_wall_args(const lbmi_t * lb, const int isboundary[3]) {
  if (lb == NULL || isboundary == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  return 0;
}

int lbmi_wall_map(lbmi_t * lb, const int isboundary[3], char * status) {
  char * host = NULL;
  const int h = lb ? lb->kp.nhalo : 0;
  size_t ns;
  int ifail = lbmi_wall_args(lb, isboundary);
  if (ifail) return ifail;
  if (status == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  HIPCHECK(hipSetDevice(lb->device));
  ns = (size_t) lb->kp.nsite;
  host = (char *) malloc(ns);
  if (host == NULL) return lbmi_fail(LBMI_ERR_HIP, "wall map");
  if (hipMemcpy(host, status, ns, hipMemcpyDeviceToHost) != hipSuccess) {
    free(host);
    return lbmi_fail(LBMI_ERR_HIP, "wall map: device to host");
  }
  for (int ic = 0; ic < lb->kp.nall[X]; ic++) {
    for (int jc = 0; jc < lb->kp.nall[Y]; jc++) {
      for (int kc = 0; kc < lb->kp.nall[Z]; kc++) {
	/* coordinate 0 or nlocal + 1 in the reference's numbering */
	int wall = 0;
	/* X slabs: the global coordinate 0 lies below the first rank, the
	 * global ntotal + 1 above the last (noffset in wall.c:1236-1240) */
	{
	  const int c[3] = {ic, jc, kc};
	  for (int d = 0; d < 3; d++) {
	    /* in the decomposed direction only the first rank has the low
	     * wall, only the last one the high wall */
	    const int first = (lb->ccoord[d] == 0);
	    const int last = (lb->ccoord[d] == lb->csz[d] - 1);
	    if (isboundary[d] && c[d] == h - 1 && first) wall = 1;
	    if (isboundary[d] && c[d] == h + lb->kp.nlocal[d] && last) wall = 1;
	  }
	}
	if (wall) {
	  host[(size_t) ic*lb->kp.strx + (size_t) jc*lb->kp.stry + kc] = LBMI_MAP_BOUNDARY;
	}
      }
    }
  }
  if (hipMemcpy(status, host, ns, hipMemcpyHostToDevice) != hipSuccess) {
    free(host);
    return lbmi_fail(LBMI_ERR_HIP, "wall map: host to device");
  }
  free(host);
  return 0;
}

/* link_host -> link_dev, and the momentum accumulator */

static int lbmi_wall_upload(lbmi_t * lb, int nlink) {
  for (int k = 0; k < 4; k++) {
    size_t sz = sizeof(int)*(size_t) (nlink > 0 ? nlink : 1);
    if (hipMalloc((void **) &lb->link_dev[k], sz) != hipSuccess ||
	hipMemcpy(lb->link_dev[k], lb->link_host[k], sz, hipMemcpyHostToDevice) != hipSuccess) {
      lbmi_wall_release(lb);
      return lbmi_fail(LBMI_ERR_HIP, "wall links: device arrays");
    }
  }
  if (hipMalloc((void **) &lb->wall_fnet, 3*sizeof(double)) != hipSuccess ||
      hipMemset(lb->wall_fnet, 0, 3*sizeof(double)) != hipSuccess) {
    lbmi_wall_release(lb);
    return lbmi_fail(LBMI_ERR_HIP, "wall momentum workspace");
  }
  lb->nlink = nlink;
  return 0;
}

int lbmi_wall_links_build(lbmi_t * lb, const char * status,
			  const int isboundary[3], int * nlink_out) {
  char * host = NULL;
  size_t ns;
  int nlink = 0;
  int iw = -1;
  const int h = lb ? lb->kp.nhalo : 0;
  int ifail = lbmi_wall_args(lb, isboundary);
  if (ifail) return ifail;
  if (status == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  HIPCHECK(hipSetDevice(lb->device));
  HIPCHECK(hipStreamSynchronize(lb->stream));
  lbmi_wall_release(lb);

  ns = (size_t) lb->kp.nsite;
  host = (char *) malloc(ns);
  if (host == NULL) return lbmi_fail(LBMI_ERR_HIP, "wall links");
  if (hipMemcpy(host, status, ns, hipMemcpyDeviceToHost) != hipSuccess) {
    free(host);
    return lbmi_fail(LBMI_ERR_HIP, "wall links: device to host");
  }
  if (isboundary[X] + isboundary[Y] + isboundary[Z] == 1) {
    iw = isboundary[X] ? X : (isboundary[Y] ? Y : Z);
  }

  /* wall_init_boundaries: count, allocate, fill (wall.c:381-470) */
  for (int pass = 0; pass < 2; pass++) {
    int n = 0;
    for (int ic = h; ic < h + lb->kp.nlocal[X]; ic++) {
      for (int jc = h; jc < h + lb->kp.nlocal[Y]; jc++) {
	for (int kc = h; kc < h + lb->kp.nlocal[Z]; kc++) {
	  size_t i = (size_t) ic*lb->kp.strx + (size_t) jc*lb->kp.stry + kc;
	  if (host[i] != LBMI_MAP_FLUID) continue;
	  for (int p = 1; p < lb->kp.nvel; p++) {
	    size_t j = i + lb->cv[p][X]*lb->kp.strx + lb->cv[p][Y]*lb->kp.stry
	      + lb->cv[p][Z];
	    if (host[j] != LBMI_MAP_BOUNDARY) continue;
	    if (pass == 1) {
	      lb->link_host[0][n] = (int) i;
	      lb->link_host[1][n] = (int) j;
	      lb->link_host[2][n] = p;
	      lb->link_host[3][n] = 0;                       /* WALL_UZERO */
	      /* wall_init_uw (wall.c:864-890) */
	      if (iw >= 0 && lb->cv[p][iw] == -1) lb->link_host[3][n] = 2;
	      if (iw >= 0 && lb->cv[p][iw] == +1) lb->link_host[3][n] = 1;
	    }
	    n += 1;
	  }
	}
      }
    }
    if (pass == 0) {
      nlink = n;
      for (int k = 0; k < 4 && ifail == 0; k++) {
	lb->link_host[k] = (int *) calloc((size_t) (nlink > 0 ? nlink : 1), sizeof(int));
	if (lb->link_host[k] == NULL) ifail = lbmi_fail(LBMI_ERR_HIP, "wall links");
      }
      if (ifail) break;
    }
  }
  free(host);
  if (ifail) {
    lbmi_wall_release(lb);
    return ifail;
  }

    ifail = lbmi_wall_upload(lb, nlink);
  if (ifail) return ifail;
  if (nlink_out) *nlink_out = nlink;
  return 0;
}

/* wall_init_boundaries done by the caller (the reference keeps the result in
 * the HOST arrays wall->linki, linkj, linkp, linku, wall.c:399-451): every
 * record is checked here, on the host, before a kernel can see it, and the
 * device copies belong to the handle. */

int lbmi_wall_links_set(lbmi_t * lb, int nlink, const int * linki,
			const int * linkj, const int * linkp,
			const int * linku) {
  const int * in[4] = {linki, linkj, linkp, linku};
  if (lb == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  if (nlink < 0 || (nlink > 0 && (!linki || !linkj || !linkp || !linku))) {
    return lbmi_fail(LBMI_ERR_ARGUMENT, "lbmi_wall_links_set: bad argument");
  }
  for (int n = 0; n < nlink; n++) {
    const int p = linkp[n];
    long long j;
    if (p < 1 || p >= lb->kp.nvel || linki[n] < 0 ||
	(long long) linki[n] >= lb->kp.nsite || linku[n] < 0 || linku[n] > 2) {
      return lbmi_fail(LBMI_ERR_ARGUMENT, "link %d of %d: (i, j, p, u) = (%d, "
		       "%d, %d, %d) is not a link of this lattice (nsite %lld, "
		       "nvel %d)", n, nlink, linki[n], linkj[n], p, linku[n],
		       lb->kp.nsite, lb->kp.nvel);
    }
    j = (long long) linki[n] + lb->cv[p][X]*(long long) lb->kp.strx
      + lb->cv[p][Y]*(long long) lb->kp.stry + lb->cv[p][Z];
    if ((long long) linkj[n] != j || j < 0 || j >= lb->kp.nsite) {
      return lbmi_fail(LBMI_ERR_ARGUMENT, "link %d of %d: j = %d is not "
		       "i + c_p = %lld (i %d, p %d)", n, nlink, linkj[n], j,
		       linki[n], p);
    }
  }
  HIPCHECK(hipSetDevice(lb->device));
  HIPCHECK(hipStreamSynchronize(lb->stream));
  lbmi_wall_release(lb);
  for (int k = 0; k < 4; k++) {
    lb->link_host[k] = (int *) calloc((size_t) (nlink > 0 ? nlink : 1), sizeof(int));
    if (lb->link_host[k] == NULL) {
      lbmi_wall_release(lb);
      return lbmi_fail(LBMI_ERR_HIP, "wall links");
    }
    if (nlink > 0) memcpy(lb->link_host[k], in[k], sizeof(int)*(size_t) nlink);
  }
  return lbmi_wall_upload(lb, nlink);
}

int lbmi_wall_links(lbmi_t * lb, int * linki, int * linkj, int * linkp,
		    int * linku) {
  int * out[4] = {linki, linkj, linkp, linku};
  if (lb == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  if (lb->link_host[0] == NULL) return lbmi_fail(LBMI_ERR_STATE, "no links built");
  for (int k = 0; k < 4; k++) {
    if (out[k]) memcpy(out[k], lb->link_host[k], sizeof(int)*(size_t) lb->nlink);
  }
  return 0;
}

int lbmi_wall_velocity_set(lbmi_t * lb, const double ubot[3],
			   const double utop[3]) {
  if (lb == NULL || ubot == NULL || utop == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  for (int ia = 0; ia < 3; ia++) {
    lb->wall_ubot[ia] = ubot[ia];
    lb->wall_utop[ia] = utop[ia];
  }
  return 0;
}

/* The map the bounce-back kernels test for MAP_COLLOID (or NULL: no test) */

int lbmi_wall_status_set(lbmi_t * lb, const char * status) {
  if (lb == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  lb->wall_status = status;
  return 0;
}

/* The bounce-back kernels never dereference a record that would address
 * outside f: they skip it and leave its index + 1 in lb->wall_err (pinned
 * host memory the device writes). Arrays of the handle were checked on the
 * host when they were made; arrays the caller owns are checked the first time
 * they are seen (one synchronisation), and from then on whatever an earlier
 * launch left is reported by the next call, without waiting. */

static int lbmi_wall_err_report(lbmi_t * lb) {
  if (lb->wall_err != NULL && *lb->wall_err != 0) {
    const int n = *lb->wall_err - 1;
    *lb->wall_err = 0;
    return lbmi_fail(LBMI_ERR_ARGUMENT, "bounce-back: link record %d would "
		     "address outside the distributions (it was skipped; need "
		     "0 <= i, j < nsite, 1 <= p < nvel, u in {0, 1, 2})", n);
  }
  return 0;
}

static int lbmi_wall_after_launch(lbmi_t * lb, const void * a, const void * b,
				  int nlink) {
  if (a == (const void *) lb->link_dev[0]) return 0;      /* host-checked */
  if (a != lb->wall_seen[0] || b != lb->wall_seen[1] ||
      nlink != lb->wall_seen_nlink) {
    HIPCHECK(hipStreamSynchronize(lb->stream));
    lb->wall_seen[0] = a;
    lb->wall_seen[1] = b;
    lb->wall_seen_nlink = nlink;
  }
  return lbmi_wall_err_report(lb);
}

int lbmi_wall_fnet_bind(lbmi_t * lb, double * fnet) {
  if (lb == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  lb->wall_fnet_ext = fnet;
  return 0;
}

/* wall_bbl on link arrays owned by the caller (the reference keeps
 * wall->target->linki, linkj, linkp, linku and fnet on the device) */

int lbmi_wall_bbl_arrays(lbmi_t * lb, int nlink, const int * linki,
			 const int * linkj, const int * linkp,
			 const int * linku, const double ubot[3],
			 const double utop[3], double * fnet) {
  lbmi_wall_tab_t tab;
  double na[LBMI_NVEL_MAX];
  double ma[LBMI_NVEL_MAX*LBMI_NVEL_MAX];
  if (lb == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  if (lb->f == NULL) return lbmi_fail(LBMI_ERR_STATE, "no distributions bound");
  if (lb->opts.mode != LBMI_MODE_EAGER &&
      !(lb->opts.mode == LBMI_MODE_FUSED_HALO && !lb->pending_prop)) {
    /* bounce-back acts on the post-collision state between lb_halo and
     * lb_propagation: that state must exist (ludwig.c:836-858) */
    return lbmi_fail(LBMI_ERR_STATE, "walls need LBMI_MODE_EAGER or "
		     "LBMI_MODE_FUSED_HALO (before lb_propagation)");
  }
  if (nlink == 0) return 0;                          /* wall.c:967 */
  if (nlink < 0 || !linki || !linkj || !linkp || !linku || !ubot || !utop || !fnet) {
    return lbmi_fail(LBMI_ERR_ARGUMENT, "lbmi_wall_bbl_arrays: bad argument");
  }
  lb->halo_fresh = 0;                /* f is written: a halo swap after this one is real */
  HIPCHECK(hipSetDevice(lb->device));
  if (lb->wall_part == NULL || lb->wall_part_nblk < lbmi_k_wall_nblk(nlink)) {
    if (lb->wall_part) HIPCHECK(hipFree(lb->wall_part));
    lb->wall_part = NULL;
    lb->wall_part_nblk = lbmi_k_wall_nblk(nlink);
    HIPCHECK(hipMalloc((void **) &lb->wall_part,
		       sizeof(double)*3*(size_t) lb->wall_part_nblk));
  }
  memset(&tab, 0, sizeof(tab));
  tab.nvel = lb->kp.nvel;
  tab.ndist = lb->opts.ndist;
  tab.rho0 = lb->rho0;
  if (lbmi_k_model(lb->kp.nvel, &tab.cv[0][0], tab.wv, na, ma) != 0) {
    return lbmi_fail(LBMI_ERR_UNSUPPORTED, "model tables");
  }
  for (int ia = 0; ia < 3; ia++) {
    tab.uw[0][ia] = 0.0;
    tab.uw[1][ia] = utop[ia];
    tab.uw[2][ia] = ubot[ia];
  }
    KCHECK(lbmi_k_wall_bbl(&lb->kp, &tab, lb->f, nlink, linki, linkj, linkp,
			 linku, lb->wall_status, lb->wall_part, fnet,
			 lb->wall_err, lb->stream));
  return lbmi_wall_after_launch(lb, linki, linku, nlink);
}

/* wall_slip (wall.c:285-316): faces, then the 12 edges XB_YB XB_YT XB_ZB
 * XB_ZT XT_YB XT_YT XT_ZB XT_ZT YB_ZB YB_ZT YT_ZB YT_ZT (wall.h:25-41) */

static int lbmi_slip_table(const double sbot[3], const double stop[3],
			   double s[19]) {
  const double face[3][2] = {{sbot[X], stop[X]}, {sbot[Y], stop[Y]},
			     {sbot[Z], stop[Z]}};
  int active = 0;
  s[0] = 0.0;
  for (int ia = 0; ia < 3; ia++) {
    s[1 + 2*ia] = face[ia][0];
    s[2 + 2*ia] = face[ia][1];
    if (face[ia][0] != 0.0 || face[ia][1] != 0.0) active = 1;
  }
  for (int ta = 0; ta < 2; ta++) {
    for (int tb = 0; tb < 2; tb++) {
      s[7 + 4*ta + tb]  = 0.5*(face[X][ta] + face[Y][tb]);
      s[9 + 4*ta + tb]  = 0.5*(face[X][ta] + face[Z][tb]);
      s[15 + 2*ta + tb] = 0.5*(face[Y][ta] + face[Z][tb]);
    }
  }
  return active;
}

int lbmi_wall_slip_set(lbmi_t * lb, const char * status,
		       const double sbot[3], const double stop[3]) {
  char * host = NULL;
  int8_t * q8 = NULL;
  int8_t * s8 = NULL;
  size_t ns;
  int nlink;
  int ifail = 0;
  double stab[19];

  if (lb == NULL || sbot == NULL || stop == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  if (lb->link_host[0] == NULL) return lbmi_fail(LBMI_ERR_STATE, "no links built");
  for (int ia = 0; ia < 3; ia++) {                    /* wall_slip_valid */
    if (!(sbot[ia] >= 0.0 && sbot[ia] <= 1.0) || !(stop[ia] >= 0.0 && stop[ia] <= 1.0)) {
      return lbmi_fail(LBMI_ERR_ARGUMENT, "slip fractions must lie in [0, 1]");
    }
  }
  HIPCHECK(hipSetDevice(lb->device));
  HIPCHECK(hipStreamSynchronize(lb->stream));
  lbmi_slip_release(lb);
  if (!lbmi_slip_table(sbot, stop, stab)) return 0;
  if (status == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");

  nlink = lb->nlink;
  ns = (size_t) lb->kp.nsite;
  host = (char *) malloc(ns);
  q8 = (int8_t *) calloc((size_t) nlink + 1, 1);
  s8 = (int8_t *) calloc((size_t) nlink + 1, 1);
  for (int k = 0; k < 3; k++) {
    lb->slip_host[k] = (int *) calloc((size_t) nlink + 1, sizeof(int));
    if (lb->slip_host[k] == NULL) ifail = 1;
  }
  if (host == NULL || q8 == NULL || s8 == NULL) ifail = 1;
  if (!ifail && hipMemcpy(host, status, ns, hipMemcpyDeviceToHost) != hipSuccess) ifail = 1;

  for (int n = 0; n < nlink && !ifail; n++) {
    const int i = lb->link_host[0][n];
    const int p = lb->link_host[2][n];
    const ptrdiff_t str[3] = {lb->kp.strx, lb->kp.stry, 1};
    int wn[3], wt[3], cdotn = 0, modwn = 0, modwt = 0;
    /* wall_link_normal: the axis neighbours of i along c_p that are not fluid */
    for (int ia = 0; ia < 3; ia++) {
      wn[ia] = (host[i + str[ia]*lb->cv[p][ia]] != LBMI_MAP_FLUID) ? -lb->cv[p][ia] : 0;
      cdotn += lb->cv[p][ia]*wn[ia];
      modwn += wn[ia]*wn[ia];
    }
    if (modwn == 0 || modwn != -cdotn) {
      /* a convex edge: wall.c:557-558 asserts this away (flat walls only) */
      ifail = 2;
      break;
    }
    for (int ia = 0; ia < 3; ia++) {
      wt[ia] = lb->cv[p][ia] - cdotn*wn[ia]/modwn;
      modwt += wt[ia]*wt[ia];
    }
    if (modwt == 0) {
      /* nothing tangential: plain bounce-back (k, q valid but unused) */
      lb->slip_host[0][n] = i;
      lb->slip_host[1][n] = p;
      lb->slip_host[2][n] = 0;
    }
    else {
      int q = -1, s = 0;
      for (int m = 0; m < lb->kp.nvel; m++) {
	if (lb->cv[m][X] == -2*wn[X] - lb->cv[p][X] &&
	    lb->cv[m][Y] == -2*wn[Y] - lb->cv[p][Y] &&
	    lb->cv[m][Z] == -2*wn[Z] - lb->cv[p][Z]) q = m;
      }
      if (q <= 0) { ifail = 2; break; }
      if (modwn == 1) {
	for (int ia = 0; ia < 3; ia++) {
	  if (wn[ia] == +1) s = 1 + 2*ia;
	  if (wn[ia] == -1) s = 2 + 2*ia;
	}
      }
      if (modwn == 2) {
	if (wn[X] != 0 && wn[Y] != 0) s = 7 + 4*(wn[X] == -1) + (wn[Y] == -1);
	if (wn[X] != 0 && wn[Z] != 0) s = 9 + 4*(wn[X] == -1) + (wn[Z] == -1);
	if (wn[Y] != 0 && wn[Z] != 0) s = 15 + 2*(wn[Y] == -1) + (wn[Z] == -1);
      }
      lb->slip_host[0][n] = (int) (i + str[X]*wt[X] + str[Y]*wt[Y] + wt[Z]);
      lb->slip_host[1][n] = q;
      lb->slip_host[2][n] = s;                        /* corners: 0 */
    }
    q8[n] = (int8_t) lb->slip_host[1][n];
    s8[n] = (int8_t) lb->slip_host[2][n];
  }

  if (!ifail) {
    size_t n1 = (size_t) nlink + 1;
    if (hipMalloc((void **) &lb->slip_k_dev, n1*sizeof(int)) != hipSuccess ||
	hipMalloc((void **) &lb->slip_q_dev, n1) != hipSuccess ||
	hipMalloc((void **) &lb->slip_s_dev, n1) != hipSuccess ||
	hipMemcpy(lb->slip_k_dev, lb->slip_host[0], n1*sizeof(int), hipMemcpyHostToDevice) != hipSuccess ||
	hipMemcpy(lb->slip_q_dev, q8, n1, hipMemcpyHostToDevice) != hipSuccess ||
	hipMemcpy(lb->slip_s_dev, s8, n1, hipMemcpyHostToDevice) != hipSuccess) {
      ifail = 1;
    }
  }
  free(host);
  free(q8);
  free(s8);
  if (ifail) {
    lbmi_slip_release(lb);
    if (ifail == 2) {
      return lbmi_fail(LBMI_ERR_UNSUPPORTED, "slip: a link without a wall "
		       "normal (solid sites other than flat walls)");
    }
    return lbmi_fail(LBMI_ERR_HIP, "slip links");
  }
  memcpy(lb->slip_s, stab, sizeof(stab));
  lb->slip_active = 1;
  return 0;
}

int lbmi_wall_slip_links(lbmi_t * lb, int * linkk, int * linkq, int * links) {
  int * out[3] = {linkk, linkq, links};
  if (lb == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  if (!lb->slip_active) return lbmi_fail(LBMI_ERR_STATE, "slip is not active");
  for (int k = 0; k < 3; k++) {
    if (out[k]) memcpy(out[k], lb->slip_host[k], sizeof(int)*(size_t) lb->nlink);
  }
  return 0;
}

/* wall_init_boundaries_slip done by the caller (HOST arrays wall->linkk,
 * linkq, links in the reference's types, wall.h:79-81, and the table
 * wall->param->slip.s): checked here, copied, owned by the handle. */

int lbmi_wall_slip_links_set(lbmi_t * lb, const int * linkk,
			     const signed char * linkq,
			     const signed char * links, const double stab[19]) {
  size_t n1;
  if (lb == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  if (lb->link_host[0] == NULL) return lbmi_fail(LBMI_ERR_STATE, "no links built");
  if (!linkk || !linkq || !links || !stab) {
    return lbmi_fail(LBMI_ERR_ARGUMENT, "lbmi_wall_slip_links_set: NULL");
  }
  for (int n = 0; n < lb->nlink; n++) {
    if (linkk[n] < 0 || (long long) linkk[n] >= lb->kp.nsite || linkq[n] < 0 ||
	linkq[n] >= lb->kp.nvel || links[n] < 0 || links[n] >= 19) {
      return lbmi_fail(LBMI_ERR_ARGUMENT, "slip link %d of %d: (k, q, s) = "
		       "(%d, %d, %d) out of range", n, lb->nlink, linkk[n],
		       (int) linkq[n], (int) links[n]);
    }
  }
  for (int n = 0; n < 19; n++) {
    if (!(stab[n] >= 0.0 && stab[n] <= 1.0)) {
      return lbmi_fail(LBMI_ERR_ARGUMENT, "slip fractions must lie in [0, 1]");
    }
  }
  HIPCHECK(hipSetDevice(lb->device));
  HIPCHECK(hipStreamSynchronize(lb->stream));
  lbmi_slip_release(lb);
  n1 = (size_t) lb->nlink + 1;
  for (int k = 0; k < 3; k++) {
    lb->slip_host[k] = (int *) calloc(n1, sizeof(int));
    if (lb->slip_host[k] == NULL) {
      lbmi_slip_release(lb);
      return lbmi_fail(LBMI_ERR_HIP, "slip links");
    }
  }
  for (int n = 0; n < lb->nlink; n++) {
    lb->slip_host[0][n] = linkk[n];
    lb->slip_host[1][n] = linkq[n];
    lb->slip_host[2][n] = links[n];
  }
  if (hipMalloc((void **) &lb->slip_k_dev, n1*sizeof(int)) != hipSuccess ||
      hipMalloc((void **) &lb->slip_q_dev, n1) != hipSuccess ||
      hipMalloc((void **) &lb->slip_s_dev, n1) != hipSuccess ||
      hipMemcpy(lb->slip_k_dev, linkk, (n1 - 1)*sizeof(int), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(lb->slip_q_dev, linkq, n1 - 1, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(lb->slip_s_dev, links, n1 - 1, hipMemcpyHostToDevice) != hipSuccess) {
    lbmi_slip_release(lb);
    return lbmi_fail(LBMI_ERR_HIP, "slip links: device arrays");
  }
  memcpy(lb->slip_s, stab, 19*sizeof(double));
  lb->slip_active = 1;
  return 0;
}

int lbmi_wall_bbl_slip_arrays(lbmi_t * lb, int nlink, const int * linki,
			      const int * linkj, const int * linkp,
			      const int * linkk, const signed char * linkq,
			      const signed char * links,
			      const double stab[19], double * fnet) {
  lbmi_wall_tab_t tab;
  double na[LBMI_NVEL_MAX];
  double ma[LBMI_NVEL_MAX*LBMI_NVEL_MAX];
  if (lb == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  if (lb->f == NULL) return lbmi_fail(LBMI_ERR_STATE, "no distributions bound");
  if (lb->opts.mode != LBMI_MODE_EAGER &&
      !(lb->opts.mode == LBMI_MODE_FUSED_HALO && !lb->pending_prop)) {
    return lbmi_fail(LBMI_ERR_STATE, "walls need LBMI_MODE_EAGER or "
		     "LBMI_MODE_FUSED_HALO (before lb_propagation)");
  }
  if (nlink == 0) return 0;
  if (nlink < 0 || !linki || !linkj || !linkp || !linkk || !linkq || !links ||
      !stab || !fnet) {
    return lbmi_fail(LBMI_ERR_ARGUMENT, "lbmi_wall_bbl_slip_arrays: bad argument");
  }
  lb->halo_fresh = 0;                /* f is written: a halo swap after this one is real */
  HIPCHECK(hipSetDevice(lb->device));
  if (lb->wall_part == NULL || lb->wall_part_nblk < lbmi_k_wall_nblk(nlink)) {
    if (lb->wall_part) HIPCHECK(hipFree(lb->wall_part));
    lb->wall_part = NULL;
    lb->wall_part_nblk = lbmi_k_wall_nblk(nlink);
    HIPCHECK(hipMalloc((void **) &lb->wall_part,
		       sizeof(double)*3*(size_t) lb->wall_part_nblk));
  }
  memset(&tab, 0, sizeof(tab));
  tab.nvel = lb->kp.nvel;
  tab.ndist = lb->opts.ndist;
  tab.rho0 = lb->rho0;
  if (lbmi_k_model(lb->kp.nvel, &tab.cv[0][0], tab.wv, na, ma) != 0) {
    return lbmi_fail(LBMI_ERR_UNSUPPORTED, "model tables");
  }
  for (int n = 0; n < 19; n++) tab.slip[n] = stab[n];
  KCHECK(lbmi_k_wall_bbl_slip(&lb->kp, &tab, lb->f, nlink, linki, linkj, linkp,
			      linkk, (const int8_t *) linkq,
			      				      (const int8_t *) links, lb->wall_status,
				      lb->wall_part, fnet, lb->wall_err,
				      lb->stream));
  return lbmi_wall_after_launch(lb, linki, linkk, nlink);
}

int lbmi_wall_bbl(lbmi_t * lb) {
  double * fnet = NULL;
  if (lb == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
    if (lb->link_host[0] == NULL) return lbmi_fail(LBMI_ERR_STATE, "no links built");
  fnet = lb->wall_fnet_ext ? lb->wall_fnet_ext : lb->wall_fnet;
  if (lb->slip_active) {                                     /* wall.c:971 */
    return lbmi_wall_bbl_slip_arrays(lb, lb->nlink, lb->link_dev[0],
				     lb->link_dev[1], lb->link_dev[2],
				     lb->slip_k_dev,
				     (const signed char *) lb->slip_q_dev,
				     					     (const signed char *) lb->slip_s_dev,
					     lb->slip_s, fnet);
  }
  return lbmi_wall_bbl_arrays(lb, lb->nlink, lb->link_dev[0], lb->link_dev[1],
			      lb->link_dev[2], lb->link_dev[3], lb->wall_ubot,
			      lb->wall_utop, fnet);
}

int lbmi_wall_momentum(lbmi_t * lb, double fnet[3]) {
  if (lb == NULL || fnet == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  fnet[0] = fnet[1] = fnet[2] = 0.0;
  if (lb->wall_fnet == NULL) return 0;
  HIPCHECK(hipSetDevice(lb->device));
  /* accumulate to the host and zero the device total (wall.c:1306-1325) */
  HIPCHECK(hipMemcpyAsync(fnet, lb->wall_fnet, 3*sizeof(double),
			  hipMemcpyDeviceToHost, lb->stream));
  HIPCHECK(hipMemsetAsync(lb->wall_fnet, 0, 3*sizeof(double), lb->stream));
  HIPCHECK(hipStreamSynchronize(lb->stream));
  return 0;
}

/*****************************************************************************
 *
 *  The two-distribution step of free_energy symmetric_lb:
 *  phi_lb_to_field (phi_lb_coupler.c:39-112) and lb_collision_binary
 *  (collision.c:610-1027)
 *
 *****************************************************************************/

/* lb_collide with fe->use_stress_relaxation (collision.c:413-429,
 * FE_FORCE_METHOD_RELAXATION_SYMM) for the symmetric free energy and ONE
 * distribution. Always an in-place collision on the canonical state: a
 * deferred halo swap and propagation are materialised first. */

int lbmi_lb_collide_fe(lbmi_t * lb, const lbmi_hydro_t * hydro,
		       const lbmi_fe_symm_t * fe) {
  lbmi_hydro_dev_t h = lbmi_hydro_dev(hydro);
  int ifail;
  if (lb == NULL || fe == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  h.gstride = fe->nsite;
  if (lb->f == NULL) return lbmi_fail(LBMI_ERR_STATE, "no distributions bound");
  if (lb->opts.ndist != 1) return lbmi_fail(LBMI_ERR_STATE, "needs ndist = 1");
  if (lb->noise_state != NULL) {
    return lbmi_fail(LBMI_ERR_STATE, "fluctuations: lbmi_lb_collide only");
  }
  if (!fe->phi || !fe->grad || !fe->delsq) {
    return lbmi_fail(LBMI_ERR_ARGUMENT, "lbmi_lb_collide_fe: phi, grad and "
		     "delsq are required");
  }
  if (hydro == NULL) return 0;                       /* collision.c:149 */
  if (lbmi_deferred(lb) && lb->pending_halo && !lb->pending_prop) {
    return lbmi_fail(LBMI_ERR_STATE, "lb_collide after lb_halo without "
		     "lb_propagation");
  }
  if (lb->pending_prop && lb->pending_halo && !lbmi_inplace(lb) &&
      (lb->opts.mode == LBMI_MODE_FUSED_HALO || lb->opts.mode == LBMI_MODE_FUSED)) {
    /* the propagation of the step before folded into this collision: one pass
     * over f. FUSED on one rank: every direction wrapped by index;
     * FUSED_HALO: the halo swap has been done, pull from the array as it is;
     * FUSED on several ranks: the halo swap it has only noted, now, then the
     * same. */
    const int ranks = (lb->opts.cartsz > 1 || lb->have_comm);
    const int wrap = (lb->opts.mode == LBMI_MODE_FUSED && !ranks) ? lbmi_wrapmask(lb) : 0;
    ifail = lbmi_hydro_materialise(lb);      /* (rho, u an earlier collision owes) */
    if (ifail) return ifail;
    if (lb->blocked) {
      ifail = lbmi_unblock(lb);
      if (ifail) return ifail;
    }
    if (lb->opts.mode == LBMI_MODE_FUSED && ranks && !lb->halo_done) {
      ifail = lbmi_halo(lb, lb->f, lb->opts.halo_scheme);
      if (ifail) return ifail;
    }
    lb->hydro_stale = 0;
    lbmi_known_zero_drop(lb, h.rho);
    lbmi_known_zero_drop(lb, h.u);
    ifail = lbmi_time_begin(lb);
    if (ifail) return ifail;
    KCHECK(lbmi_k_propagate_collide_fe(&lb->kp, lb->f, lb->fprime, &h, fe->a, fe->b,
				       fe->kappa, fe->phi, fe->grad, fe->delsq, wrap,
				       lb->stream));
    ifail = lbmi_time_end(lb);
    if (ifail) return ifail;
    lbmi_swapf(lb);
    lb->pending_prop = 0;
    lb->pending_halo = 0;
    lb->halo_done = 0;
    lb->xsend_valid = 0; lb->halo_fresh = 0;
    return 0;
  }
  ifail = lbmi_lb_flush(lb);
  if (ifail) return ifail;
  lb->xsend_valid = 0; lb->halo_fresh = 0;
  KCHECK(lbmi_k_collide_fe(&lb->kp, lb->f, &h, fe->a, fe->b, fe->kappa,
			   fe->phi, fe->grad, fe->delsq, lb->stream));
  return 0;
}

int lbmi_lb_phi_to_field(lbmi_t * lb, double * phi) {
  if (lb == NULL || phi == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  if (lb->f == NULL) return lbmi_fail(LBMI_ERR_STATE, "no distributions bound");
  if (lb->opts.ndist != 2) return lbmi_fail(LBMI_ERR_STATE, "needs ndist = 2");
  HIPCHECK(hipSetDevice(lb->device));
  /* FUSED_HALO with the propagation pending: phi of the propagated state,
   * straight from the post-collision array (which has its halo) */
  KCHECK(lbmi_k_phi_from_g(&lb->kp, lb->f, phi, lb->pending_prop,
			   (lb->pending_prop && !lb->halo_done) ? 7 : 0,
			   lb->blocked, lb->stream));
  return 0;
}

int lbmi_lb_collide_binary(lbmi_t * lb, const lbmi_hydro_t * hydro,
			   const lbmi_fe_symm_t * fe) {
  lbmi_hydro_dev_t h = lbmi_hydro_dev(hydro);
  if (lb == NULL || fe == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  h.gstride = fe->nsite;
  if (lb->f == NULL) return lbmi_fail(LBMI_ERR_STATE, "no distributions bound");
  if (lb->opts.ndist != 2) return lbmi_fail(LBMI_ERR_STATE, "needs ndist = 2");
  if (!fe->phi || !fe->grad || !fe->delsq) {
    return lbmi_fail(LBMI_ERR_ARGUMENT, "lbmi_lb_collide_binary: phi, grad "
		     "and delsq are required");
  }
  /* lb_collide returns at once without a hydro object (collision.c:149);
   * the binary collision has no use for rho and status */
  if (hydro == NULL) return 0;
  h.rho = NULL;
  h.status = NULL;
  h.eta = NULL;                    /* fixed rates, collision.c:862-876 */
  if (lb->noise_state != NULL) {
    /* lb_collision_fluctuations at every site (collision.c:884-900) */
    h.noise = lb->noise_state;
    h.noise_stride = lb->noise_stride;
    h.noise_kt = lb->noise_kt;
    h.noise_ghosts = lb->noise_ghosts;
  }
  HIPCHECK(hipSetDevice(lb->device));
  if (lb->pending_halo && !lb->pending_prop) {
    return lbmi_fail(LBMI_ERR_STATE, "lb_collide after lb_halo without "
		     "lb_propagation");
  }
  /* (1/tau_2) = 2/(2M + 1), collision.c:1965-1968 */
  if (lb->pending_prop) {
    /* propagation(t) of both distributions inside collision(t+1): from the
     * halo where lb_halo has filled it (FUSED_HALO), by index wrap where that
     * is pending as well (FUSED on one GPU) */
    /* FUSED on one GPU (the halo swap pending as well: every pull wraps by
     * index): the deferred state in the blocked order, as for one
     * distribution. lay: 0 SoA -> SoA, 1 SoA -> blocked, 2 blocked -> blocked */
    int lay = (!lb->halo_done && lbmi_blocked_ok(lb)) ? (lb->blocked ? 2 : 1) : 0;
    if (lay == 0 && lb->blocked) {
      int ifail = lbmi_unblock(lb);
      if (ifail) return ifail;
    }
    if (lb->nt_store_mode >= 0) {
      lb->kp.nt_store = lb->nt_store_mode;
    }
    else {
      size_t bytes = 4*sizeof(double)*(size_t) lb->kp.nsite*(size_t) lb->kp.nvel;
      lb->kp.nt_store = (bytes > ((size_t) 256 << 20)) ? 1 : 0;
    }
    KCHECK(lbmi_k_collide_binary(&lb->kp, lb->f, lb->fprime, &h, fe->a, fe->b,
				 fe->kappa, 2.0/(1.0 + 2.0*fe->mobility),
				 fe->phi, fe->grad, fe->delsq,
				 lb->halo_done ? 0 : 7, lay, lb->stream));
    lb->pending_prop = 0;
    lb->pending_halo = 0;
    lb->halo_done = 0;
    lbmi_swapf(lb);
    lb->blocked = (lay != 0);
    return 0;
  }
  if (lb->blocked) {
    int ifail = lbmi_unblock(lb);
    if (ifail) return ifail;
  }
  KCHECK(lbmi_k_collide_binary(&lb->kp, lb->f, lb->f, &h, fe->a, fe->b,
			       fe->kappa, 2.0/(1.0 + 2.0*fe->mobility), fe->phi,
			       fe->grad, fe->delsq, 0, 0, lb->stream));
  return 0;
}

/*****************************************************************************
 *
 *  lbmi_lb_halo  (lb_halo, model.c:553-563)
 *
 *****************************************************************************/

int lbmi_lb_halo(lbmi_t * lb) {

  if (lb == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  if (lb->f == NULL) return lbmi_fail(LBMI_ERR_STATE, "no distributions bound");

  if (lbmi_deferred(lb)) {
    if (lb->early_prop) {
      lb->halo_seen = 1;               /* k_aa_odd has wrapped by index */
      return 0;
    }
    if (lb->pending_prop) {
      return lbmi_fail(LBMI_ERR_STATE, "lb_halo while a propagation is pending");
    }
    if (lb->opts.mode == LBMI_MODE_FUSED_HALO && lb->halo_fresh) {
      /* the kernel of the collision has computed the shell already and
       * nothing has written to f since (anything that does clears the flag) */
      lb->halo_done = 1;
    }
    else if (lb->opts.mode == LBMI_MODE_FUSED_HALO ||
	(lb->opts.ndist == 2 && (lb->opts.cartsz > 1 || lb->have_comm))) {
      /* eager: f gets its halo now; only the propagation will be deferred
       * (two distributions on slabs: FUSED is FUSED_HALO; on one GPU the
       * next collision wraps by index instead) */
      for (int n = 0; n < lb->opts.ndist; n++) {
	size_t off = (size_t) n*(size_t) lb->kp.nvel*(size_t) lb->kp.nsite;
	int ifail = lbmi_halo(lb, lb->f + off, lb->opts.halo_scheme);
	if (ifail) return ifail;
      }
      lb->halo_done = 1;
    }
    lb->pending_halo = 1;
    return 0;
  }

  /* every distribution (halo_swap_packed moves ndist*nvel values per site) */
  for (int n = 0; n < lb->opts.ndist; n++) {
    size_t off = (size_t) n*(size_t) lb->kp.nvel*(size_t) lb->kp.nsite;
    int ifail = lbmi_halo(lb, lb->f + off, lb->opts.halo_scheme);
    if (ifail) return ifail;
  }
  return 0;
}

/*****************************************************************************
 *
 *  lbmi_lb_propagation  (lb_propagation, propagation.c:43-98)
 *
 *****************************************************************************/

int lbmi_lb_propagation(lbmi_t * lb) {

  if (lb == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  if (lb->f == NULL) return lbmi_fail(LBMI_ERR_STATE, "no distributions bound");

  if (lbmi_deferred(lb)) {
    if (lb->early_prop) {
      /* done inside k_aa_odd: f is canonical now */
      lb->early_prop = 0;
      lb->halo_seen = 0;
      return 0;
    }
    if (lb->pending_prop) {
      return lbmi_fail(LBMI_ERR_STATE, "two propagations without a collision");
    }
    if (!lb->pending_halo) {
            /* A propagation that follows no halo swap (stale halos) cannot be
       * deferred faithfully: run it now */
      HIPCHECK(hipSetDevice(lb->device));
      {
	int ifail = lbmi_hydro_materialise(lb);
	if (ifail) return ifail;
      }
      if (lb->layout_swapped) {
	KCHECK(lbmi_k_aa_unswap(&lb->kp, lb->f, lb->stream));
	lb->layout_swapped = 0;
      }
      if (lb->blocked) {
	int ifail = lbmi_unblock(lb);
	if (ifail) return ifail;
      }
      for (int n = 0; n < lb->opts.ndist; n++) {
	size_t off = (size_t) n*(size_t) lb->kp.nvel*(size_t) lb->kp.nsite;
	KCHECK(lbmi_k_propagate(&lb->kp, lb->f + off, lb->fprime + off, lb->stream));
      }
      lbmi_swapf(lb);
      return 0;
    }
    lb->pending_prop = 1;
    return 0;
  }

  HIPCHECK(hipSetDevice(lb->device));
  for (int n = 0; n < lb->opts.ndist; n++) {
    size_t off = (size_t) n*(size_t) lb->kp.nvel*(size_t) lb->kp.nsite;
    KCHECK(lbmi_k_propagate(&lb->kp, lb->f + off, lb->fprime + off, lb->stream));
  }
  lbmi_swapf(lb);

  return 0;
}

/*****************************************************************************
 *
 *  lbmi_lb_flush
 *
 *  Materialise a pending halo swap and propagation so that f is exactly
 *  what the reference holds at this point of the time step.
 *
 *****************************************************************************/

int lbmi_lb_flush(lbmi_t * lb) {

  if (lb == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
    if (lb->f == NULL) return lbmi_fail(LBMI_ERR_STATE, "no distributions bound");
  HIPCHECK(hipSetDevice(lb->device));

  /* rho, u still owed (hydro_lazy) come from the post-collision state: now,
   * before a pending propagation turns it into the next one */
  {
    int ifail = lbmi_hydro_materialise(lb);
    if (ifail) return ifail;
  }

  if (lb->early_prop) {
    /* k_aa_odd has already applied P(t+1); the caller is between
     * lb_collide and lb_propagation, where f must be the post-collision
     * state: undo the propagation (exactly: it is a permutation) */
    KCHECK(lbmi_k_unpropagate_wrap(&lb->kp, lb->f, lb->fprime, 7, lb->stream));
    lbmi_swapf(lb);
    lb->early_prop = 0;
    if (lb->halo_seen) {
      int ifail = lbmi_halo(lb, lb->f, lb->opts.halo_scheme);
      if (ifail) return ifail;
      lb->halo_seen = 0;
    }
    return 0;
  }
  if (lb->layout_swapped) {
    KCHECK(lbmi_k_aa_unswap(&lb->kp, lb->f, lb->stream));
    lb->layout_swapped = 0;
  }
  if (lb->blocked) {
    int ifail = lbmi_unblock(lb);
    if (ifail) return ifail;
  }

  if (lb->pending_halo) {
    if (!lb->halo_done) {
      for (int n = 0; n < lb->opts.ndist; n++) {
	size_t off = (size_t) n*(size_t) lb->kp.nvel*(size_t) lb->kp.nsite;
	int ifail = lbmi_halo(lb, lb->f + off, lb->opts.halo_scheme);
	if (ifail) return ifail;
      }
    }
    lb->pending_halo = 0;
    lb->halo_done = 0;
  }
  if (lb->pending_prop) {
    for (int n = 0; n < lb->opts.ndist; n++) {
      size_t off = (size_t) n*(size_t) lb->kp.nvel*(size_t) lb->kp.nsite;
      KCHECK(lbmi_k_propagate(&lb->kp, lb->f + off, lb->fprime + off, lb->stream));
    }
    lbmi_swapf(lb);
    lb->pending_prop = 0;
  }

  return 0;
}

/* Change the execution mode of an existing handle, at any call point: what is
 * deferred is materialised first (lbmi_lb_flush), so the new mode starts
 * from the state the reference holds there. */

int lbmi_lb_mode_set(lbmi_t * lb, int mode) {
  if (lb == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  if (mode != LBMI_MODE_EAGER && mode != LBMI_MODE_FUSED &&
      mode != LBMI_MODE_INPLACE && mode != LBMI_MODE_FUSED_HALO) {
    return lbmi_fail(LBMI_ERR_ARGUMENT, "mode = %d", mode);
  }
  if (lb->opts.ndist == 2 && mode == LBMI_MODE_INPLACE) {
    return lbmi_fail(LBMI_ERR_UNSUPPORTED, "ndist = 2 needs LBMI_MODE_EAGER, LBMI_MODE_FUSED_HALO or LBMI_MODE_FUSED");
  }
  if ((mode == LBMI_MODE_FUSED || mode == LBMI_MODE_INPLACE) &&
      (lb->xdim == Z || lb->multi) && (lb->opts.cartsz > 1 || lb->have_comm)) {
    mode = LBMI_MODE_FUSED_HALO;     /* slabs along Z, or more than one
					direction decomposed: see lbmi_create */
  }
  if (mode == lb->opts.mode) return 0;
  if (lb->f != NULL) {
    int ifail = lbmi_lb_flush(lb);
    if (ifail) return ifail;
  }
  lbmi_run_graph_release(lb);
  lb->opts.mode = mode;
  return 0;
}

/* nsteps of the LB part of the main loop (ludwig.c:802-860) in one call:
 * lb_collide, lb_halo, lb_propagation with the same hydro object each step */

static int lbmi_one_step(lbmi_t * lb, const lbmi_hydro_t * hydro) {
  int ifail = lbmi_lb_collide(lb, hydro);
  if (ifail) return ifail;
  ifail = lbmi_lb_halo(lb);
  if (ifail) return ifail;
  return lbmi_lb_propagation(lb);
}

static void lbmi_run_graph_release(lbmi_t * lb) {
  if (lb->run_graph) hipGraphExecDestroy(lb->run_graph);
  lb->run_graph = NULL;
  memset(&lb->run_key, 0, sizeof(lb->run_key));
}

/* The steady state of FUSED on one GPU is ONE kernel per step, f -> fprime
 * and back: two steps leave the handle where it was, so a graph of two steps
 * can be launched any number of times. Captured through the ordinary step
 * functions (capture records the launches, the host-side state advances as
 * usual), keyed on everything the two launches carry by value or address. */

static int lbmi_run_graph_pairs(lbmi_t * lb, const lbmi_hydro_t * hydro,
				int npairs) {
  /* (what the captured launches carry: after the substitutions of
   * lbmi_lb_collide, and whether they store rho, u) */
  lbmi_hydro_dev_t h = lbmi_hydro_effective(lb, hydro);
  /* the legacy default stream cannot be captured: borrow the private one */
  hipStream_t user = lb->stream;
  hipStream_t cs = (user != NULL) ? user : lb->own_stream;

  if (lb->ev_graph == NULL) {
    HIPCHECK(hipEventCreateWithFlags(&lb->ev_graph, hipEventDisableTiming));
  }
  if (lb->run_graph == NULL || lb->run_key.f != lb->f ||
      lb->run_key.fprime != lb->fprime || lb->run_key.stream != cs ||
      lb->run_key.nt_store_mode != lb->nt_store_mode ||
      lb->run_key.hydro_lazy != lb->hydro_lazy ||
      memcmp(&lb->run_key.h, &h, sizeof(h)) != 0 ||
      memcmp(&lb->run_key.kp, &lb->kp, sizeof(lb->kp)) != 0) {
    hipGraph_t graph = NULL;
    int ifail = 0;
    lbmi_run_graph_release(lb);
    lb->run_key.f = lb->f;
    lb->run_key.fprime = lb->fprime;
    lb->run_key.h = h;
    lb->run_key.nt_store_mode = lb->nt_store_mode;
    lb->run_key.hydro_lazy = lb->hydro_lazy;
    lb->run_key.stream = cs;
    HIPCHECK(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
    lb->stream = cs;
    ifail = lbmi_one_step(lb, hydro);
    if (ifail == 0) ifail = lbmi_one_step(lb, hydro);
    lb->stream = user;
    if (hipStreamEndCapture(cs, &graph) != hipSuccess || ifail != 0 ||
	graph == NULL) {
      if (graph) hipGraphDestroy(graph);
      lbmi_run_graph_release(lb);
      return ifail ? ifail : lbmi_fail(LBMI_ERR_HIP, "lbmi_lb_run: stream capture");
    }
    /* (kp after the capture: the launches set kp.nt_store on the way) */
    lb->run_key.kp = lb->kp;
    if (hipGraphInstantiate(&lb->run_graph, graph, NULL, NULL, 0) != hipSuccess) {
      hipGraphDestroy(graph);
      lbmi_run_graph_release(lb);
      return lbmi_fail(LBMI_ERR_HIP, "lbmi_lb_run: hipGraphInstantiate");
    }
    HIPCHECK(hipGraphDestroy(graph));
    /* nothing has run yet: the capture advanced the host-side state by two
     * steps, which leaves it where it was; the launches below do the work */
  }

  if (cs != user) {
    HIPCHECK(hipEventRecord(lb->ev_graph, user));
    HIPCHECK(hipStreamWaitEvent(cs, lb->ev_graph, 0));
  }
  for (int n = 0; n < npairs; n++) {
    HIPCHECK(hipGraphLaunch(lb->run_graph, cs));
  }
  if (cs != user) {
    HIPCHECK(hipEventRecord(lb->ev_graph, cs));
    HIPCHECK(hipStreamWaitEvent(user, lb->ev_graph, 0));
  }
  return 0;
}

int lbmi_lb_run(lbmi_t * lb, const lbmi_hydro_t * hydro, int nsteps) {
  int n = 0;
  if (lb == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");

  if (lb->use_graph && lb->opts.mode == LBMI_MODE_FUSED &&
      lb->opts.cartsz == 1 && !lb->have_comm && lb->opts.ndist == 1 &&
      hydro != NULL && !lb->timing && nsteps >= 6) {
    /* two ordinary steps reach the steady state from wherever the handle is
     * (after a flush: collide in place, then SoA -> blocked) */
    for (; n < 2; n++) {
      int ifail = lbmi_one_step(lb, hydro);
      if (ifail) return ifail;
    }
    if (lb->pending_prop && lb->pending_halo && !lb->layout_swapped &&
	(lb->blocked != 0) == (lbmi_blocked_ok(lb) != 0)) {
      int npairs = (nsteps - n)/2;
      int ifail;
      HIPCHECK(hipSetDevice(lb->device));
      ifail = lbmi_run_graph_pairs(lb, hydro, npairs);
      if (ifail) return ifail;
      n += 2*npairs;
    }
  }

  for (; n < nsteps; n++) {
    int ifail = lbmi_one_step(lb, hydro);
    if (ifail) return ifail;
  }
  return 0;
}

int lbmi_lb_state(lbmi_t * lb, int state[3]) {
  if (lb == NULL || state == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  state[0] = ((lb->pending_halo && !lb->halo_done) || lb->halo_seen);
  state[1] = (lb->pending_prop || lb->early_prop);
  state[2] = lb->blocked ? 1 : (lb->layout_swapped ? 2 : 0);
  return 0;
}

/*****************************************************************************
 *
 *  lbmi_lb_memcpy_*  (lb_memcpy, model.c:228-266)
 *
 *****************************************************************************/

int lbmi_lb_memcpy_h2d(lbmi_t * lb, const double * f_host) {
  size_t sz;
  if (lb == NULL || f_host == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  if (lb->f == NULL) return lbmi_fail(LBMI_ERR_STATE, "no distributions bound");
  HIPCHECK(hipSetDevice(lb->device));
    sz = sizeof(double)*(size_t) lb->kp.nsite*(size_t) lb->kp.nvel
    *(size_t) lb->opts.ndist;
  {
    int ifail = lbmi_hydro_materialise(lb);     /* of the state that goes */
    if (ifail) return ifail;
  }
  lb->pending_halo = 0;
  lb->pending_prop = 0;
  lb->layout_swapped = 0;
  lb->early_prop = 0;
  lb->halo_seen = 0;
  lb->halo_done = 0;
  lb->blocked = 0;
  lb->xsend_valid = 0; lb->halo_fresh = 0;
  HIPCHECK(hipMemcpyAsync(lb->f, f_host, sz, hipMemcpyHostToDevice, lb->stream));
  HIPCHECK(hipStreamSynchronize(lb->stream));
  return 0;
}

/* The caller has rewritten the current f on the device by means of its own
 * (the reference's lb_memcpy host -> device after a Lees-Edwards reprojection
 * on the host, model_le.c:72-83; a foreign kernel between two steps): nothing
 * the handle derived from the old contents survives -- the planes a slab left
 * in its send buffers for messages already under way, rho and u still owed.
 * Needs the canonical state (lbmi_lb_flush first), as the writer did. */

int lbmi_lb_dirty(lbmi_t * lb) {
  if (lb == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  if (lb->f == NULL) return lbmi_fail(LBMI_ERR_STATE, "no distributions bound");
  if (lb->pending_prop || lb->blocked || lb->layout_swapped || lb->early_prop ||
      (lb->pending_halo && !lb->halo_done)) {
    return lbmi_fail(LBMI_ERR_STATE, "lbmi_lb_dirty: a deferred state cannot "
		     "have been rewritten by the caller (lbmi_lb_flush first)");
  }
  lb->xsend_valid = 0; lb->halo_fresh = 0;
  lb->hydro_stale = 0;
  return 0;
}

int lbmi_lb_memcpy_d2h(lbmi_t * lb, double * f_host) {
  size_t sz;
  int ifail;
  if (lb == NULL || f_host == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  ifail = lbmi_lb_flush(lb);
  if (ifail) return ifail;
  sz = sizeof(double)*(size_t) lb->kp.nsite*(size_t) lb->kp.nvel
    *(size_t) lb->opts.ndist;
  HIPCHECK(hipMemcpyAsync(f_host, lb->f, sz, hipMemcpyDeviceToHost, lb->stream));
  HIPCHECK(hipStreamSynchronize(lb->stream));
  return 0;
}

int lbmi_lb_moments(lbmi_t * lb, const char * status, double out[9]) {
  int ifail;
  if (lb == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  ifail = lbmi_lb_flush(lb);
  if (ifail) return ifail;
  return lbmi_moments(lb, lb->f, status, out);
}

/* Statistics of a scalar device field (nsite doubles) over the interior sites
 * that are fluid in `status` (or all, status == NULL): what cahn_stats_reduce
 * collects (cahn_hilliard_stats.c:123-215). out (HOST): volume, sum
 * (Kahan-compensated), sum of squares, minimum, maximum. Local to this rank. */

int lbmi_field_stats(lbmi_t * lb, const double * field, const char * status,
		     double out[5]) {
  double tmp[9];
  if (lb == NULL || field == NULL || out == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  HIPCHECK(hipSetDevice(lb->device));
  {
    int ifail = lbmi_hydro_touch(lb, field, NULL);
    if (ifail) return ifail;
  }
  KCHECK(lbmi_k_field_stats(&lb->kp, field, status, lb->mom_work, lb->mom_out,
			    lb->stream));
  HIPCHECK(hipMemcpyAsync(tmp, lb->mom_out, 9*sizeof(double),
			  hipMemcpyDeviceToHost, lb->stream));
  HIPCHECK(hipStreamSynchronize(lb->stream));
  out[0] = tmp[0];
  out[1] = tmp[5];
  out[2] = tmp[2];
  out[3] = tmp[3];
  out[4] = tmp[4];
  return 0;
}

/* lb_0th_moment of every interior site (first distribution), (ic, jc, kc)
 * order, to a HOST array of nlocal[X]*nlocal[Y]*nlocal[Z] doubles: what
 * stats_distribution_print sums up. fprime, dead once nothing is pending,
 * is the device scratch. */

int lbmi_lb_density(lbmi_t * lb, double * rho_host) {
  size_t n;
  int ifail;
  if (lb == NULL || rho_host == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  ifail = lbmi_lb_flush(lb);
  if (ifail) return ifail;
  n = (size_t) lb->kp.nlocal[X]*(size_t) lb->kp.nlocal[Y]*(size_t) lb->kp.nlocal[Z];
  KCHECK(lbmi_k_density(&lb->kp, lb->f, lb->fprime, lb->stream));
  HIPCHECK(hipMemcpyAsync(rho_host, lb->fprime, sizeof(double)*n,
			  hipMemcpyDeviceToHost, lb->stream));
  HIPCHECK(hipStreamSynchronize(lb->stream));
  return 0;
}

/*****************************************************************************
 *
 *  Rows "next": hydro housekeeping and the distribution record stream
 *
 *****************************************************************************/

int lbmi_hydro_field_set(lbmi_t * lb, double * field, int ncomp,
			 const double * values) {
  if (lb == NULL || field == NULL || values == NULL) {
    return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  }
    if (ncomp < 1 || ncomp > 3) {
    return lbmi_fail(LBMI_ERR_ARGUMENT, "ncomp = %d (1..3)", ncomp);
  }
  {
    int zero = 1;
    for (int n = 0; n < ncomp; n++) zero = zero && (values[n] == 0.0);
    /* zeros over zeros: nothing to do (hydro_f_zero every step on a force
     * field nobody has written to since) */
    if (zero && lbmi_known_zero(lb, field)) return 0;
    HIPCHECK(hipSetDevice(lb->device));
        if (lb->hydro_stale) {
      /* an array that is overwritten whole is no longer owed (hydro_u_zero
       * before the next collision, ludwig.c:791); a force that changes would
       * no longer be the one the collision used: settle first */
      if (field == lb->lazy_h.u) lb->lazy_h.u = NULL;
      if (field == lb->lazy_h.rho) lb->lazy_h.rho = NULL;
      if (lb->lazy_h.u == NULL && lb->lazy_h.rho == NULL) lb->hydro_stale = 0;
      if (field == lb->lazy_h.force) {
	int ifail = lbmi_hydro_materialise(lb);
	if (ifail) return ifail;
      }
    }
    KCHECK(lbmi_k_field_set(&lb->kp, ncomp, field, values, lb->stream));
    if (zero) lbmi_known_zero_add(lb, field);
    else lbmi_known_zero_drop(lb, field);
  }
  return 0;
}

/* field_halo (field.c) for an SoA field of nel components with a halo swap
 * of nswap layers (single rank; halo_swap_packed, halo_swap.c:709) */

int lbmi_field_halo_n(lbmi_t * lb, int nel, int nswap, double * data) {
  lbmi_halo_sel_t sel;
  if (lb == NULL || data == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  if (nel < 1 || nel > LBMI_NVEL_MAX) {
    return lbmi_fail(LBMI_ERR_ARGUMENT, "nel = %d (1..%d)", nel, LBMI_NVEL_MAX);
  }
  if (nswap < 1 || nswap > lb->kp.nhalo) {
    return lbmi_fail(LBMI_ERR_ARGUMENT, "nswap = %d (1..nhalo = %d)", nswap,
		     lb->kp.nhalo);
  }
  for (int d = 0; d < 3; d++) {
    if (nswap > lb->kp.nlocal[d]) {
      return lbmi_fail(LBMI_ERR_ARGUMENT, "nswap exceeds nlocal[%d]", d);
    }
  }
  HIPCHECK(hipSetDevice(lb->device));
  {
    int ifail = lbmi_hydro_touch(lb, data, NULL);    /* hydro_u_halo of a lazy u */
    if (ifail) return ifail;
  }
  memset(&sel, 0, sizeof(sel));
  for (int n = 0; n < nel; n++) {
    sel.lo[sel.nlo++] = (int8_t) n;
    sel.hi[sel.nhi++] = (int8_t) n;
  }
  for (int d = 0; d < 3; d++) {
    if (lbmi_dec(lb, d)) {
      /* slabs: the planes of every layer over the ring (device to device,
       * where the reference stages them through the host, halo_swap.c:762-881) */
      for (int layer = 0; layer < nswap; layer++) {
	int ifail = lbmi_x_exchange_along(lb, d, &sel, data, layer, lb->stream);
	if (ifail) return ifail;
      }
    }
    else {
      KCHECK(lbmi_k_halo_copy(&lb->kp, d, &sel, data, nswap, lb->stream));
    }
  }
  return 0;
}

/* field_grad_compute with grad_3d_7pt_fluid_d2 (gradient_3d_7pt_fluid.c) */

/* dst <- src at the interior sites of an SoA device field of ncomp
 * components: what puts the result of an out-of-place update
 * (lbmi_cahn_hilliard, phi -> phi_out) back where a caller that updates in
 * place keeps it, halo of dst untouched */

int lbmi_field_interior_copy(lbmi_t * lb, int ncomp, const double * src,
			     double * dst) {
  if (lb == NULL || src == NULL || dst == NULL || src == dst) {
    return lbmi_fail(LBMI_ERR_ARGUMENT, "lbmi_field_interior_copy: bad pointers");
  }
  if (ncomp < 1 || ncomp > LBMI_NVEL_MAX) {
    return lbmi_fail(LBMI_ERR_ARGUMENT, "ncomp = %d (1..%d)", ncomp, LBMI_NVEL_MAX);
  }
  HIPCHECK(hipSetDevice(lb->device));
  {
    int ifail = lbmi_hydro_touch(lb, src, dst);
    if (ifail) return ifail;
  }
  KCHECK(lbmi_k_interior_copy(&lb->kp, ncomp, src, dst, lb->stream));
  return 0;
}

int lbmi_fe_scheme_set(lbmi_t * lb, int grad_npt, int advection_order) {
  if (lb == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  if (grad_npt != 7 && grad_npt != 27) {
    return lbmi_fail(LBMI_ERR_ARGUMENT, "gradient stencil: 7 or 27 points");
  }
  if (advection_order < 1 || advection_order > 4) {
    /* the reference aborts likewise ("Unexpected advection scheme order",
     * advection.c:477); its order 5 needs a 3-layer halo and stays out */
    return lbmi_fail(LBMI_ERR_ARGUMENT, "advection scheme order: 1..4");
  }
  if (advection_order > 2 && lb->kp.nhalo < 2) {
    return lbmi_fail(LBMI_ERR_ARGUMENT, "advection order 3, 4 needs nhalo >= 2");
  }
  lb->grad_npt = grad_npt;
  lb->adv_order = advection_order;
  return 0;
}

static int lbmi_field_grad_npt(lbmi_t * lb, int npt, const double * phi,
			       double * grad, double * delsq) {
  if (lb == NULL || !phi || !grad || !delsq) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  /* the reference computes nextra = nhalo - 1 >= 0 layers beyond the
   * interior and reads one further: it needs the halo to exist */
  if (lb->kp.nhalo < 1) return lbmi_fail(LBMI_ERR_ARGUMENT, "nhalo < 1");
  HIPCHECK(hipSetDevice(lb->device));
  KCHECK(lbmi_k_grad(&lb->kp, npt, phi, grad, delsq, lb->stream));
  return 0;
}

int lbmi_field_grad_7pt(lbmi_t * lb, const double * phi, double * grad,
			double * delsq) {
  return lbmi_field_grad_npt(lb, 7, phi, grad, delsq);
}

int lbmi_field_grad_27pt(lbmi_t * lb, const double * phi, double * grad,
			 double * delsq) {
  return lbmi_field_grad_npt(lb, 27, phi, grad, delsq);
}

int lbmi_field_grad(lbmi_t * lb, const double * phi, double * grad,
		    double * delsq) {
  if (lb == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  return lbmi_field_grad_npt(lb, lb->grad_npt, phi, grad, delsq);
}

/* phi_force_calculation for the symmetric free energy and
 * FE_FORCE_METHOD_STRESS_DIVERGENCE without walls (phi_force.c:100-108) */

int lbmi_symmetric_force(lbmi_t * lb, double a, double b, double kappa,
			 const double * phi, const double * grad,
			 const double * delsq, double * force) {
  if (lb == NULL || !phi || !force) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  if ((grad == NULL) != (delsq == NULL)) {
    return lbmi_fail(LBMI_ERR_ARGUMENT, "grad and delsq: both or neither");
  }
  if (grad == NULL && lb->kp.nhalo < 2) {
    return lbmi_fail(LBMI_ERR_ARGUMENT, "force from phi needs nhalo >= 2");
  }
    HIPCHECK(hipSetDevice(lb->device));
  {
    int ifail = lbmi_hydro_touch(lb, NULL, force);
    if (ifail) return ifail;
  }
  KCHECK(lbmi_k_symm_force(&lb->kp, lb->grad_npt, a, b, kappa, phi, grad,
			   delsq, force, lb->stream));
  return 0;
}

/* phi_cahn_hilliard (phi_cahn_hilliard.c:195-284) for the symmetric free
 * energy: no noise, no walls, no Lees-Edwards planes */

int lbmi_cahn_hilliard(lbmi_t * lb, double a, double b, double kappa,
		       double mobility, const double * phi,
		       const double * delsq, const double * u,
		       double * phi_out) {
  if (lb == NULL || !phi || !u || !phi_out) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  if (phi_out == phi) {
    return lbmi_fail(LBMI_ERR_ARGUMENT, "lbmi_cahn_hilliard: phi_out must "
		     "not alias phi (every site reads its neighbours)");
  }
  if (delsq == NULL && lb->kp.nhalo < 2) {
    return lbmi_fail(LBMI_ERR_ARGUMENT, "mu from phi needs nhalo >= 2");
  }
    HIPCHECK(hipSetDevice(lb->device));
  {
    int ifail = lbmi_hydro_touch(lb, u, NULL);
    if (ifail) return ifail;
  }
  KCHECK(lbmi_k_cahn_hilliard(&lb->kp, lb->grad_npt, lb->adv_order, a, b,
			      kappa, mobility, phi, delsq, u, phi_out,
			      lb->stream));
  return 0;
}

/* phi_force_calculation + phi_cahn_hilliard in one pass over phi */

int lbmi_symmetric_step(lbmi_t * lb, double a, double b, double kappa,
			double mobility, const double * phi, const double * u,
			double * force, double * phi_out, int accumulate) {
  if (lb == NULL || !phi || !u || !force || !phi_out) {
    return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  }
  if (phi_out == phi) return lbmi_fail(LBMI_ERR_ARGUMENT, "phi_out aliases phi");
  if (lb->kp.nhalo < 2) return lbmi_fail(LBMI_ERR_ARGUMENT, "needs nhalo >= 2");
  HIPCHECK(hipSetDevice(lb->device));
  {
    int ifail = lbmi_hydro_touch(lb, u, force);
    if (ifail) return ifail;
  }
  KCHECK(lbmi_k_symm_fe_step(&lb->kp, lb->grad_npt, lb->adv_order, a, b,
			     kappa, mobility, phi, NULL, NULL, u, force,
			     phi_out, accumulate, 0, lb->stream));
  return 0;
}

/* The same on a single rank with periodic boundaries WITHOUT the halo swaps
 * of phi and u in front of it: the kernel wraps by index what field_halo
 * and hydro_u_halo would have supplied. */

int lbmi_symmetric_step_periodic(lbmi_t * lb, double a, double b,
				 double kappa, double mobility,
				 const double * phi, const double * u,
				 double * force, double * phi_out,
				 int accumulate) {
  if (lb == NULL || !phi || !u || !force || !phi_out) {
    return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  }
  if (phi_out == phi) return lbmi_fail(LBMI_ERR_ARGUMENT, "phi_out aliases phi");
  if (lb->kp.nhalo < 2) return lbmi_fail(LBMI_ERR_ARGUMENT, "needs nhalo >= 2");
  if (lb->opts.cartsz > 1 || lb->have_comm) {
    return lbmi_fail(LBMI_ERR_UNSUPPORTED, "lbmi_symmetric_step_periodic: one "
		     "rank (slabs: field halos + lbmi_symmetric_step)");
  }
  for (int d = 0; d < 3; d++) {
    if (lb->kp.nlocal[d] < lb->kp.nhalo) {
      return lbmi_fail(LBMI_ERR_ARGUMENT, "nlocal[%d] < nhalo", d);
    }
  }
  HIPCHECK(hipSetDevice(lb->device));
  {
    int ifail = lbmi_hydro_touch(lb, u, force);
    if (ifail) return ifail;
  }
  KCHECK(lbmi_k_symm_fe_step(&lb->kp, lb->grad_npt, lb->adv_order, a, b,
			     kappa, mobility, phi, NULL, NULL, u, force,
			     phi_out, accumulate, 1, lb->stream));
  return 0;
}

/* One complete step of the binary fluid with the finite-difference order
 * parameter (ludwig.c:537-860 for free_energy symmetric): thermodynamic force,
 * Cahn-Hilliard update, lb_collide, lb_halo, lb_propagation. In the steady
 * state of LBMI_MODE_FUSED on one rank that is ONE launch (k_symm_lb_step);
 * anywhere else the same results come from the separate calls. */

static int lbmi_symmetric_lb_impl(lbmi_t * lb, const lbmi_hydro_t * hydro,
				  const double * u_prev, double a, double b,
				  double kappa, double mobility, const double * phi,
				  double * phi_out, int whole_step);

int lbmi_symmetric_lb_step(lbmi_t * lb, const lbmi_hydro_t * hydro,
			   const double * u_prev, double a, double b,
			   double kappa, double mobility, const double * phi,
			   double * phi_out) {
  return lbmi_symmetric_lb_impl(lb, hydro, u_prev, a, b, kappa, mobility, phi,
				phi_out, 1);
}

int lbmi_symmetric_lb_collide(lbmi_t * lb, const lbmi_hydro_t * hydro,
			      const double * u_prev, double a, double b,
			      double kappa, double mobility, const double * phi,
			      double * phi_out) {
  return lbmi_symmetric_lb_impl(lb, hydro, u_prev, a, b, kappa, mobility, phi,
				phi_out, 0);
}

static int lbmi_symmetric_lb_impl(lbmi_t * lb, const lbmi_hydro_t * hydro,
				  const double * u_prev, double a, double b,
				  double kappa, double mobility, const double * phi,
				  double * phi_out, int whole_step) {
  lbmi_hydro_t hy;
  int fused;
  if (lb == NULL || hydro == NULL || !u_prev || !phi || !phi_out) {
    return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  }
  if (lb->f == NULL) return lbmi_fail(LBMI_ERR_STATE, "no distributions bound");
  if (phi_out == phi) return lbmi_fail(LBMI_ERR_ARGUMENT, "phi_out aliases phi");
  if (hydro->u == NULL || hydro->u == u_prev) {
    return lbmi_fail(LBMI_ERR_ARGUMENT, "lbmi_symmetric_lb_step: hydro->u must "
		     "be an array other than u_prev (the neighbours' old "
		     "velocities are read while the new ones are written)");
  }
  if (lb->opts.ndist != 1) return lbmi_fail(LBMI_ERR_STATE, "needs ndist = 1");
  if (lb->kp.nhalo < 2) return lbmi_fail(LBMI_ERR_ARGUMENT, "needs nhalo >= 2");
  if (lb->opts.cartsz > 1 || lb->have_comm) {
    return lbmi_fail(LBMI_ERR_UNSUPPORTED, "lbmi_symmetric_lb_step: one rank "
		     "(slabs: field halos, lbmi_symmetric_step, the LB calls)");
  }
  for (int d = 0; d < 3; d++) {
    if (lb->kp.nlocal[d] < lb->kp.nhalo) {
      return lbmi_fail(LBMI_ERR_ARGUMENT, "nlocal[%d] < nhalo", d);
    }
  }
  if (hydro->force != NULL && !lbmi_known_zero(lb, hydro->force)) {
    return lbmi_fail(LBMI_ERR_ARGUMENT, "lbmi_symmetric_lb_step: hydro->force "
		     "must be NULL or known to hold zeros (lbmi_hydro_field_set): "
		     "the thermodynamic force goes from the registers of the "
		     "thread that evaluates it into its collision; with other "
		     "contributions use lbmi_symmetric_step + lbmi_lb_collide");
  }
  HIPCHECK(hipSetDevice(lb->device));
  {
    /* u_prev is read: formed first if a lazy collision still owes it */
    int ifail = lbmi_hydro_touch(lb, u_prev, NULL);
    if (ifail) return ifail;
  }

  hy = *hydro;
  hy.force = NULL;

  fused = (lb->opts.mode == LBMI_MODE_FUSED && lb->pending_prop &&
	   lb->pending_halo && !lb->layout_swapped && lb->noise_state == NULL &&
	   lb->grad_npt == 7 &&
	   (lb->kp.scheme == LBMI_RELAXATION_M10 ||
	    lb->kp.scheme == LBMI_RELAXATION_BGK) &&
	   lb->kp.nlocal[X] >= 4 && lb->kp.nlocal[Y] >= 4 && lb->kp.nlocal[Z] >= 4 &&
	   (hydro->nsite == 0 || hydro->nsite == lb->kp.nsite));

  if (fused) {
    lbmi_hydro_dev_t h = lbmi_hydro_dev(&hy);
    int lay = lbmi_blocked_ok(lb) ? (lb->blocked ? 2 : 1) : 0;
    int owed = 0;
    int ifail;
    if (lay == 0 && lb->blocked) {
      ifail = lbmi_unblock(lb);
      if (ifail) return ifail;
    }
    if (lb->nt_store_mode >= 0) {
      lb->kp.nt_store = lb->nt_store_mode;
    }
    else {
      size_t bytes = 2*sizeof(double)*(size_t) lb->kp.nsite*(size_t) lb->kp.nvel;
      lb->kp.nt_store = (bytes > ((size_t) 256 << 20)) ? 1 : 0;
    }
    lb->hydro_stale = 0;              /* superseded by this collision */
    lbmi_known_zero_drop(lb, h.u);
    lbmi_known_zero_drop(lb, phi_out);
    if (lb->hydro_lazy && h.rho != NULL) {
      /* u is stored (the next step reads it); rho is owed until asked for */
      lb->lazy_h = h;
      lb->lazy_h.u = NULL;
      lb->lazy_h.force = NULL;
      for (int ia = 0; ia < 3; ia++) lb->lazy_fbody[ia] = lb->kp.fbody[ia];
      h.rho = NULL;
      owed = 1;
    }
    else {
      lbmi_known_zero_drop(lb, h.rho);
    }
    ifail = lbmi_time_begin(lb);
    if (ifail) return ifail;
    KCHECK(lbmi_k_symm_lb_step(&lb->kp, lb->f, lb->fprime, &h, a, b, kappa,
			       mobility, lb->adv_order, phi, u_prev, phi_out,
			       lay, lb->stream));
    ifail = lbmi_time_end(lb);
    if (ifail) return ifail;
    lb->blocked = (lay != 0);
    lbmi_swapf(lb);
    lb->hydro_stale = owed;
    if (!whole_step) {
      /* where lbmi_lb_collide leaves the handle: lb_halo and lb_propagation
       * of this step are the caller's to call */
      lb->pending_prop = 0;
      lb->pending_halo = 0;
      lb->halo_done = 0;
    }
    /* (the whole step: lb_halo and lb_propagation pending again, as they were) */
    return 0;
  }

  /* Anywhere else (first step after a flush, another mode, 27-point
   * gradients, TRT, fluctuations ...): the separate calls. The force
   * needs an array: one of the handle's own. */
  {
    int ifail;
    const int lazy = lb->hydro_lazy;
    if (lb->fe_force == NULL) {
      HIPCHECK(hipMalloc((void **) &lb->fe_force,
			 3*sizeof(double)*(size_t) lb->kp.nsite));
    }
    KCHECK(lbmi_k_symm_fe_step(&lb->kp, lb->grad_npt, lb->adv_order, a, b,
			       kappa, mobility, phi, NULL, NULL, u_prev,
			       lb->fe_force, phi_out, 0, 1, lb->stream));
    hy.force = lb->fe_force;
    hy.nsite = 0;
    lb->hydro_lazy = lazy ? 2 : 0;    /* the next step reads this u */
    ifail = whole_step ? lbmi_one_step(lb, &hy) : lbmi_lb_collide(lb, &hy);
    lb->hydro_lazy = lazy;
    return ifail;
  }
}

int lbmi_symmetric_step_grad(lbmi_t * lb, double a, double b, double kappa,
			     double mobility, const double * phi,
			     const double * grad, const double * delsq,
			     const double * u, double * force,
			     double * phi_out, int accumulate) {
  if (lb == NULL || !phi || !grad || !delsq || !u || !force || !phi_out) {
    return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  }
  if (phi_out == phi) return lbmi_fail(LBMI_ERR_ARGUMENT, "phi_out aliases phi");
  if (lb->kp.nhalo < 2) return lbmi_fail(LBMI_ERR_ARGUMENT, "needs nhalo >= 2");
  HIPCHECK(hipSetDevice(lb->device));
  {
    int ifail = lbmi_hydro_touch(lb, u, force);
    if (ifail) return ifail;
  }
  KCHECK(lbmi_k_symm_fe_step(&lb->kp, 0, lb->adv_order, a, b, kappa, mobility,
			     phi, grad, delsq, u, force, phi_out, accumulate,
			     0, lb->stream));
  return 0;
}

int lbmi_lb_records_pack(lbmi_t * lb, double * records) {
  int ifail;
  if (lb == NULL || records == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  ifail = lbmi_lb_flush(lb);
  if (ifail) return ifail;
  KCHECK(lbmi_k_records(&lb->kp, lb->opts.ndist, lb->f, records, 1, lb->stream));
  return 0;
}

int lbmi_lb_records_unpack(lbmi_t * lb, const double * records) {
  if (lb == NULL || records == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  if (lb->f == NULL) return lbmi_fail(LBMI_ERR_STATE, "no distributions bound");
  HIPCHECK(hipSetDevice(lb->device));
    {
    int ifail = lbmi_hydro_materialise(lb);     /* of the state that goes */
    if (ifail) return ifail;
  }
  /* reading a checkpoint replaces the state: nothing stays pending */
  lb->pending_halo = 0;
  lb->pending_prop = 0;
  lb->early_prop = 0;
  lb->halo_seen = 0;
  lb->halo_done = 0;
  lb->blocked = 0;
  lb->xsend_valid = 0; lb->halo_fresh = 0;
  if (lb->layout_swapped) {
    lb->layout_swapped = 0;
  }
  KCHECK(lbmi_k_records(&lb->kp, lb->opts.ndist, lb->f, (double *) records, 0,
			lb->stream));
  return 0;
}

/*****************************************************************************
 *
 *  Distribution files (lb_io_write / lb_io_read, model.c:1568-1649, in the
 *  reference's MPI-IO mode with an i/o grid 1_1_1)
 *
 *****************************************************************************/

static int lbmi_io_filename_file(const char * dir, const char * stub, int timestep,
				 int fmt, const lbmi_io_file_t * file,
				 char * buf, size_t bufsz) {
  int n;
  if (file == NULL || file->nfile <= 1) {
    return lbmi_io_filename_fmt(dir, stub, timestep, fmt, buf, bufsz);
  }
  if (dir == NULL || stub == NULL || buf == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  /* io_subfile_name (io_subfile.c): 1 + index of nfile */
  n = snprintf(buf, bufsz, "%s/%s-%9.9d.%3.3d-%3.3d", dir, stub, timestep,
	       1 + file->index, file->nfile);
  if (n < 0 || (size_t) n >= bufsz) return lbmi_fail(LBMI_ERR_ARGUMENT, "file name too long");
  return 0;
}

int lbmi_io_filename_fmt(const char * dir, const char * stub, int timestep,
			 int fmt, char * buf, size_t bufsz) {
  int n;
  if (dir == NULL || stub == NULL || buf == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  if (fmt & LBMI_IO_SINGLE) {
    /* lb_io_write, model.c:1583-1587: "dist-%8.8d", then io_write_data_s
     * (io_harness.c): "%s.%3.3d-%3.3d" with 1, 1 */
    n = snprintf(buf, bufsz, "%s/%s-%8.8d.%3.3d-%3.3d", dir, stub, timestep, 1, 1);
  }
  else {
    /* io_subfile_name: "%s-%9.9d.%3.3d-%3.3d", file index 1 of 1 */
    n = snprintf(buf, bufsz, "%s/%s-%9.9d.%3.3d-%3.3d", dir, stub, timestep, 1, 1);
  }
  if (n < 0 || (size_t) n >= bufsz) return lbmi_fail(LBMI_ERR_ARGUMENT, "file name too long");
  return 0;
}

int lbmi_io_filename(const char * dir, const char * stub, int timestep,
		     char * buf, size_t bufsz) {
  return lbmi_io_filename_fmt(dir, stub, timestep, 0, buf, bufsz);
}

/* Text records (io_options_t::iorformat == IO_RECORD_ASCII, the input key
 * distribution_io_format ascii): lb_write_buf_ascii (model.c:1438-1462) puts
 * the record of a site on nvel lines, line p holding the ndist values
 * f(n, p) as " %22.15e" (23 characters each) and a newline */

enum {LBMI_ASCII_DATUM = 23};

static size_t lbmi_ascii_record(int nvel, int ndist) {
  return (size_t) nvel*((size_t) ndist*LBMI_ASCII_DATUM + 1);
}

int lbmi_io_metadata_write(const char * dir, const char * stub, int nel,
			   const int ntotal[3]) {
  return lbmi_io_metadata_write_fmt(dir, stub, nel, 1, ntotal, 0);
}

int lbmi_io_metadata_write_fmt(const char * dir, const char * stub, int nvel,
			       int ndist, const int ntotal[3], int fmt) {
  return lbmi_io_metadata_write_file(dir, stub, nvel, ndist, ntotal, fmt, NULL);
}

int lbmi_io_metadata_write_file(const char * dir, const char * stub, int nvel,
				int ndist, const int ntotal[3], int fmt,
				const lbmi_io_file_t * file) {
  const int nel = nvel*ndist;
  const int ascii = (fmt & LBMI_IO_ASCII);
  const int single = (fmt & LBMI_IO_SINGLE);
  /* io_subfile_create (io_subfile.c:49-91) for an i/o grid {nfile, 1, 1} */
  lbmi_io_file_t one = {1, 0, 0, 0, {1, 1, 1}};
  char fn[1024];
  FILE * fp = NULL;
  int n;
  if (dir == NULL || stub == NULL || ntotal == NULL || nvel < 1 || ndist < 1) {
    return lbmi_fail(LBMI_ERR_ARGUMENT, "lbmi_io_metadata_write: bad argument");
  }
  one.file_nx = ntotal[0];
  if (file == NULL) file = &one;
  if (file->nfile < 1 || file->index < 0 || file->index >= file->nfile ||
      file->file_x0 < 0 || file->file_nx < 1 ||
      file->file_x0 + file->file_nx > ntotal[0] ||
      (file->nfile == 1 && (file->file_x0 != 0 || file->file_nx != ntotal[0]))) {
    return lbmi_fail(LBMI_ERR_ARGUMENT, "file %d of %d, planes %d + %d of %d",
		     file->index, file->nfile, file->file_x0, file->file_nx, ntotal[0]);
  }
  if (file->nfile > 1 && single) {
    /* (several old-style files are IO_MODE_MULTIPLE, io_write_data_p) */
    return lbmi_fail(LBMI_ERR_UNSUPPORTED, "the single mode has one file");
  }
  n = snprintf(fn, sizeof(fn), "%s/%s-metadata.%3.3d-%3.3d", dir, stub,
	       1 + file->index, file->nfile);
  if (n < 0 || (size_t) n >= sizeof(fn)) return lbmi_fail(LBMI_ERR_ARGUMENT, "file name too long");
  fp = fopen(fn, "w");
  if (fp == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "%s: %s", fn, strerror(errno));
  /* The objects and keys of io_metadata_to_json (io_metadata.c): cs_to_json,
   * io_options_to_json (defaults of io_options_with_mode(IO_MODE_MPIIO)),
   * io_element_to_json, io_subfile_to_json; laid out as cJSON_Print does */
  fprintf(fp, "{\n");
  fprintf(fp, "\t\"coords\":\t{\n");
  fprintf(fp, "\t\t\"options\":\t{\n");
  fprintf(fp, "\t\t\t\"System size (total)\":\t[%d, %d, %d],\n", ntotal[0], ntotal[1], ntotal[2]);
  fprintf(fp, "\t\t\t\"Periodic boundaries\":\t[%d, %d, %d],\n",
	  file->periodic[0], file->periodic[1], file->periodic[2]);
  fprintf(fp, "\t\t\t\"Left-end limit Lmin\":\t[0.5, 0.5, 0.5]\n");
  fprintf(fp, "\t\t},\n");
  fprintf(fp, "\t\t\"lees_edwards\":\t{\n");
  fprintf(fp, "\t\t\t\"Number of planes\":\t0\n");
  fprintf(fp, "\t\t}\n");
  fprintf(fp, "\t},\n");
  fprintf(fp, "\t\"io_options\":\t{\n");
  /* io_options_with_mode(IO_MODE_MPIIO), or io_options_default() = the
   * single mode (io_options.c:27-45, 198-222) */
  fprintf(fp, "\t\t\"Mode\":\t\"%s\",\n", single ? "single" : "mpiio");
  fprintf(fp, "\t\t\"Record format\":\t\"%s\",\n", ascii ? "ascii" : "binary");
  fprintf(fp, "\t\t\"Metadata version\":\t%d,\n", single ? 1 : 3);
  fprintf(fp, "\t\t\"Report\":\t%s,\n", single ? "false" : "true");
  fprintf(fp, "\t\t\"Asynchronous\":\tfalse,\n");
  fprintf(fp, "\t\t\"Compression level\":\t0,\n");
  fprintf(fp, "\t\t\"I/O grid\":\t[%d, 1, 1]\n", file->nfile);
  fprintf(fp, "\t},\n");
  fprintf(fp, "\t\"io_element\":\t{\n");
  if (ascii) {
    /* lb_data_create: the element of a text record is its characters
     * (model.c:118-129: MPI_CHAR, nvel*(ndist*23 + 1) of them) */
    fprintf(fp, "\t\t\"MPI_Datatype\":\t\"MPI_CHAR\",\n");
    fprintf(fp, "\t\t\"Size (bytes)\":\t1,\n");
    fprintf(fp, "\t\t\"Count\":\t%d,\n", (int) lbmi_ascii_record(nvel, ndist));
  }
  else {
    fprintf(fp, "\t\t\"MPI_Datatype\":\t\"MPI_DOUBLE\",\n");
    fprintf(fp, "\t\t\"Size (bytes)\":\t8,\n");
    fprintf(fp, "\t\t\"Count\":\t%d,\n", nel);
  }
  fprintf(fp, "\t\t\"Endianness\":\t\"LITTLE_ENDIAN\"\n");
  fprintf(fp, "\t},\n");
  fprintf(fp, "\t\"io_subfile\":\t{\n");
  fprintf(fp, "\t\t\"Number of files\":\t%d,\n", file->nfile);
  fprintf(fp, "\t\t\"File index\":\t%d,\n", file->index);
  fprintf(fp, "\t\t\"Topology\":\t[%d, 1, 1],\n", file->nfile);
  fprintf(fp, "\t\t\"Coordinate\":\t[%d, 0, 0],\n", file->index);
  fprintf(fp, "\t\t\"Data ndims\":\t3,\n");
  fprintf(fp, "\t\t\"File size (sites)\":\t[%d, %d, %d],\n", file->file_nx, ntotal[1], ntotal[2]);
  fprintf(fp, "\t\t\"File offset (sites)\":\t[%d, 0, 0]\n", file->file_x0);
  fprintf(fp, "\t}\n");
  fprintf(fp, "}");
  if (fclose(fp) != 0) return lbmi_fail(LBMI_ERR_ARGUMENT, "%s: %s", fn, strerror(errno));
  return 0;
}

/* The text file the old-style i/o keeps beside its data (io_write_metadata_file,
 * io_harness.c:369-466): a stub and one line per rank of the i/o group in
 * rank order. Slabs along cartdim: rank r has Cartesian coordinate r in that
 * direction, nslab[r] planes. */

int lbmi_io_single_metadata_write(const char * dir, const char * stub, int nvel,
				  int ndist, const int ntotal[3], int cartdim,
				  int cartsz, const int * nslab) {
  char fn[1024];
  FILE * fp = NULL;
  int n, off = 0;
  if (dir == NULL || stub == NULL || ntotal == NULL || nvel < 1 || ndist < 1 ||
      cartdim < X || cartdim > Z || cartsz < 1 || (cartsz > 1 && nslab == NULL)) {
    return lbmi_fail(LBMI_ERR_ARGUMENT, "lbmi_io_single_metadata_write: bad argument");
  }
  for (int r = 0; r < cartsz && cartsz > 1; r++) off += nslab[r];
  if (cartsz > 1 && off != ntotal[cartdim]) {
    return lbmi_fail(LBMI_ERR_ARGUMENT, "slabs of %d planes in all, %d sites",
		     off, ntotal[cartdim]);
  }
  n = snprintf(fn, sizeof(fn), "%s/%s.%3.3d-%3.3d.meta", dir, stub, 1, 1);
  if (n < 0 || (size_t) n >= sizeof(fn)) return lbmi_fail(LBMI_ERR_ARGUMENT, "file name too long");
  fp = fopen(fn, "w");
  if (fp == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "%s: %s", fn, strerror(errno));
  {
    int sz[3] = {1, 1, 1};
    sz[cartdim] = cartsz;
    fprintf(fp, "Metadata for file set prefix:    %s\n", stub);
    /* lb_io_info_set, model.c:457 */
    fprintf(fp, "Data description:                %1d x Distribution: d3q%d\n", ndist, nvel);
    fprintf(fp, "Data size per site (bytes):      %d\n", (int) sizeof(double)*nvel*ndist);
    fprintf(fp, "is_bigendian():                  %d\n", 0);
    fprintf(fp, "Number of processors:            %d\n", cartsz);
    fprintf(fp, "Cartesian communicator topology: %d %d %d\n", sz[X], sz[Y], sz[Z]);
    fprintf(fp, "Total system size:               %d %d %d\n", ntotal[X], ntotal[Y], ntotal[Z]);
    fprintf(fp, "Lees-Edwards planes:             %d\n", 0);
    fprintf(fp, "Lees-Edwards plane speed         %16.14f\n", 0.0);
    fprintf(fp, "Number of I/O groups (files):    %d\n", 1);
    fprintf(fp, "I/O communicator topology:       %d %d %d\n", 1, 1, 1);
    fprintf(fp, "Write order:\n");
  }
  off = 0;
  for (int r = 0; r < cartsz; r++) {
    int coords[3] = {0, 0, 0};
    int nlocal[3] = {ntotal[X], ntotal[Y], ntotal[Z]};
    int noff[3] = {0, 0, 0};
    coords[cartdim] = r;
    if (cartsz > 1) nlocal[cartdim] = nslab[r];
    noff[cartdim] = off;
    off += nlocal[cartdim];
    fprintf(fp, "%3d %3d %3d %3d %d %d %d %d %d %d\n", r, coords[X], coords[Y], coords[Z],
	    nlocal[X], nlocal[Y], nlocal[Z], noff[X], noff[Y], noff[Z]);
  }
  if (fclose(fp) != 0) return lbmi_fail(LBMI_ERR_ARGUMENT, "%s: %s", fn, strerror(errno));
  return 0;
}

/* How cs_init cuts ntotal sites into sz parts (coords.c:577-592): the
 * quotient each, and one more for the parts 1 + n*(sz/nremainder) */

static void lbmi_cs_parts(int ntotal, int sz, int * nslab) {
  const int nrem = ntotal % sz;
  for (int r = 0; r < sz; r++) nslab[r] = ntotal/sz;
  for (int n = 0; n < nrem; n++) nslab[1 + n*(sz/nrem)] += 1;
}

/* records <-> file through a pinned staging buffer; dev = fprime */

#define LBMI_IO_CHUNK ((size_t) 32*1024*1024)

/* binary records (ndist*nvel doubles per site, [n][p]) <-> text records */

static void lbmi_ascii_from_records(const double * rec, size_t nsites, int nvel,
				    int ndist, char * text) {
  const size_t len = lbmi_ascii_record(nvel, ndist);
  for (size_t k = 0; k < nsites; k++) {
    const double * r = rec + k*(size_t) (nvel*ndist);
    char * t = text + k*len;
    for (int p = 0; p < nvel; p++) {
      char * line = t + (size_t) p*((size_t) ndist*LBMI_ASCII_DATUM + 1);
      for (int n = 0; n < ndist; n++) {
	char tmp[64];
	/* (a value that does not fit the width is cut, as snprintf(tmp,
	 * nbyte + 1, ...) cuts it there) */
	snprintf(tmp, LBMI_ASCII_DATUM + 1, " %22.15e", r[n*nvel + p]);
	memcpy(line + n*LBMI_ASCII_DATUM, tmp, LBMI_ASCII_DATUM);
      }
      line[ndist*LBMI_ASCII_DATUM] = '\n';
    }
  }
}

static int lbmi_records_from_ascii(const char * text, size_t nsites, int nvel,
				   int ndist, double * rec) {
  const size_t len = lbmi_ascii_record(nvel, ndist);
  for (size_t k = 0; k < nsites; k++) {
    double * r = rec + k*(size_t) (nvel*ndist);
    const char * t = text + k*len;
    for (int p = 0; p < nvel; p++) {
      const char * line = t + (size_t) p*((size_t) ndist*LBMI_ASCII_DATUM + 1);
      for (int n = 0; n < ndist; n++) {
	char tmp[LBMI_ASCII_DATUM + 1];
	char * end = NULL;
	memcpy(tmp, line + n*LBMI_ASCII_DATUM, LBMI_ASCII_DATUM);
	tmp[LBMI_ASCII_DATUM] = '\0';
	r[n*nvel + p] = strtod(tmp, &end);             /* sscanf "%le" there */
	if (end == tmp) return -1;
      }
    }
  }
  return 0;
}

/* nbytes, offset: of the BINARY record stream; with text records the file
 * holds lbmi_ascii_record() characters where that stream has nel*8 bytes */

static int lbmi_io_transfer(lbmi_t * lb, const char * fn, int writing,
			    double * dev, size_t nbytes, off_t offset) {
  void * stage = NULL;
  char * text = NULL;
  const int ascii = (lb->io_ascii & LBMI_IO_ASCII);
  const size_t recb = sizeof(double)*(size_t) lb->kp.nvel*(size_t) lb->opts.ndist;
  const size_t rect = lbmi_ascii_record(lb->kp.nvel, lb->opts.ndist);
  const size_t chunk = (LBMI_IO_CHUNK/recb)*recb;       /* whole records */
  size_t done = 0;
  int fd = -1;
  int ifail = 0;

  fd = writing ? open(fn, O_WRONLY | O_CREAT, 0644) : open(fn, O_RDONLY);
  if (fd < 0) return lbmi_fail(LBMI_ERR_ARGUMENT, "%s: %s", fn, strerror(errno));
  if (hipHostMalloc(&stage, LBMI_IO_CHUNK, hipHostMallocDefault) != hipSuccess) {
    close(fd);
    return lbmi_fail(LBMI_ERR_HIP, "hipHostMalloc of the i/o staging buffer");
  }
  if (ascii) {
    text = (char *) malloc((chunk/recb)*rect);
    if (text == NULL) {
      hipHostFree(stage);
      close(fd);
      return lbmi_fail(LBMI_ERR_ARGUMENT, "malloc of the text buffer");
    }
  }
  while (done < nbytes && ifail == 0) {
    size_t n = nbytes - done;
    size_t io = 0;
    /* what goes to / comes from the file for this chunk, and where */
    char * fbuf = (char *) stage;
    size_t fn_bytes;
    off_t foff;
    if (n > chunk) n = chunk;
    fn_bytes = ascii ? (n/recb)*rect : n;
    foff = ascii ? (off_t) ((((size_t) offset + done)/recb)*rect)
      : offset + (off_t) done;
    if (ascii) fbuf = text;
    if (writing) {
      if (hipMemcpyAsync(stage, (char *) dev + done, n, hipMemcpyDeviceToHost,
			 lb->stream) != hipSuccess ||
	  hipStreamSynchronize(lb->stream) != hipSuccess) {
	ifail = lbmi_fail(LBMI_ERR_HIP, "device to host copy of records");
	break;
      }
      if (ascii) {
	lbmi_ascii_from_records((const double *) stage, n/recb, lb->kp.nvel,
				lb->opts.ndist, text);
      }
    }
    while (io < fn_bytes) {
      ssize_t r = writing
	? pwrite(fd, fbuf + io, fn_bytes - io, foff + (off_t) io)
	: pread(fd, fbuf + io, fn_bytes - io, foff + (off_t) io);
      if (r <= 0) {
	ifail = lbmi_fail(LBMI_ERR_ARGUMENT, "%s: %s", fn,
			  (r == 0) ? "file too short" : strerror(errno));
	break;
      }
      io += (size_t) r;
    }
    if (ifail) break;
    if (!writing) {
      if (ascii && lbmi_records_from_ascii(text, n/recb, lb->kp.nvel,
					   lb->opts.ndist, (double *) stage) != 0) {
	ifail = lbmi_fail(LBMI_ERR_ARGUMENT, "%s: not a number in a text record", fn);
	break;
      }
      if (hipMemcpyAsync((char *) dev + done, stage, n, hipMemcpyHostToDevice,
			 lb->stream) != hipSuccess ||
	  hipStreamSynchronize(lb->stream) != hipSuccess) {
	ifail = lbmi_fail(LBMI_ERR_HIP, "host to device copy of records");
	break;
      }
    }
    done += n;
  }
  hipHostFree(stage);
  free(text);
  if (close(fd) != 0 && ifail == 0) {
    ifail = lbmi_fail(LBMI_ERR_ARGUMENT, "%s: %s", fn, strerror(errno));
  }
  return ifail;
}

static int lbmi_io_args(lbmi_t * lb, const char * dir, int ntotal_x,
			int offset_x) {
  if (lb == NULL || dir == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  if (lb->f == NULL) return lbmi_fail(LBMI_ERR_STATE, "no distributions bound");
  if ((lb->xdim != X || lb->multi) && lb->opts.cartsz > 1) {
    return lbmi_fail(LBMI_ERR_UNSUPPORTED, "distribution files: a slab along Y or "
		     "Z is not a contiguous byte range of the file (X slabs, or "
		     "the reference's lb_io_write)");
  }
  if (offset_x < 0 || offset_x + lb->kp.nlocal[X] > ntotal_x) {
    return lbmi_fail(LBMI_ERR_ARGUMENT, "planes %d..%d outside 0..%d", offset_x,
		     offset_x + lb->kp.nlocal[X] - 1, ntotal_x - 1);
  }
  return 0;
}

int lbmi_io_file_set(lbmi_t * lb, const lbmi_io_file_t * file) {
  if (lb == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  if (file == NULL) {
    lb->io_file_set = 0;
    return 0;
  }
  if (file->nfile < 1 || file->index < 0 || file->index >= file->nfile ||
      file->file_x0 < 0 || file->file_nx < 1) {
    return lbmi_fail(LBMI_ERR_ARGUMENT, "lbmi_io_file_set: file %d of %d, planes %d + %d",
		     file->index, file->nfile, file->file_x0, file->file_nx);
  }
  for (int ia = 0; ia < 3; ia++) {
    if (file->periodic[ia] != 0 && file->periodic[ia] != 1) {
      return lbmi_fail(LBMI_ERR_ARGUMENT, "lbmi_io_file_set: periodic[%d] = %d", ia,
		       file->periodic[ia]);
    }
  }
  lb->io_file = *file;
  lb->io_file_set = 1;
  return 0;
}

/* this rank's file: the whole lattice unless lbmi_io_file_set said otherwise */

static int lbmi_io_file(lbmi_t * lb, int ntotal_x, int offset_x, lbmi_io_file_t * file) {
  lbmi_io_file_t one = {1, 0, 0, 0, {1, 1, 1}};
  one.file_nx = ntotal_x;
  *file = lb->io_file_set ? lb->io_file : one;
  if (file->nfile == 1) {
    file->file_x0 = 0;
    file->file_nx = ntotal_x;
  }
  if (file->nfile > 1 && (lb->io_ascii & LBMI_IO_SINGLE)) {
    return lbmi_fail(LBMI_ERR_UNSUPPORTED, "the single mode has one file");
  }
  if (offset_x < file->file_x0 ||
      offset_x + lb->kp.nlocal[X] > file->file_x0 + file->file_nx ||
      file->file_x0 + file->file_nx > ntotal_x) {
    return lbmi_fail(LBMI_ERR_ARGUMENT, "planes %d..%d are not in file %d of %d "
		     "(planes %d..%d of %d)", offset_x, offset_x + lb->kp.nlocal[X] - 1,
		     file->index, file->nfile, file->file_x0,
		     file->file_x0 + file->file_nx - 1, ntotal_x);
  }
  return 0;
}

int lbmi_io_format_set(lbmi_t * lb, int fmt) {
  if (lb == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  if (fmt < 0 || fmt > (LBMI_IO_ASCII | LBMI_IO_SINGLE)) {
    return lbmi_fail(LBMI_ERR_ARGUMENT, "lbmi_io_format_set: %d", fmt);
  }
  if (fmt == (LBMI_IO_ASCII | LBMI_IO_SINGLE)) {
    /* lb_f_write_ascii (model.c:1341-1383) writes "%d %d %d " and "%le " items
     * of no fixed width, and lb_io_info_set gives the text record no size, so
     * io_write_data_s buffers nothing: there is no file of the reference to
     * be equal to */
    return lbmi_fail(LBMI_ERR_UNSUPPORTED, "text records in the single mode");
  }
  lb->io_ascii = fmt;
  return 0;
}

int lbmi_lb_io_write(lbmi_t * lb, const char * dir, int timestep,
		     int ntotal_x, int offset_x) {
  char fn[1024];
  size_t plane, nbytes;
  lbmi_io_file_t file;
  int ifail = lbmi_io_args(lb, dir, ntotal_x, offset_x);
  if (ifail) return ifail;
  ifail = lbmi_io_file(lb, ntotal_x, offset_x, &file);
  if (ifail) return ifail;
  HIPCHECK(hipSetDevice(lb->device));
  ifail = lbmi_lb_flush(lb);
  if (ifail) return ifail;
  if (offset_x == file.file_x0) {
    /* (io_metadata_write: rank 0 of the file's communicator) */
    int ntotal[3] = {ntotal_x, lb->kp.nlocal[Y], lb->kp.nlocal[Z]};
    ifail = lbmi_io_metadata_write_file(dir, "dist", lb->kp.nvel, lb->opts.ndist,
					ntotal, lb->io_ascii, &file);
    if (ifail) return ifail;
    if (lb->io_ascii & LBMI_IO_SINGLE) {
      /* io_write_data_s -> io_write_metadata; the lines of the other ranks
       * by the rule cs_init cut the lattice with, which this rank's own
       * planes have to agree with */
      int nslab[LBMI_RING_MAX];
      const int sz = lb->opts.cartsz;
      if (sz > LBMI_RING_MAX) return lbmi_fail(LBMI_ERR_UNSUPPORTED, "%d ranks", sz);
      lbmi_cs_parts(ntotal_x, sz, nslab);
      if (nslab[0] != lb->kp.nlocal[X]) {
	return lbmi_fail(LBMI_ERR_UNSUPPORTED, "slabs not cut as cs_init cuts them: "
			 "write %s.001-001.meta with lbmi_io_single_metadata_write",
			 "dist");
      }
      ifail = lbmi_io_single_metadata_write(dir, "dist", lb->kp.nvel, lb->opts.ndist,
					    ntotal, X, sz, nslab);
      if (ifail) return ifail;
    }
  }
  ifail = lbmi_io_filename_file(dir, "dist", timestep, lb->io_ascii, &file, fn, sizeof(fn));
  if (ifail) return ifail;
  /* fprime is dead between steps once nothing is pending: pack there */
  KCHECK(lbmi_k_records(&lb->kp, lb->opts.ndist, lb->f, lb->fprime, 1, lb->stream));
  plane = sizeof(double)*(size_t) lb->kp.nvel*lb->opts.ndist*lb->kp.nlocal[Y]*lb->kp.nlocal[Z];
  nbytes = plane*(size_t) lb->kp.nlocal[X];
  /* (io_impl_mpio.c:179-272: the file view is the rank's block at its
   * position in the FILE's block) */
  return lbmi_io_transfer(lb, fn, 1, lb->fprime, nbytes,
			  (off_t) (plane*(size_t) (offset_x - file.file_x0)));
}

int lbmi_lb_io_read(lbmi_t * lb, const char * dir, int timestep,
		    int ntotal_x, int offset_x) {
  char fn[1024];
  size_t plane, nbytes;
  lbmi_io_file_t file;
  int ifail = lbmi_io_args(lb, dir, ntotal_x, offset_x);
  if (ifail) return ifail;
  ifail = lbmi_io_file(lb, ntotal_x, offset_x, &file);
  if (ifail) return ifail;
  HIPCHECK(hipSetDevice(lb->device));
  ifail = lbmi_io_filename_file(dir, "dist", timestep, lb->io_ascii, &file, fn, sizeof(fn));
  if (ifail) return ifail;
  plane = sizeof(double)*(size_t) lb->kp.nvel*lb->opts.ndist*lb->kp.nlocal[Y]*lb->kp.nlocal[Z];
  nbytes = plane*(size_t) lb->kp.nlocal[X];
  ifail = lbmi_io_transfer(lb, fn, 0, lb->fprime, nbytes,
			   (off_t) (plane*(size_t) (offset_x - file.file_x0)));
  if (ifail) return ifail;
  /* replaces the state: whatever was pending is dropped */
  return lbmi_lb_records_unpack(lb, lb->fprime);
}

int lbmi_synchronize(lbmi_t * lb) {
  if (lb == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  HIPCHECK(hipSetDevice(lb->device));
  HIPCHECK(hipStreamSynchronize(lb->comm_stream));
  HIPCHECK(hipStreamSynchronize(lb->bnd_stream));
  HIPCHECK(hipStreamSynchronize(lb->stream));
  return lbmi_wall_err_report(lb);
}

int lbmi_set_stream(lbmi_t * lb, void * stream) {
  if (lb == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  HIPCHECK(hipSetDevice(lb->device));
  HIPCHECK(hipStreamSynchronize(lb->comm_stream));
  HIPCHECK(hipStreamSynchronize(lb->stream));
  lb->stream = (hipStream_t) stream;
  lb->nev = 0;                       /* timing events belong to a stream */
  return 0;
}

int lbmi_stream(lbmi_t * lb, void ** stream) {
  if (lb == NULL || stream == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  *stream = (void *) lb->stream;
  return 0;
}

/*****************************************************************************
 *
 *  RCCL ring over the X slabs. Replaces the MPI Cartesian communicator of
 *  the reference (coords.c, halo_swap.c:742-784) for the one decomposed
 *  direction.
 *
 *****************************************************************************/

int lbmi_comm_unique_id(void * id) {
  ncclUniqueId nid;
  if (id == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  if (sizeof(ncclUniqueId) > LBMI_UNIQUE_ID_BYTES) {
    return lbmi_fail(LBMI_ERR_RCCL, "ncclUniqueId is %zu bytes",
		     sizeof(ncclUniqueId));
  }
  NCCLCHECK(ncclGetUniqueId(&nid));
  memset(id, 0, LBMI_UNIQUE_ID_BYTES);
  memcpy(id, &nid, sizeof(nid));
  return 0;
}

/* The buffers of the X exchange: four plane buffers large enough for every
 * component (full scheme, or a generic field of up to LBMI_NVEL_MAX
 * components), and the four of the FUSED step (reduced selection) */

static int lbmi_comm_buffers(lbmi_t * lb) {
  size_t bytes;
  const int dim = lb->xdim;
  size_t psz = (size_t) (lb->kp.nsite/lb->kp.nall[dim]);
  int nred = lb->sel_reduced[dim].nlo > lb->sel_reduced[dim].nhi
    ? lb->sel_reduced[dim].nlo : lb->sel_reduced[dim].nhi;
  /* (the largest plane among the decomposed directions) */
  for (int d = 0; d < 3; d++) {
    size_t pd = (size_t) (lb->kp.nsite/lb->kp.nall[d]);
    if (lb->multi && lb->csz[d] > 1 && pd > psz) psz = pd;
  }
  lb->xbuf_doubles = psz*LBMI_NVEL_MAX;
  bytes = sizeof(double)*lb->xbuf_doubles;
  if (hipMalloc((void **) &lb->sendlo, bytes) != hipSuccess ||
      hipMalloc((void **) &lb->sendhi, bytes) != hipSuccess ||
      hipMalloc((void **) &lb->recvlo, bytes) != hipSuccess ||
      hipMalloc((void **) &lb->recvhi, bytes) != hipSuccess) {
    return lbmi_fail(LBMI_ERR_HIP, "hipMalloc (halo staging buffers) failed");
  }
  bytes = sizeof(double)*psz*(size_t) nred;
  for (int k = 0; k < 4; k++) {
    if (hipMalloc((void **) &lb->fx[k], bytes) != hipSuccess ||
	hipMemset(lb->fx[k], 0, bytes) != hipSuccess) {
      return lbmi_fail(LBMI_ERR_HIP, "hipMalloc (exchange buffers of the fused step) failed");
    }
  }
  lb->xsend_valid = 0; lb->halo_fresh = 0;
  return 0;
}

int lbmi_comm_init(lbmi_t * lb, const void * id) {

  ncclUniqueId nid;
  int ifail;

  if (lb == NULL || id == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  if (lb->have_comm) return lbmi_fail(LBMI_ERR_STATE, "communicator exists");
  HIPCHECK(hipSetDevice(lb->device));

  memcpy(&nid, id, sizeof(nid));
  NCCLCHECK(ncclCommInitRank(&lb->comm, lb->opts.cartsz, nid,
			     lb->opts.cartrank));
  lb->have_comm = 1;

  ifail = lbmi_comm_buffers(lb);
  if (ifail == 0) ifail = lbmi_lb_mode_set(lb, lb->opts.mode);   /* Z slabs: no FUSED */
  if (ifail) lbmi_comm_free(lb);
  return ifail;
}

/* The same ring inside one process: every handle of it on one device, driven
 * by a thread of its own (an exchange waits for the neighbours' posts) */

int lbmi_comm_init_ring(lbmi_t * lb, lbmi_ring_t * ring) {
  int ifail;
  if (lb == NULL || ring == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  if (lb->have_comm) return lbmi_fail(LBMI_ERR_STATE, "communicator exists");
  if (ring->nranks != lb->opts.cartsz) {
    return lbmi_fail(LBMI_ERR_ARGUMENT, "a ring of %d for cartsz = %d",
		     ring->nranks, lb->opts.cartsz);
  }
  if (ring->nranks > 1 && lb->stream != lb->own_stream) {
    /* on a stream shared by the ranks, the work of one queues up behind the
     * waits of another for data that one has yet to send */
    return lbmi_fail(LBMI_ERR_STATE, "a ring of several handles in one "
		     "process needs each on its own stream (lbmi_set_stream "
		     "has given this one the caller's)");
  }
  HIPCHECK(hipSetDevice(lb->device));
  pthread_mutex_lock(&ring->mu);
  if (ring->device[lb->opts.cartrank] >= 0) {
    pthread_mutex_unlock(&ring->mu);
    return lbmi_fail(LBMI_ERR_STATE, "rank %d of the ring is taken", lb->opts.cartrank);
  }
  ring->device[lb->opts.cartrank] = lb->device;
  ring->nattached += 1;
  pthread_mutex_unlock(&ring->mu);
  /* the receiver pulls from its neighbours' devices: this device must be
   * able to reach both (whichever of them are other devices; a neighbour
   * that attaches later enables its own direction, and the copies name both
   * devices, so an order of attachment is not required) */
  for (int dn = 0; dn < 6; dn++) {
    int pn[2] = {0, 0};
    int peer;
    int pdev;
    if (lb->csz[dn/2] < 2) continue;
    lbmi_cart_nbr(lb->csz, lb->ccoord, dn/2, &pn[0], &pn[1]);
    peer = pn[dn % 2];
    pthread_mutex_lock(&ring->mu);
    pdev = ring->device[peer];
    pthread_mutex_unlock(&ring->mu);
    if (pdev >= 0 && pdev != lb->device) {
      int can = 0;
      hipError_t e;
      HIPCHECK(hipDeviceCanAccessPeer(&can, lb->device, pdev));
      if (!can) {
	pthread_mutex_lock(&ring->mu);
	ring->device[lb->opts.cartrank] = -1;
	ring->nattached -= 1;
	pthread_mutex_unlock(&ring->mu);
	return lbmi_fail(LBMI_ERR_UNSUPPORTED, "device %d cannot access device "
			 "%d as a peer", lb->device, pdev);
      }
      e = hipDeviceEnablePeerAccess(pdev, 0);
      if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) {
	(void) hipGetLastError();
	pthread_mutex_lock(&ring->mu);
	ring->device[lb->opts.cartrank] = -1;
	ring->nattached -= 1;
	pthread_mutex_unlock(&ring->mu);
	return lbmi_fail(LBMI_ERR_HIP, "hipDeviceEnablePeerAccess(%d): %s", pdev,
			 hipGetErrorString(e));
      }
      (void) hipGetLastError();        /* (already enabled is not an error) */
      if (pdev < 64) lb->peers_enabled |= (1ULL << pdev);
    }
  }
  lb->ring = ring;
  lb->have_comm = 1;
  ifail = lbmi_comm_buffers(lb);
  if (ifail == 0) ifail = lbmi_lb_mode_set(lb, lb->opts.mode);   /* Z slabs: no FUSED */
  if (ifail) lbmi_comm_free(lb);
  return ifail;
}

/* Ranks of the ring this handle exchanges with, and the transport */

int lbmi_comm_info(lbmi_t * lb, int * nranks, int * rank, int * transport) {
  if (lb == NULL) return lbmi_fail(LBMI_ERR_ARGUMENT, "NULL");
  if (nranks) *nranks = 0;
  if (rank) *rank = -1;
  if (transport) *transport = 0;
  if (!lb->have_comm) return 0;
  if (lb->ring) {
    if (nranks) *nranks = lb->ring->nranks;
    if (rank) *rank = lb->opts.cartrank;
    if (transport) *transport = 2;
    return 0;
  }
  {
    int n = 0, r = -1;
    NCCLCHECK(ncclCommCount(lb->comm, &n));
    NCCLCHECK(ncclCommUserRank(lb->comm, &r));
    if (nranks) *nranks = n;
    if (rank) *rank = r;
    if (transport) *transport = 1;
  }
  return 0;
}

int lbmi_comm_free(lbmi_t * lb) {
  if (lb == NULL) return 0;
  if (lb->have_comm) {
    if (lb->sendlo) hipFree(lb->sendlo);
    if (lb->sendhi) hipFree(lb->sendhi);
    if (lb->recvlo) hipFree(lb->recvlo);
    if (lb->recvhi) hipFree(lb->recvhi);
    lb->sendlo = lb->sendhi = lb->recvlo = lb->recvhi = NULL;
    for (int k = 0; k < 4; k++) {
      if (lb->fx[k]) hipFree(lb->fx[k]);
      lb->fx[k] = NULL;
    }
    lb->xsend_valid = 0; lb->halo_fresh = 0;
    if (lb->ring) {
      pthread_mutex_lock(&lb->ring->mu);
      lb->ring->nattached -= 1;
      lb->ring->device[lb->opts.cartrank] = -1;
      pthread_mutex_unlock(&lb->ring->mu);
      lb->ring = NULL;
    }
    else {
      ncclCommDestroy(lb->comm);
    }
    lb->have_comm = 0;
  }
  return 0;
}
