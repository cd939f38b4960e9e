/*****************************************************************************
 *
 *  lbmi.h
 *
 *  C-ABI of liblbmi: an MI355X (gfx950) native lattice-Boltzmann time step
 *  -- collision, halo swap, propagation and the conserved-moment statistics --
 *  behind the call surface of the reference (zazu29/ludwig v0.20.1):
 *
 *      lb_collide()      src/collision.h:27     (called ludwig.c:802)
 *      lb_halo()         src/lb_data.h:159      (called ludwig.c:816)
 *      lb_halo_swap()    src/lb_data.h:160
 *      lb_propagation()  src/propagation.h:21   (called ludwig.c:860)
 *      lb_memcpy()       src/lb_data.h:156
 *      stats_distribution_print() / distribution_stats_momentum()
 *                        src/stats_distribution.c:55,201
 *
 *  and, one step out from that path: the
 *  hydro housekeeping and field halos, the symmetric free-energy coupling
 *  (gradients, force, Cahn-Hilliard), the distribution files, the
 *  two-distribution collision and bounce-back on links for flat walls.
 *
 *  Plain C: pointers, ints and doubles only. All array arguments are DEVICE
 *  pointers (hipMalloc or equivalent) unless stated otherwise. There is no
 *  CPU fallback: every compute entry point launches hand-written HIP kernels
 *  and fails (negative return, lbmi_last_error()) when no device is present.
 *
 *  Memory layout is the reference's, unchanged (so that other Ludwig kernels
 *  can keep dereferencing lb->target->f with LB_ADDR):
 *
 *    SoA ("reverse", -DADDR_SOA, memory.h:187-188):
 *      f[(n*nvel + p)*nsite + index]           LB_ADDR, lb_data.h:136
 *      hydro vectors: v[nsite*ia + index]      addr_rank1, memory.h:184
 *      index = (nhalo+ic-1)*nall[Y]*nall[Z] + (nhalo+jc-1)*nall[Z]
 *              + (nhalo+kc-1)                  cs_index, coords.c:617-631
 *      nall[] = nlocal[] + 2*nhalo, nsite = nall[X]*nall[Y]*nall[Z]
 *
 *  All functions return 0 on success and a negative lbmi_error_t otherwise;
 *  lbmi_last_error() then returns a static description (the reference itself
 *  aborts via pe_fatal()/tdpAssert(); the shim in INTEGRATION.md maps a
 *  non-zero return to pe_fatal()).
 *
 *  Threading: one host thread per handle, one GPU per handle (the reference
 *  runs one MPI rank per GPU, ludwig.c:467-492). No global mutable state
 *  apart from the thread-local error string.
 *
 *****************************************************************************/

#ifndef LBMI_H
#define LBMI_H

#ifdef __cplusplus
extern "C" {
#endif

#include <stddef.h>
#include <stdint.h>

#define LBMI_VERSION_MAJOR 0
#define LBMI_VERSION_MINOR 3

typedef struct lbmi_s lbmi_t;           /* opaque handle ~ lb_t + halo_swap_t */
typedef struct lbmi_ring_s lbmi_ring_t; /* the peer transport: a ring of handles inside one process */

typedef enum lbmi_error_e {
  LBMI_SUCCESS         =  0,
  LBMI_ERR_ARGUMENT    = -1,            /* bad argument (assert() upstream) */
  LBMI_ERR_UNSUPPORTED = -2,            /* e.g. d3q27 + TRT, ndist = 3      */
  LBMI_ERR_NODEVICE    = -3,            /* no HIP device / wrong arch       */
  LBMI_ERR_HIP         = -4,            /* a HIP call or an allocation failed */
  LBMI_ERR_RCCL        = -5,            /* an RCCL call failed              */
  LBMI_ERR_STATE       = -6             /* call out of order                */
} lbmi_error_t;

/* Relaxation scheme: lb_relaxation_enum_t, src/lb_data_options.h:19-22 */
typedef enum lbmi_relaxation_e {
  LBMI_RELAXATION_M10 = 0,
  LBMI_RELAXATION_BGK = 1,
  LBMI_RELAXATION_TRT = 2
} lbmi_relaxation_t;

/* Halo scheme. FULL has the semantics of LB_HALO_TARGET (halo_swap.c:709):
 * every population in the complete width-1 shell. REDUCED has those of
 * LB_HALO_OPENMP_REDUCED (model.c:1192-1219): only populations that
 * propagate into the interior are guaranteed. */
typedef enum lbmi_halo_e {
  LBMI_HALO_FULL    = 0,
  LBMI_HALO_REDUCED = 2
} lbmi_halo_t;

/* How the three calls of a time step are executed.
 * EAGER: lb_collide = in-place collision kernel, lb_halo = halo kernels,
 *        lb_propagation = pull kernel + pointer swap; f is canonical after
 *        every call (exactly the reference's observable behaviour).
 * FUSED: lb_halo and lb_propagation only record that a halo swap and a
 *        propagation are pending; the next lb_collide runs ONE kernel that
 *        pulls (propagation), wraps periodic directions by index (halo),
 *        collides, and writes fprime, then swaps. lbmi_lb_flush() (called by
 *        lbmi_lb_memcpy_d2h, lbmi_lb_moments and lbmi_lb_f) materialises
 *        the pending halo + propagation so that any reader sees the same
 *        f as in EAGER mode. Requires an all-fluid or bounce-back-free step
 *        (nothing may modify f between lb_collide and lb_propagation).
 *        With lbmi_tune(lb, "blocked", 1) on a single GPU the deferred state
 *        is kept in a block-contiguous order ([site/256][p][site%256] in the
 *        same nsite*nvel doubles) that the memory system serves ~5 % faster
 *        than nvel far-apart streams; the flush converts it back.
 * INPLACE: as FUSED, but f is streamed IN PLACE (AA pattern): lb_collide
 *        alternates between a local collision that stores into swapped
 *        slots and a pull + collide + push kernel, both reading and writing
 *        the same addresses of the ONE array f (fprime is only touched by
 *        lbmi_lb_flush). The second kernel performs the propagation of its
 *        own step early, so the lb_halo / lb_propagation calls that follow
 *        it only acknowledge. Same legality conditions as FUSED, and
 *        cartsz == 1 (with slabs it falls back to FUSED).
 * FUSED_HALO: lb_collide and lb_halo are as observable as in EAGER -- after
 *        them f is the reference's post-collision state with its halo, in the
 *        reference's order, so link bounce-back (wall_bbl,
 *        bounce_back_on_links) and open boundaries may modify it -- and only
 *        lb_propagation is deferred: the next lb_collide pulls straight from
 *        that array, halo sites included (no index wrap). Two passes over f
 *        per step instead of EAGER's three; the mode for walls and colloids.
 *        Readers of f AFTER lb_propagation need lbmi_lb_flush. */
typedef enum lbmi_mode_e {
  LBMI_MODE_EAGER = 0,
  LBMI_MODE_FUSED = 1,
  LBMI_MODE_INPLACE = 2,
  LBMI_MODE_FUSED_HALO = 3
} lbmi_mode_t;

typedef struct lbmi_options_s {
  int nvel;                 /* 19 or 27 (lb_data.h:33-44)                    */
  int ndist;                /* 1, or 2 (symmetric_lb; not LBMI_MODE_INPLACE)  */
  int nlocal[3];            /* local lattice extent (cs_nlocal)              */
  int nhalo;                /* halo width of the allocation (cs_nhalo)       */
  int device;               /* HIP device ordinal, or -1: current device     */
  int mode;                 /* lbmi_mode_t                                   */
  int halo_scheme;          /* lbmi_halo_t                                   */
  int cartsz;               /* number of slabs (1 = single GPU)              */
  int cartrank;             /* this rank's slab, 0 <= cartrank < cartsz      */
  int cartdim;              /* the decomposed direction: 0 = X (the default:
			       boundary planes are contiguous, and the fused
			       step overlaps their exchange with the interior
			       launch), 1 = Y, 2 = Z (grid 1_N_1 / 1_1_N,
			       coords_rt.c:46-47: planes are gathered and
			       scattered, halo_swap.c:1074-1274. Y: the fused
			       step overlaps the exchange with ONE launch over
			       all planes and redoes the two boundary planes
			       against the exchange buffers afterwards. Z: a
			       plane is one value out of every row, FUSED runs
			       as FUSED_HALO -- the exchange where lb_halo is
			       called, the propagation folded into the next
			       collision: measured faster).
			       3 = LBMI_CART_GENERAL: the Cartesian
			       decomposition of cartgrid / cartcoords (the
			       reference's default for N ranks is
			       MPI_Dims_create's, e.g. 2_2_2 for eight,
			       coords.c:520-560). With more than one direction
			       decomposed the halo swap is the reference's
			       sequence of passes X, Y, Z, each over the full
			       extent of the other two (halo_swap.c:709-1063:
			       edges and corners complete), device to device;
			       FUSED runs as FUSED_HALO.                      */
  int cartgrid[3];          /* cartdim 3: ranks along X, Y, Z (product =
			       cartsz); otherwise must be zero               */
  int cartcoords[3];        /* cartdim 3: this rank's coordinates; cartrank =
			       (cx*gy + cy)*gz + cz, the rank of
			       MPI_Cart_create without reordering            */
} lbmi_options_t;
enum {LBMI_CART_GENERAL = 3};

/* Borrowed per-call fields of lb_collide(): hydro_t and map_t device arrays
 * (collision.c:200-202, 329-333, 571-579). Any pointer may be NULL:
 * force == NULL is a zero force field; status == NULL is all MAP_FLUID;
 * rho/u == NULL are not written; eta == NULL is the constant viscosity.
 * Zero-initialise the struct: members may be added at its end. */
typedef struct lbmi_hydro_s {
  const double * force;     /* hydro->force->data, 3*nsite                   */
  const char   * status;    /* map->status, nsite bytes, 0 = MAP_FLUID       */
  double       * rho;       /* hydro->rho->data, nsite                       */
  double       * u;         /* hydro->u->data, 3*nsite                       */
  const double * eta;       /* hydro->eta->data, nsite: the local shear
			       viscosity of a viscosity model (lb_collide
			       with visc != NULL, collision.c:386-404; the
			       bulk viscosity keeps the ratio of
			       lbmi_set_relaxation); NULL = constant         */
  long long      nsite;     /* hydro->nsite: the distance between the
			       components of force and of u, in doubles;
			       0 = the lattice's own nsite. (With Lees-
			       Edwards planes the reference's hydro arrays
			       carry buffer planes, lees_edw_nsites, and the
			       distributions do not: hydro.c:75-88,
			       model.c:295.) Honoured by lbmi_lb_collide,
			       its lazy rho/u and lbmi_lb_collide_binary.    */
} lbmi_hydro_t;

/* ---- life cycle: lb_data_create / lb_free (model.c:56-213) -------------- */

int lbmi_options_default(lbmi_options_t * opts);
int lbmi_create(const lbmi_options_t * opts, lbmi_t ** handle);
int lbmi_free(lbmi_t * lb);
const char * lbmi_last_error(void);

int lbmi_nsite(const lbmi_t * lb, size_t * nsite);
int lbmi_nall(const lbmi_t * lb, int nall[3]);

/* Model tables as the reference builds them (lb_d3q19.c, lb_d3q27.c,
 * model.c:358-390); host arrays of nvel, 3*nvel, nvel, nvel*nvel. */
int lbmi_model(int nvel, int8_t * cv, double * wv, double * na, double * ma);

/* ---- parameters -------------------------------------------------------- */

/* lb_collision_relaxation_times_set (collision.c:1181-1264) with the
 * global physics_t values passed explicitly. */
int lbmi_set_relaxation(lbmi_t * lb, int scheme, double rho0,
			double eta_shear, double eta_bulk);
/* force_global of lb_collision_parameters_commit (collision.c:1928-1980) */
int lbmi_set_body_force(lbmi_t * lb, const double fbody[3]);
/* The relaxation rates in use: rtau_shear, rtau_bulk, ghost even, ghost odd */
int lbmi_relaxation_rates(const lbmi_t * lb, double rtau[4]);

/* ---- stateless kernels on caller-owned device arrays -------------------- */

/* lb_collision_mrt1 (collision.c:223-599): in-place collision of interior
 * fluid sites. */
int lbmi_collide(lbmi_t * lb, double * f, const lbmi_hydro_t * hydro);

/* lb_halo_swap (model.c:565-595): fill the width-1 halo shell of f (single
 * rank: periodic wrap in all directions; cartsz > 1: X through RCCL). */
int lbmi_halo(lbmi_t * lb, double * f, int scheme);

/* The X pass of the halo swap split into its device-side halves, for
 * callers that bring their own transport (e.g. GPU-aware MPI in place of
 * halo_swap.c:762-881): pack the two boundary planes into contiguous device
 * buffers, exchange them, unpack into the halo planes, then run the local
 * Y and Z passes. Buffer layout [component][plane site]; message lengths in
 * doubles from lbmi_halo_x_count().
 *   sendlo: first interior plane  -> lower neighbour's recvhi
 *   sendhi: last interior plane   -> upper neighbour's recvlo          */
int lbmi_halo_x_count(lbmi_t * lb, int scheme, size_t * nsendlo,
		      size_t * nsendhi);
int lbmi_halo_x_pack(lbmi_t * lb, const double * f, int scheme,
		     double * sendlo, double * sendhi);
int lbmi_halo_x_unpack(lbmi_t * lb, double * f, int scheme,
		       const double * recvlo, const double * recvhi);
int lbmi_halo_yz(lbmi_t * lb, double * f, int scheme);

/* lb_propagation_kernel (propagation.c:162-212): fprime <- pull(f). The
 * caller swaps the pointers (lb_model_swapf, propagation.c:223-252). */
int lbmi_propagate(lbmi_t * lb, const double * f, double * fprime);

/* Fused propagation(t) + collision(t+1): fprime <- collide(pull(f)).
 * wrap != 0: periodic directions local to this rank are wrapped by index
 * arithmetic (no halo needed in Y, Z, and in X when cartsz == 1);
 * wrap == 0: f must have a valid halo shell. */
int lbmi_propagate_collide(lbmi_t * lb, const double * f, double * fprime,
			   const lbmi_hydro_t * hydro, int wrap);

/* halo_swap_packed for a generic SoA field of nel components, e.g.
 * hydro->u (hydro_u_halo, hydro.c:190-215). */
int lbmi_field_halo(lbmi_t * lb, int nel, double * data);

/* stats_distribution_print + distribution_stats_momentum
 * (stats_distribution.c:55-117, 201-350) over interior fluid sites.
 * out (HOST array of 9): volume, sum rho, sum rho^2, min rho, max rho,
 * g_x, g_y, g_z (Kahan-compensated), 0. Local to this rank. */
int lbmi_moments(lbmi_t * lb, const double * f, const char * status,
		 double out[9]);

/* ---- the lb_t-like stateful surface ------------------------------------ */

/* Attach the two distribution arrays (lb->target->f, lb->target->fprime).
 * With f == NULL the library allocates (and owns) both, zero-initialised
 * as lb_data_create does (model.c:106-109,131-147). */
int lbmi_lb_bind(lbmi_t * lb, double * f, double * fprime);
/* Current arrays: what lb->target->f / fprime must point at after a call
 * (the swap of propagation.c:240-248). */
int lbmi_lb_pointers(lbmi_t * lb, double ** f, double ** fprime);
/* The same for a caller that keeps the pair in DEVICE memory (the members f,
 * fprime of the device copy of lb_t, which lb_model_swapf rewrites with
 * blocking copies, propagation.c:240-248): f_slot and fprime_slot are device
 * addresses of two double*; they receive the current pair from a one-thread
 * kernel on the handle's stream, ordered with the step and without a host
 * synchronisation. */
int lbmi_lb_pointers_store(lbmi_t * lb, double ** f_slot, double ** fprime_slot);

int lbmi_lb_collide(lbmi_t * lb, const lbmi_hydro_t * hydro);  /* lb_collide */
int lbmi_lb_halo(lbmi_t * lb);                                 /* lb_halo    */
int lbmi_lb_propagation(lbmi_t * lb);                          /* lb_propagation */
int lbmi_lb_flush(lbmi_t * lb);
/* Another lbmi_mode_t for an existing handle, at any point of the step: a
 * flush, then the calls that follow run in the new mode. */
int lbmi_lb_mode_set(lbmi_t * lb, int mode);
/* nsteps x (lbmi_lb_collide, lbmi_lb_halo, lbmi_lb_propagation): the LB part
 * of the reference's main loop for callers that have nothing to do in
 * between (a host language with expensive foreign calls, a benchmark). */
int lbmi_lb_run(lbmi_t * lb, const lbmi_hydro_t * hydro, int nsteps);

/* ndist = 2, free_energy symmetric_lb (LBMI_MODE_EAGER; LBMI_MODE_FUSED_HALO:
 * lbmi_lb_propagation of both distributions is deferred into the next
 * lbmi_lb_collide_binary, and lbmi_lb_phi_to_field called in between takes
 * phi of the propagated state from the pending array; LBMI_MODE_FUSED: on one
 * GPU the halo swap of both is deferred as well -- the pulls wrap by index --
 * and on slabs it is FUSED_HALO): the second distribution carries the order
 * parameter. f holds both,
 * f[(n*nvel + p)*nsite + index]; lbmi_lb_halo and lbmi_lb_propagation move
 * both; lbmi_lb_moments looks at n = 0 (as stats_distribution.c does).
 *   lbmi_lb_phi_to_field   phi_lb_to_field (phi_lb_coupler.c:39-112):
 *                          phi = sum_p g_p at the interior sites.
 *   lbmi_lb_collide_binary lb_collide -> lb_collision_binary
 *                          (collision.c:143-163, 610-1027) without noise:
 *                          reads hydro->force, writes hydro->u (not rho; no
 *                          status test, as the reference); phi, grad, delsq =
 *                          the field and its field_grad_compute arrays at
 *                          the interior sites; mobility M gives the
 *                          relaxation rate 2/(1 + 2M) of the phi flux. */
/* Flat walls and solid sites with bounce-back on links (wall.c), EAGER or
 * FUSED_HALO mode.
 * The step with walls is lb_collide, lb_halo, wall_bbl, lb_propagation
 * (ludwig.c:802-860): "no halo updates between bounce back and propagation".
 *   lbmi_wall_map        wall_init_map (wall.c:1219-1268): MAP_BOUNDARY (1)
 *                        into the DEVICE map `status` (nsite chars) at every
 *                        site, halo included, whose GLOBAL coordinate in a
 *                        wall direction is 0 or ntotal+1 (with X slabs: the
 *                        low X wall on the first rank, the high one on the
 *                        last).
 *   lbmi_wall_links_build  wall_init_boundaries + wall_init_uw (wall.c:
 *                        381-470, 864-890): a link for every interior
 *                        MAP_FLUID site i and p >= 1 with i + c_p
 *                        MAP_BOUNDARY, in the reference's order; with walls
 *                        in exactly one direction the links carry the
 *                        top/bottom wall velocity. Built on the host from a
 *                        copy of the map (initialisation only).
 *   lbmi_wall_links      copy the links out (host arrays of nlink ints, any
 *                        may be NULL): fluid site, solid site, p, velocity id.
 *   lbmi_wall_velocity_set  wall_param_t ubot / utop.
 *   lbmi_wall_bbl        wall_bbl (wall.c:960-1107): f[j, nvel-p] =
 *                        f[i, p] - 2 rcs2 w_p rho0 c_p.u_w on every link (both
 *                        distributions with ndist = 2) and the momentum
 *                        given to the walls accumulated on the device.
 *   lbmi_wall_momentum   wall_momentum (wall.c:1299-1330): read the
 *                        accumulated momentum and zero it. */
int lbmi_wall_map(lbmi_t * lb, const int isboundary[3], char * status);
int lbmi_wall_links_build(lbmi_t * lb, const char * status,
			  const int isboundary[3], int * nlink);
int lbmi_wall_links(lbmi_t * lb, int * linki, int * linkj, int * linkp,
		    int * linku);
int lbmi_wall_velocity_set(lbmi_t * lb, const double ubot[3],
			   const double utop[3]);
int lbmi_wall_bbl(lbmi_t * lb);
/* The same on DEVICE link arrays the caller owns (the reference's
 * wall->target->linki, linkj, linkp, linku), fnet = 3 doubles on the device
 * that the momentum is added to (wall->target->fnet). The kernels skip a
 * record that would address outside the distributions and report it:
 * LBMI_ERR_ARGUMENT from the first call that sees given arrays (that call
 * waits for its kernel), or from the call after the launch that met it. */
int lbmi_wall_bbl_arrays(lbmi_t * lb, int nlink, const int * linki,
			 const int * linkj, const int * linkp,
			 const int * linku, const double ubot[3],
			 const double utop[3], double * fnet);
int lbmi_wall_momentum(lbmi_t * lb, double fnet[3]);
/* Links made by the caller: HOST arrays of nlink ints as wall_init_boundaries
 * and wall_init_uw leave them in wall->linki, linkj, linkp, linku (wall.c:
 * 399-451, 864-890). Every record is checked on the host (0 <= i < nsite,
 * 1 <= p < nvel, j = i + c_p inside the array, u in {0, 1, 2};
 * LBMI_ERR_ARGUMENT names the first bad one) and copied: lbmi_wall_bbl then
 * works on device arrays the handle owns, whatever becomes of the caller's. */
int lbmi_wall_links_set(lbmi_t * lb, int nlink, const int * linki,
			const int * linkj, const int * linkp,
			const int * linku);
/* Where lbmi_wall_bbl adds the momentum: 3 doubles on the DEVICE that the
 * caller owns (wall->target->fnet), or NULL = the handle's accumulator
 * (lbmi_wall_momentum). The pointer is kept. */
int lbmi_wall_fnet_bind(lbmi_t * lb, double * fnet);
/* map->target->status (DEVICE, nsite chars) for the MAP_COLLOID test of the
 * bounce-back kernels (wall.c:1046-1061, 1146-1161): a link whose fluid site a
 * colloid covers is left to the colloid's own bounce-back and only enters the
 * momentum accounting. NULL (the default): no test. The pointer is kept. */
int lbmi_wall_status_set(lbmi_t * lb, const char * status);

/* Partial slip at flat walls (wall_slip_t, wall.h:25-50).
 *   lbmi_wall_slip_set   wall_slip + wall_init_boundaries_slip (wall.c:285-316,
 *                        489-593, with wall_link_normal :606-642,
 *                        wall_link_slip_direction :658-693, wall_link_slip
 *                        :707-757): slip fractions 0 <= s <= 1 of the bottom
 *                        and top wall of each direction; for every link built by
 *                        lbmi_wall_links_build the partner fluid site k, the
 *                        partner direction q and the index of s (faces; edges
 *                        = mean of the two faces; corners no slip). `status` =
 *                        the DEVICE map the links were built from. All
 *                        fractions zero: slip off again. From then on
 *                        lbmi_wall_bbl runs wall_bbl_slip_kernel (wall.c:
 *                        971, 1118-1205): f[j, nvel-p] = (1-s) f[i,p] +
 *                        s f[k,q], walls at rest, first distribution only,
 *                        as the reference.
 *   lbmi_wall_slip_links copy linkk, linkq, links out (host, nlink ints each).
 *   lbmi_wall_bbl_slip_arrays  the same on DEVICE arrays the caller owns, in
 *                        the reference's types (wall.h:79-81); stab = the 19
 *                        fractions wall->param->slip.s (host). */
int lbmi_wall_slip_set(lbmi_t * lb, const char * status,
		       const double sbot[3], const double stop[3]);
int lbmi_wall_slip_links(lbmi_t * lb, int * linkk, int * linkq, int * links);
/* Slip records made by the caller (HOST arrays wall->linkk, linkq, links in
 * the reference's types, stab = wall->param->slip.s) for the links of
 * lbmi_wall_links_set / _build: checked (0 <= k < nsite, 0 <= q < nvel,
 * 0 <= s < 19), copied, and lbmi_wall_bbl takes the slip kernel from then on. */
int lbmi_wall_slip_links_set(lbmi_t * lb, const int * linkk,
			     const signed char * linkq,
			     const signed char * links, const double stab[19]);
int lbmi_wall_bbl_slip_arrays(lbmi_t * lb, int nlink, const int * linki,
			      const int * linkj, const int * linkp,
			      const int * linkk, const signed char * linkq,
			      const signed char * links,
			      const double stab[19], double * fnet);

typedef struct lbmi_fe_symm_s {
  double a, b, kappa;        /* fe_symm_param_t, symmetric.h */
  double mobility;           /* physics_mobility */
  const double * phi;        /* device, nsite */
  const double * grad;       /* device, 3*nsite */
  const double * delsq;      /* device, nsite */
  long long nsite;           /* distance between the components of grad, in
				doubles (field_grad_t of a field with Lees-
				Edwards buffer planes); 0 = the lattice's
				nsite. Honoured by lbmi_lb_collide_fe and
				lbmi_lb_collide_binary. Zero-initialise the
				struct: members may be added at its end.     */
} lbmi_fe_symm_t;
int lbmi_lb_phi_to_field(lbmi_t * lb, double * phi);
/* lb_collide with fe->use_stress_relaxation (collision.c:413-429; force
 * method relaxation_symm, ludwig.c:1235-1237) for the symmetric free energy
 * and ONE distribution: the stress of fe joins the equilibrium stress (the
 * mobility member is not used). EAGER: in place on the canonical state.
 * FUSED_HALO, and FUSED on one rank: the pending propagation of the step
 * before runs inside this collision (one pass over f, SoA -> SoA; rho and u
 * are stored by every such collision). FUSED on several ranks: the halo swap
 * that mode had only noted is done first, then the same. */
int lbmi_lb_collide_fe(lbmi_t * lb, const lbmi_hydro_t * hydro,
		       const lbmi_fe_symm_t * fe);
int lbmi_lb_collide_binary(lbmi_t * lb, const lbmi_hydro_t * hydro,
			   const lbmi_fe_symm_t * fe);
/* What lbmi_lb_flush would have to do right now: state[0] = a halo swap is
 * pending, state[1] = a propagation is pending (or, INPLACE, already applied
 * early), state[2] = the order of f: 0 the reference's SoA, 1 the internal
 * blocked order of a deferred FUSED state (lbmi_tune "blocked"), 2 the
 * slot-swapped order of INPLACE. All zero: f is what the reference holds. */
int lbmi_lb_state(lbmi_t * lb, int state[3]);

/* lb_memcpy (model.c:228-266): whole-array copies between a HOST array of
 * nvel*nsite doubles and the current f. */
int lbmi_lb_memcpy_h2d(lbmi_t * lb, const double * f_host);
int lbmi_lb_memcpy_d2h(lbmi_t * lb, double * f_host);
/* The caller has overwritten the current f on the device itself (the
 * reference's lb_memcpy host -> device, e.g. after the Lees-Edwards
 * reprojection on the host, model_le.c:72-83): the handle drops what it
 * derived from the old contents (planes of a slab already under way to the
 * neighbours for the next fused step, rho / u still owed). The state must be
 * canonical (lbmi_lb_flush first; LBMI_ERR_STATE otherwise). */
int lbmi_lb_dirty(lbmi_t * lb);
int lbmi_lb_moments(lbmi_t * lb, const char * status, double out[9]);
/* lb_0th_moment (model.c:817-832) of every interior site of the first
 * distribution, summed in p order as there, to a HOST array of
 * nlocal[X]*nlocal[Y]*nlocal[Z] doubles in (ic, jc, kc) order: the numbers
 * stats_distribution_print adds up (stats_distribution.c:73-88). A caller
 * that adds them in that order prints the reference's CPU digits. */
int lbmi_lb_density(lbmi_t * lb, double * rho_host);

/* cahn_stats_reduce (cahn_hilliard_stats.c:123-215; the numbers behind the
 * "[phi]" line of cahn_hilliard_stats and phi->field_init_sum of
 * cahn_hilliard_stats_time0) for any scalar device field of nsite doubles:
 * over the interior sites that are MAP_FLUID in `status` (NULL: all), out
 * (HOST array of 5) = volume, sum (Kahan-compensated), sum of squares,
 * minimum, maximum. One pass and a tree reduction where the reference runs
 * three kernels with a serial loop under a lock each. Local to this rank. */
int lbmi_field_stats(lbmi_t * lb, const double * field, const char * status,
		     double out[5]);

/* hydro->rho and hydro->u on demand (lbmi_tune "hydro_lazy", 1; FUSED and
 * FUSED_HALO): lbmi_lb_collide then does not store them (32 B/site per step)
 * but remembers the arrays it was given, and lbmi_lb_hydro_sync -- or
 * anything in this library that reads u or is about to change the
 * post-collision state: lbmi_lb_flush and what calls it, a propagation that
 * runs at once, lbmi_lb_memcpy_h2d, the free-energy calls given that u --
 * forms them from the post-collision distributions: rho = sum f'_p,
 * u = (sum f'_p c_p - F/2)/rho, the values the collision used (collision.c:
 * 376-382) to rounding. The next lbmi_lb_collide supersedes what is still
 * owed, exactly as it would overwrite the arrays. The arrays of the last
 * collision (rho, u, force, status) must stay valid, and the force unchanged
 * by others, until then. Readers of rho / u outside this library call
 * lbmi_lb_hydro_sync first (the binding: INTEGRATION.md). */
int lbmi_lb_hydro_sync(lbmi_t * lb);

/* Isothermal fluctuations in lbmi_lb_collide (collision.c:476-518 with
 * lb_fluctuations_var_eta/_bulk/_ghost, _stress, _ghosts, :1745-1920): every
 * fluid site draws from its own generator (noise.c:397-424, 467-487) a random
 * stress of the fluctuation-dissipation variance for temperature kt and,
 * with ghosts_on (lb->param->isghost == LB_GHOST_ON), a random part for each
 * ghost mode; solid sites draw nothing. state is the reference's
 * noise->state on the device: 4 unsigned ints per site, component ia of
 * site i at state[ia*nsites + i] (noise.c:330-366 with ADDR_SOA), read and
 * advanced by every collision. state = NULL: off (the default).
 * D3Q19 only -- the reference's NNOISE_MAX = 10 cannot serve the 17 ghost
 * modes of D3Q27 (noise.h:18). lbmi_lb_collide: every mode but
 * LBMI_MODE_INPLACE (LBMI_ERR_STATE); lbmi_lb_collide_binary
 * (lb_collision_fluctuations, collision.c:884-900, 1663-1745: every site
 * draws, no status test): every mode; lbmi_lb_collide_fe: LBMI_ERR_STATE. */
int lbmi_noise_set(lbmi_t * lb, unsigned int * state, long long nsites,
		   double kt, int ghosts_on);
int lbmi_hydro_field_dirty(lbmi_t * lb, const double * field);

/* ---- rows "next" of the scope table ----------------------- */

/* hydro_u_zero / hydro_f_zero / hydro_rho0 (hydro.c:279-370, kernel
 * hydro_field_set): every site (halo included) of an SoA device field of
 * ncomp = 1..3 components is set to values[]. hydro_u_halo (hydro.c:190)
 * is lbmi_field_halo(lb, 3, u).
 *
 * An array set to zeros here is remembered as holding zeros until a call of
 * this library writes to it (a collision its rho / u, the free-energy calls
 * their force) or the caller reports a write of its own with
 * lbmi_hydro_field_dirty. While it is: setting it to zeros again launches
 * nothing, and lbmi_lb_collide given it as hydro->force does not read it
 * (F = the body force, bit for bit what the reference computes from
 * force_global + 0, collision.c:329-333). In the reference's single-fluid
 * step that is hydro_f_zero every step (ludwig.c:537) on a field nobody
 * writes to: 24 B/site written and 24 B/site read per step, gone.
 * ANY writer of such an array outside this library must call
 * lbmi_hydro_field_dirty before the next lbmi_lb_collide. */
int lbmi_hydro_field_set(lbmi_t * lb, double * field, int ncomp,
			 const double * values);

/* field_halo (field.c:field_halo -> halo_swap_packed) for an SoA field of
 * nel components whose halo swap is nswap layers wide (phi with the
 * symmetric free energy: 2). With slabs the X planes of every layer travel
 * over RCCL, device to device. */
int lbmi_field_halo_n(lbmi_t * lb, int nel, int nswap, double * data);

/* The finite-difference choices of the free-energy sector, as the input
 * keys fd_gradient_calculation (3d_7pt_fluid | 3d_27pt_fluid; gradient_rt.c)
 * and fd_advection_scheme_order (advection_order_set, advection.c:93-97;
 * orders 1..4 of advection_x, advection.c:433-482) select them. Defaults 7
 * and 1, the reference's. They apply to every lbmi_* free-energy call below
 * that evaluates gradients or fluxes itself. Orders 3 and 4 need nhalo >= 2
 * and a 2-layer halo of phi. */
int lbmi_fe_scheme_set(lbmi_t * lb, int grad_npt, int advection_order);

/* field_grad_compute with grad_3d_7pt_fluid_d2 (gradient_3d_7pt_fluid.c:
 * 232-320) / grad_3d_27pt_fluid_d2 (gradient_3d_27pt_fluid.c:85-364): grad
 * (3*nsite, SoA) and delsq (nsite) of the scalar phi for the interior and
 * nhalo-1 layers around it. phi needs a valid halo. lbmi_field_grad uses
 * the stencil of lbmi_fe_scheme_set. */
int lbmi_field_grad_7pt(lbmi_t * lb, const double * phi, double * grad,
			double * delsq);
int lbmi_field_grad_27pt(lbmi_t * lb, const double * phi, double * grad,
			 double * delsq);
int lbmi_field_grad(lbmi_t * lb, const double * phi, double * grad,
		    double * delsq);

/* phi_force_calculation for the symmetric free energy with the stress-
 * divergence method and no walls (phi_force.c:100-108 = pth_stress_compute
 * + pth_force_fluid_driver; stress symmetric.c:371-420): F_a = -d_b P_ab is
 * ADDED to force (hydro->force, 3*nsite) at the interior sites. With grad
 * and delsq given they are used as the reference uses them; with both NULL
 * the force is evaluated straight from phi (needs nhalo >= 2 and a valid
 * 2-layer halo of phi) and no gradient arrays are needed at all. */
int lbmi_symmetric_force(lbmi_t * lb, double a, double b, double kappa,
			 const double * phi, const double * grad,
			 const double * delsq, double * force);

/* phi_cahn_hilliard (phi_cahn_hilliard.c:195-284) for the symmetric free
 * energy without noise, walls or Lees-Edwards planes: advection in u at the
 * order of lbmi_fe_scheme_set (advection.c:433-482), diffusive flux -M grad
 * mu, forward step. phi (valid halo: 1 layer with delsq given, 2 layers with
 * delsq == NULL), u = hydro->u with a valid 1-layer halo (hydro_u_halo);
 * the new interior goes to phi_out (!= phi). The reference updates phi in
 * place through four flux arrays; the caller here swaps the two arrays. */
int lbmi_cahn_hilliard(lbmi_t * lb, double a, double b, double kappa,
		       double mobility, const double * phi,
		       const double * delsq, const double * u,
		       double * phi_out);
/* dst <- src at the interior sites of an SoA device field of ncomp components
 * (halo of dst untouched): puts phi_out back into phi for a caller that, like
 * the reference, updates its order parameter in place. */
int lbmi_field_interior_copy(lbmi_t * lb, int ncomp, const double * src,
			     double * dst);

/* lbmi_symmetric_force (from phi) and lbmi_cahn_hilliard (from phi) in ONE
 * pass: they share the seven (grad, delsq) evaluations around every site.
 * Results identical to the two separate calls. Needs nhalo >= 2.
 * accumulate != 0: force += F (the reference's behaviour after
 * hydro_f_zero); accumulate == 0: force = F at the interior sites, which
 * absorbs hydro_f_zero when nothing else contributes to the force field. */
int lbmi_symmetric_step(lbmi_t * lb, double a, double b, double kappa,
			double mobility, const double * phi, const double * u,
			double * force, double * phi_out, int accumulate);

/* lbmi_symmetric_step on ONE rank with periodic boundaries, without halo
 * swaps of phi and u in front of it: the kernel takes the halo layers of
 * phi from the periodic images and u across a face from the opposite face
 * (what field_halo and hydro_u_halo, ludwig.c:563 and phi_cahn_hilliard.c:
 * 240, would have supplied). Results identical to the halo-swapped form. */
int lbmi_symmetric_step_periodic(lbmi_t * lb, double a, double b,
				 double kappa, double mobility,
				 const double * phi, const double * u,
				 double * force, double * phi_out,
				 int accumulate);

/* One whole time step of the binary fluid with the finite-difference order
 * parameter, as ludwig.c:537-860 runs it for free_energy symmetric:
 *   hydro_f_zero, phi_force_calculation (phi_force.c:74-136),
 *   phi_cahn_hilliard (phi_cahn_hilliard.c:206-284; advects with the u of
 *   the previous collision), lb_collide, lb_halo, lb_propagation
 * on ONE rank with periodic boundaries. phi -> phi_out as in
 * lbmi_symmetric_step; u_prev (3*nsite) = the velocities the previous
 * collision stored, hydro->u (another array) receives those of this one --
 * the caller swaps the two like phi and phi_out -- and hydro->rho the
 * densities (NULL: not stored). hydro->force must be NULL or known to hold
 * zeros (lbmi_hydro_field_set): the thermodynamic force is the only
 * contribution and is not stored anywhere. Neither phi nor u needs a halo.
 * In the steady state of LBMI_MODE_FUSED (D3Q19 or D3Q27, M10 or BGK, 7-point
 * gradients of lbmi_fe_scheme_set, any advection order 1..4, no
 * fluctuations, nlocal >= 4) the step is ONE kernel: the thread that
 * collides a site evaluates its force and its Cahn-Hilliard update from the
 * 25 values of phi around it while its distributions are on their way --
 * 376 B per site and step instead of 424 B in two dependent passes. In any
 * other state or configuration the same results come from the separate calls
 * (identical to rounding, tests/test_gpu_fe.py). Afterwards the handle is
 * where lbmi_lb_propagation leaves it. */
int lbmi_symmetric_lb_step(lbmi_t * lb, const lbmi_hydro_t * hydro,
			   const double * u_prev, double a, double b,
			   double kappa, double mobility, const double * phi,
			   double * phi_out);
/* The same up to and including lb_collide -- hydro_f_zero,
 * phi_force_calculation, phi_cahn_hilliard, lb_collide in one call, ONE kernel
 * under the same conditions -- for a caller whose loop goes on to call lb_halo
 * and lb_propagation itself (ludwig.c does; the binding defers the calls of
 * the free-energy sector up to lb_collide and lands here): afterwards the
 * handle is where lbmi_lb_collide leaves it. */
int lbmi_symmetric_lb_collide(lbmi_t * lb, const lbmi_hydro_t * hydro,
			      const double * u_prev, double a, double b,
			      double kappa, double mobility, const double * phi,
			      double * phi_out);

/* The same single pass, with the gradients taken from the arrays grad and
 * delsq of lbmi_field_grad (valid on the interior and one layer around it)
 * instead of re-evaluated from phi: the cheaper route for the 27-point
 * stencil, where one evaluation costs 27 loads. */
int lbmi_symmetric_step_grad(lbmi_t * lb, double a, double b, double kappa,
			     double mobility, const double * phi,
			     const double * grad, const double * delsq,
			     const double * u, double * force,
			     double * phi_out, int accumulate);

/* The on-disk record stream of the distribution files, lb_io_aggr_pack /
 * lb_io_aggr_unpack with lb_write_buf / lb_read_buf (model.c:1385-1430,
 * 1479-1550): ndist*nvel doubles in [n][p] order per interior site, sites in
 * (ic, jc, kc) order. records: DEVICE buffer of nlocal[X]*nlocal[Y]*
 * nlocal[Z]*ndist*nvel doubles. pack flushes a deferred state first. */
int lbmi_lb_records_pack(lbmi_t * lb, double * records);
int lbmi_lb_records_unpack(lbmi_t * lb, const double * records);

/* The distribution files of the reference's MPI-IO mode with one file
 * (lb_io_write / lb_io_read, model.c:1568-1649; io_impl_mpio.c:179-272):
 *   <dir>/dist-metadata.001-001          JSON, io_metadata_write
 *                                        (io_metadata.c), written once
 *   <dir>/dist-%9.9d.001-001 (timestep)  the record stream of the GLOBAL
 *                                        lattice in (ic, jc, kc) order
 * A rank of an X-slab decomposition owns the contiguous byte range of its
 * x-planes: ntotal_x = global extent in X, offset_x = global index of this
 * rank's first plane (single rank: nlocal[X], 0). Every rank writes / reads
 * its range with pwrite / pread (no MPI needed); the rank with offset_x == 0
 * writes the metadata. The records pass through fprime (dead at that point
 * of a step) and a pinned staging buffer: no extra device memory.
 * lbmi_lb_io_read replaces the state (nothing stays pending). */
int lbmi_lb_io_write(lbmi_t * lb, const char * dir, int timestep,
		     int ntotal_x, int offset_x);
/* The record format of lbmi_lb_io_write / lbmi_lb_io_read: 0 (the default)
 * binary records of ndist*nvel doubles; 1 text records, the input key
 * distribution_io_format ascii (lb_write_buf_ascii / lb_read_buf_ascii,
 * model.c:1438-1490): per site nvel lines, line p = the ndist values f(n, p)
 * as " %22.15e", and metadata that says MPI_CHAR x nvel*(ndist*23 + 1). The
 * records are still packed on the device; the text is made on the host. */
int lbmi_io_format_set(lbmi_t * lb, int fmt);
/* fmt: 0, LBMI_IO_ASCII, or LBMI_IO_SINGLE = the old-style i/o a run gets
 * when its input names no i/o mode (io_options_default(): IO_MODE_SINGLE;
 * lb_io_write, model.c:1583-1587 -> io_write_data_s, io_harness.c; the
 * io_info_t path). Binary records only, the same record stream in
 *   <dir>/dist-%8.8d.001-001 (timestep)
 * with metadata that says "single", and the text file the old style keeps,
 *   <dir>/dist.001-001.meta   (io_write_metadata_file, io_harness.c:369-466),
 * written by the rank at offset_x == 0 with one line per rank, the slabs cut
 * as cs_init cuts them (coords.c:577-592). lb_io_read of that mode seeks to
 * the global position of every row (io_file_offset, single_file_read): the
 * same byte range per X slab. Text records in this mode have no fixed size in
 * the reference and are refused (LBMI_ERR_UNSUPPORTED). */
enum {LBMI_IO_ASCII = 1, LBMI_IO_SINGLE = 2};

/* Several files: the reference's i/o grid (input key distribution_io_grid,
 * io_subfile_create, io_subfile.c:49-91) cut along the slab direction,
 * {nfile, 1, 1}. The ranks of a group write ONE file that holds the group's
 * block of planes in the same record order: file `index` of `nfile` is
 *   <dir>/dist-%9.9d.%3.3d-%3.3d   (timestep, 1 + index, nfile)
 * with its own <dir>/dist-metadata.%3.3d-%3.3d, written by the rank whose
 * offset_x is file_x0. A rank's byte range starts at its planes' position in
 * the FILE (io_impl_mpio.c:179-272). periodic: what cs_periodic says, printed
 * in the metadata (1, 1, 1 when nothing is set). MPI-IO mode only.
 * Parity: the one-file case is pinned by the reference's files; no serial run
 * of the reference can write several, so for nfile > 1 the keys and the
 * arithmetic follow the code cited and the tests check them against the
 * one-file stream (cut at the file boundaries) -- unpinned. */
typedef struct lbmi_io_file_s {
  int nfile;         /* files in all (the i/o grid's X extent)                */
  int index;         /* this rank's file, 0 .. nfile-1                        */
  int file_nx;       /* planes in that file (io_subfile_t::sizes[X])          */
  int file_x0;       /* global index of its first plane (::offset[X])         */
  int periodic[3];   /* cs_periodic, for the metadata                         */
} lbmi_io_file_t;
/* file == NULL: back to one file of the whole lattice, periodic */
int lbmi_io_file_set(lbmi_t * lb, const lbmi_io_file_t * file);
int lbmi_lb_io_read(lbmi_t * lb, const char * dir, int timestep,
		    int ntotal_x, int offset_x);

/* Host-only helpers (no device needed): the metadata file for a lattice of
 * ntotal sites with nel = ndist*nvel doubles per record, and the data file
 * name of a time step (io_subfile_name, io_subfile.c). */
int lbmi_io_metadata_write(const char * dir, const char * stub, int nel,
			   const int ntotal[3]);
/* the same for records of ndist*nvel values in any format (fmt as for
 * lbmi_io_format_set) */
int lbmi_io_metadata_write_fmt(const char * dir, const char * stub, int nvel,
			       int ndist, const int ntotal[3], int fmt);
/* ... and of one file of several (file == NULL: the one file) */
int lbmi_io_metadata_write_file(const char * dir, const char * stub, int nvel,
				int ndist, const int ntotal[3], int fmt,
				const lbmi_io_file_t * file);
int lbmi_io_filename(const char * dir, const char * stub, int timestep,
		     char * buf, size_t bufsz);
int lbmi_io_filename_fmt(const char * dir, const char * stub, int timestep,
			 int fmt, char * buf, size_t bufsz);
/* <dir>/<stub>.001-001.meta of the single mode: cartsz slabs along cartdim,
 * nslab[r] planes in slab r (NULL for one rank) */
int lbmi_io_single_metadata_write(const char * dir, const char * stub, int nvel,
				  int ndist, const int ntotal[3], int cartdim,
				  int cartsz, const int * nslab);

/* ---- streams / synchronisation ----------------------------------------- */

int lbmi_synchronize(lbmi_t * lb);
/* hipStream_t of the compute stream, as void* */
int lbmi_stream(lbmi_t * lb, void ** stream);
/* Run all further work of this handle on the caller's hipStream_t (NULL =
 * the default stream) instead of the handle's own non-blocking stream, so
 * that it is ordered with the caller's other work on that stream (e.g.
 * Ludwig's default-stream kernels, or a torch stream). The handle does not
 * own the stream. Pending work on the previous stream is drained first. */
int lbmi_set_stream(lbmi_t * lb, void * stream);
/* Device time (ms) of the step's kernel launches since the last call,
 * measured with hipEvents on the compute stream (bench.py roofline).
 * lbmi_timing(lb, k): k = 0 off, 1 every launch, k > 1 every k-th launch (an
 * event record costs the stream a few microseconds between two kernels, so
 * a timed production loop samples). lbmi_timing_read returns the summed
 * time and the number of launches that were timed. */
int lbmi_timing(lbmi_t * lb, int on);
int lbmi_timing_read(lbmi_t * lb, double * ms_total, int * nlaunch);
/* The sampled steps of a slab (the first 256 of them) in detail: average
 * milliseconds of ms[0] the interior launch, ms[1] the exchange (pack if any,
 * messages, unpack if any), ms[2] the boundary launch, each measured on the
 * stream it runs on. Resets the samples. */
int lbmi_timing_read_detail(lbmi_t * lb, double ms[3], int * nsample);

/* Launch tuning of the fused step; results do not depend on it.
 * "xcd_group": blocks per XCD interleave group (0 = one chunk per XCD; 32);
 * "lds_cap": dynamic LDS bytes per block, caps resident blocks per CU (65536);
 * "blocked": 1 = on one GPU keep the deferred FUSED state in the blocked
 *            order [site/256][p][site%256] (default 1), 0 = SoA throughout;
 * "nt_store": bit 0 = nontemporal stores of the blocked deferred state, bit 1
 *            = of hydro->rho, u; -1 (default) = bit 0 exactly when f and
 *            fprime together exceed the 256 MiB Infinity Cache;
 * "x_packed": 1 = RCCL X exchange through packed staging buffers, one message
 *            per direction (default), 0 = zero-copy sends of the planes;
 * "x_direct": 1 = slabs, FUSED: the boundary launch takes the populations
 *            that cross the X faces straight from the receive buffers and
 *            leaves those the next exchange sends in the send buffers: no pack
 *            and no unpack kernel in a steady-state step (default; needs
 *            x_packed), 0 = pack, messages, unpack into the halo planes;
 * "eager_oop": 1 = LBMI_MODE_EAGER: lbmi_lb_collide writes the other array and
 *            swaps the two (default: in-place read-modify-write is the slower
 *            way to move the same bytes, 1.22 against 1.0 ms at 256^3), 0 = in
 *            place as the reference;
 * "x_concurrent": 1 = slabs: the two boundary planes run on a third stream
 *            beside the interior launch once the halo has arrived (default),
 *            0 = after it on the compute stream;
 * "hydro_lazy": 1 = lbmi_lb_collide leaves hydro->rho, u to lbmi_lb_hydro_sync
 *            (see there), 2 = hydro->rho only (u stored: somebody reads it
 *            before the next collision; lbmi_symmetric_lb_step treats 1 as
 *            2), 0 = both stored by every collision (default);
 * "halo_fold": 1 = LBMI_MODE_FUSED_HALO on one rank: the kernel of lbmi_lb_collide
 *            computes the width-1 halo shell of its own result (a shell site =
 *            the collision of its periodic image), and the lbmi_lb_halo that
 *            follows has nothing left to do unless f has been written in
 *            between (default), 0 = three halo launches after it;
 * "fe_xcd_group" (8), "fe_stripes" (0): block-to-XCD mapping of the one-kernel
 *            binary-fluid step (lbmi_symmetric_lb_step);
 * "graph":   1 = lbmi_lb_run on one GPU in FUSED mode issues its steps as
 *            launches of ONE hipGraph holding two steady-state steps (for
 *            lattices whose step is as short as a launch), 0 = step by step
 *            (default). */
int lbmi_tune(lbmi_t * lb, const char * key, int value);

/* ---- multi-GPU: 1-d slab decomposition (lbmi_options_t::cartdim) over RCCL -- */

#define LBMI_UNIQUE_ID_BYTES 128
/* ncclGetUniqueId: HOST buffer of LBMI_UNIQUE_ID_BYTES, call on rank 0 and
 * distribute by any out-of-band means (MPI_Bcast, torch.distributed store) */
int lbmi_comm_unique_id(void * id);
/* ncclCommInitRank over the cartsz ranks given at lbmi_create() */
int lbmi_comm_init(lbmi_t * lb, const void * id);
int lbmi_comm_free(lbmi_t * lb);
/* What the handle exchanges over: the number of ranks of its ring, its own
 * rank there, and the transport (0 none, 1 RCCL, 2 the peer ring of one process) */
int lbmi_comm_info(lbmi_t * lb, int * nranks, int * rank, int * transport);

/* The point-to-point operations of ONE X exchange of a rank, in the order
 * they are issued -- what replaces halo_swap.c:762-881 (MPI_Irecv / pack /
 * copy to the host / MPI_Isend / MPI_Waitall / copy back / unpack). A pure
 * host function of the decomposition (no device, no handle): the product
 * executes exactly this list with ncclSend / ncclRecv inside one group, and
 * a test can execute it with any other transport that, like RCCL, matches
 * the k-th send towards a peer with the k-th receive from it.
 *   packed != 0: one message per direction through staging buffers laid out
 *     [k][plane site] (k-th component of the selection, strx doubles each):
 *     SENDHI = last interior plane, components that fill a LOW halo
 *     (reduced: c_x = +1) -> next rank's RECVLO; SENDLO = first interior
 *     plane, components that fill a HIGH halo -> previous rank's RECVHI.
 *   packed == 0: one message per component straight between the planes of
 *     the array (buffer LBMI_XBUF_DATA, offset in doubles from its start). */
enum {LBMI_XOP_SEND = 0, LBMI_XOP_RECV = 1};
enum {LBMI_XBUF_SENDLO = 0, LBMI_XBUF_SENDHI = 1, LBMI_XBUF_RECVLO = 2,
      LBMI_XBUF_RECVHI = 3, LBMI_XBUF_DATA = 4};
typedef struct lbmi_xop_s {
  int kind;                 /* LBMI_XOP_SEND | LBMI_XOP_RECV */
  int peer;                 /* rank along X */
  int buffer;               /* LBMI_XBUF_* */
  long long offset;         /* doubles from the start of that buffer */
  long long count;          /* doubles */
} lbmi_xop_t;
int lbmi_x_schedule(const lbmi_options_t * opts, int scheme, int packed,
		    lbmi_xop_t * ops, int maxops, int * nops);
/* the same for the exchange along direction dim of a decomposition of any
 * kind (cartdim 3: once per decomposed direction, in the order X, Y, Z, with
 * the local periodic copies of the other directions in their places) */
int lbmi_x_schedule_dim(const lbmi_options_t * opts, int dim, int scheme,
			int packed, lbmi_xop_t * ops, int maxops, int * nops);

/* The peer transport: a ring of cartsz handles inside ONE process, each
 * driven by a host thread of its own and each on a device of its own -- one
 * process steering the GPUs of a node, the planes travelling as peer-to-peer
 * copies over xGMI (hipMemcpyPeerAsync; the receiver pulls) -- or several of
 * them on the same device. The slab path is the one that runs over RCCL:
 * kernels, schedule (lbmi_x_schedule), streams, overlap; only the execution
 * of the schedule differs (posts into per-pair FIFOs, event waits, copies).
 * It needs no RCCL bootstrap and no out-of-band exchange of an id
 * (bench.py --transport peer), and on ONE GPU it is the rehearsal of the
 * N-rank step that RCCL does not allow there ("Duplicate GPU detected").
 * lbmi_ring_create once, lbmi_comm_init_ring on every handle in place of
 * lbmi_comm_init (it enables peer access to the neighbours' devices),
 * lbmi_comm_free / lbmi_free on every handle, lbmi_ring_free. An exchange
 * waits for the neighbours' posts: with more than one rank the handles must
 * step concurrently (one thread each), each on its own stream. */
int lbmi_ring_create(int nranks, lbmi_ring_t ** ring);
int lbmi_comm_init_ring(lbmi_t * lb, lbmi_ring_t * ring);
int lbmi_ring_free(lbmi_ring_t * ring);
/* A rank whose driver has failed: exchanges of the others stop waiting and
 * return LBMI_ERR_STATE from now on. */
int lbmi_ring_abort(lbmi_ring_t * ring);

#ifdef __cplusplus
}
#endif

#endif
