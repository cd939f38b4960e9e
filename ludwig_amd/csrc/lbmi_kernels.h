/*
 * lbmi_kernels.h -- internal interface between the ANSI C host code
 * (lbmi_host.c) and the HIP kernels (lbmi_kernels.hip). Not installed.
 */

#ifndef LBMI_KERNELS_H
#define LBMI_KERNELS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LBMI_NVEL_MAX 27

/* Kernel parameter block: passed BY VALUE as a kernel argument (scalar
 * registers / kernarg segment), never through __constant__ copies. */

typedef struct lbmi_kparam_s {
  int nvel;
  int scheme;              /* lbmi_relaxation_t */
  int nlocal[3];
  int nhalo;
  int nall[3];
  int strx;                /* nall[Y]*nall[Z] */
  int stry;                /* nall[Z] */
  long long nsite;
  double rstrx;            /* (1/strx)(1 + 2^-40): exact index decode */
  double rstry;
  double rtau_shear;
  double rtau_bulk;
  double rtau_even;        /* ghost modes 10, 14, 18 (d3q19); all for d3q27 */
  double rtau_odd;         /* ghost modes 11-13, 15-17 (d3q19) */
  double rho0;             /* for local relaxation times (hydro eta) */
  double bulk_ratio;       /* eta_bulk/eta_shear of lbmi_set_relaxation */
  double fbody[3];
  /* launch tuning of the fused kernel (lbmi_tune) */
  int xcd_group;           /* blocks per XCD interleave group; 0: chunked */
  int lds_cap;             /* dynamic LDS bytes per block: occupancy cap */
  int nt_store;            /* blocked order: nontemporal stores of f */
  int fe_tiled;            /* free-energy pass: phi through an LDS tile */
  int fe_xcd_group;        /* k_symm_lb_step: blocks per XCD interleave group
			      (8: a little better than the 32 of the LB kernel
			      in every A/B, profiles/r03_rejected.txt 6) */
  int fe_stripes;          /* k_symm_lb_step: every XCD an eighth of every
			      x plane (its stencil partners behind one L2);
			      measured slower, default 0 */
} lbmi_kparam_t;

typedef struct lbmi_hydro_dev_s {
  const double * force;
  const char   * status;
  double       * rho;
  double       * u;
  const double * eta;       /* local shear viscosity, or NULL */
  /* isothermal fluctuations: the reference's generator state (noise->state,
   * 4 unsigned ints per site, component ia of site i at [ia*stride + i]),
   * or NULL = off */
  long long      stride;    /* components of force and u: 0 = kp.nsite */
  long long      gstride;   /* components of grad phi (collisions that read
			       it): 0 = kp.nsite */
  unsigned int * noise;
  long long      noise_stride;
  double         noise_kt;
  int            noise_ghosts;
} lbmi_hydro_dev_t;

/* Halo pass description: components (populations) to copy to the low-side
 * and to the high-side halo plane of direction dir. */

typedef struct lbmi_halo_sel_s {
  int nlo;
  int nhi;
  int8_t lo[LBMI_NVEL_MAX];   /* components wanted in the low halo plane  */
  int8_t hi[LBMI_NVEL_MAX];   /* components wanted in the high halo plane */
} lbmi_halo_sel_t;

/* The buffers of the X exchange of a slab, as the boundary launch of a FUSED
 * step uses them: [k][plane site] of strx doubles per component, k-th
 * population with c_x = +1 (recvlo, sendhi) or c_x = -1 (recvhi, sendlo) */

typedef struct lbmi_xbuf_s {
  const double * recvlo;     /* from the lower neighbour: its last interior plane */
  const double * recvhi;     /* from the upper neighbour: its first interior plane */
  double * sendlo;           /* our first interior plane -> lower neighbour */
  double * sendhi;           /* our last interior plane -> upper neighbour */
} lbmi_xbuf_t;

/* All launchers return hipError_t as int (0 = hipSuccess) */

int lbmi_k_collide(const lbmi_kparam_t * kp, double * f,
		   const lbmi_hydro_dev_t * h, void * stream);
/* out of place over the whole array (sites that do not collide are copied); no fluctuations */
int lbmi_k_collide_to(const lbmi_kparam_t * kp, const double * f, double * fo,
		      const lbmi_hydro_dev_t * h, void * stream);

int lbmi_k_propagate(const lbmi_kparam_t * kp, const double * f,
		     double * fprime, void * stream);

/* xlo..xhi and xlo2..xhi2: 0-based x planes (in nall) to process,
 * inclusive, in ONE launch; an empty range has xhi < xlo.
 * wrapmask: bit d set = wrap direction d by index arithmetic. */
/* lay: 0 SoA -> SoA; 1 SoA -> blocked; 2 blocked -> blocked (the blocked
 * order [site/256][p][site%256] needs wrapmask != 0 in every direction that
 * is pulled across: single GPU).
 * xb != NULL (slabs; the planes launched are the first and/or last interior
 * plane): populations that cross the X faces come from xb->recvlo / recvhi
 * instead of the halo planes of f, and those the next exchange sends are
 * stored into xb->sendlo / sendhi as well. */
int lbmi_k_propagate_collide(const lbmi_kparam_t * kp, const double * f,
			     double * fprime, const lbmi_hydro_dev_t * h,
			     int wrapmask, int lay, int xlo, int xhi, int xlo2,
			     int xhi2, const lbmi_xbuf_t * xb, void * stream);
/* The FUSED_HALO step of one rank, SoA -> SoA, pulling from f with its halo
 * as it is, AND the width-1 halo shell of fprime computed on the way (every
 * shell site = the collision of its periodic image): what three
 * lbmi_k_halo_copy launches on fprime would deliver. No fluctuations. */
int lbmi_k_propagate_collide_halo(const lbmi_kparam_t * kp, const double * f,
				  double * fprime, const lbmi_hydro_dev_t * h,
				  void * stream);
/* The first and the last plane of a slab along dim = 1 (Y) or 2 (Z), after
 * lbmi_k_propagate_collide has run over everything (its results in these two
 * planes are overwritten): face-crossing populations from xb->recvlo / recvhi,
 * the rest pulled from f (wrapmask: the two local directions), SoA -> SoA;
 * fills xb->sendlo / sendhi for the next exchange. No fluctuations. */
int lbmi_k_propagate_collide_face(const lbmi_kparam_t * kp, int dim,
				  const double * f, double * fprime,
				  const lbmi_hydro_dev_t * h, int wrapmask,
				  const lbmi_xbuf_t * xb, void * stream);
int lbmi_k_blocked_sites(const lbmi_kparam_t * kp);
/* rho, u of the collision that left the post-collision state f (SoA, or the
 * blocked order): u = (sum f'_p c_p - F/2)/rho at interior fluid sites */
int lbmi_k_hydro_from_f(const lbmi_kparam_t * kp, const double * f,
			const lbmi_hydro_dev_t * h, int blocked, void * stream);
int lbmi_k_relayout(const lbmi_kparam_t * kp, const double * src,
		    double * dst, int to_blocked, void * stream);

/* In-place streaming (AA pattern), single GPU, all directions wrapped by
 * index: even = collide in place into swapped slots; odd = pull (from the
 * swapped or the normal layout) + collide + push into the normal layout. */
int lbmi_k_aa_even(const lbmi_kparam_t * kp, double * f,
		   const lbmi_hydro_dev_t * h, void * stream);
int lbmi_k_aa_odd(const lbmi_kparam_t * kp, double * f,
		  const lbmi_hydro_dev_t * h, int wrapmask, int swapped_in,
		  void * stream);
int lbmi_k_aa_unswap(const lbmi_kparam_t * kp, double * f, void * stream);
int lbmi_k_unpropagate_wrap(const lbmi_kparam_t * kp, const double * f,
			    double * fprime, int wrapmask, void * stream);

/* In-place periodic halo copy for direction dir on an SoA field with
 * components of stride nsite. */
int lbmi_k_halo_copy(const lbmi_kparam_t * kp, int dir,
		     const lbmi_halo_sel_t * sel, double * data, int nswap,
		     void * stream);

/* X direction through buffers (multi-GPU): pack the first interior plane
 * (components sel->hi: wanted by the lower neighbour's HIGH halo) into
 * buf_lo and the last interior plane (components sel->lo) into buf_hi;
 * unpack the reverse way. Buffer layout: [component k][plane site].
 * blocked != 0: data is a distribution array in the blocked order.
 * layer: 0 for a width-1 swap; l for the (l+1)-th plane of a wider one. */
/* The same for slabs along direction dir (0 X, 1 Y, 2 Z): the planes of Y and
 * Z slabs are gathered (rows of nall[Z] values, or single values nall[Z]
 * apart); buffer layout [component k][plane site], plane sites in the order
 * of the two remaining coordinates, the faster one fastest. */
int lbmi_k_halo_pack(const lbmi_kparam_t * kp, int dir,
		     const lbmi_halo_sel_t * sel, const double * data,
		     double * buf_lo, double * buf_hi, int blocked, int layer,
		     void * stream);
int lbmi_k_halo_unpack(const lbmi_kparam_t * kp, int dir,
		       const lbmi_halo_sel_t * sel, double * data,
		       const double * buf_lo, const double * buf_hi, int blocked,
		       int layer, void * stream);
int lbmi_k_halo_pack_x(const lbmi_kparam_t * kp, const lbmi_halo_sel_t * sel,
		       const double * data, double * buf_lo, double * buf_hi,
		       int blocked, int layer, void * stream);
int lbmi_k_halo_unpack_x(const lbmi_kparam_t * kp, const lbmi_halo_sel_t * sel,
			 double * data, const double * buf_lo,
			 const double * buf_hi, int blocked, int layer,
			 void * stream);

/* Record stream of the distribution files: pack != 0: f -> rec, else
 * rec -> f (interior sites only). rec: ninterior*ndist*nvel doubles (device),
 * the record of a site is [n][p]. */
int lbmi_k_records(const lbmi_kparam_t * kp, int ndist, double * f,
		   double * rec, int pack, void * stream);

/* hydro_field_set: all nsite sites of ncomp (1..3) components := v[] */
int lbmi_k_field_set(const lbmi_kparam_t * kp, int ncomp, double * field,
		     const double * v, void * stream);

/* Symmetric free energy (row f2): 7-point gradients; thermodynamic force by
 * stress divergence, from grad/delsq arrays or (grad == NULL) from phi */
/* npt: 7 | 27 point gradient stencil (0 in lbmi_k_symm_fe_step: use the
 * arrays grad, delsq); order: advection scheme order 1..4; wrap != 0 (npt 7
 * or 27): the kernel wraps the periodic box by index, phi and u need no halo */
int lbmi_k_grad(const lbmi_kparam_t * kp, int npt, const double * phi,
		double * grad, double * delsq, void * stream);
int lbmi_k_symm_force(const lbmi_kparam_t * kp, int npt, double a, double b,
		      double kappa, const double * phi, const double * grad,
		      const double * delsq, double * force, void * stream);

int lbmi_k_cahn_hilliard(const lbmi_kparam_t * kp, int npt, int order,
			 double a, double b, double kappa, double mobility,
			 const double * phi, const double * delsq,
			 const double * u, double * phi_out, void * stream);

int lbmi_k_symm_fe_step(const lbmi_kparam_t * kp, int npt, int order,
			double a, double b, double kappa, double mobility,
			const double * phi, const double * grad,
			const double * delsq, const double * u, double * force,
			double * phi_out, int accumulate, int wrap,
			void * stream);

/* The whole binary-fluid step (symmetric free energy, 7-point gradients,
 * D3Q19, M10 or BGK) in one pass on one rank: thermodynamic force and
 * Cahn-Hilliard update of the site from phi and uprev (u of the previous
 * collision), pull + collision with that force; rho, u of this collision to
 * h->rho, h->u (h->u != uprev); h->force is not used. lay as
 * lbmi_k_propagate_collide. */
int lbmi_k_symm_lb_step(const lbmi_kparam_t * kp, const double * f,
			double * fprime, const lbmi_hydro_dev_t * h,
			double a, double b, double kappa, double mobility,
			int order, const double * phi, const double * uprev,
			double * phi_out, int lay, void * stream);

/* k_collide with fe->use_stress_relaxation for the symmetric free energy */
int lbmi_k_collide_fe(const lbmi_kparam_t * kp, double * f,
		      const lbmi_hydro_dev_t * h, double a, double b,
		      double kappa, const double * phi, const double * grad,
		      const double * delsq, void * stream);
/* propagation fused with that collision, f -> fp, SoA; wrapmask: directions
 * wrapped by index (0: pull from the halo as it is) */
int lbmi_k_propagate_collide_fe(const lbmi_kparam_t * kp, const double * f,
				double * fp, const lbmi_hydro_dev_t * h,
				double a, double b, double kappa,
				const double * phi, const double * grad,
				const double * delsq, int wrapmask,
				void * stream);

/* Two distributions (symmetric_lb): f2[(n*nvel + p)*nsite + i] */
/* pull != 0 / src != f2: a propagation is pending on the array read:
 * populations come from i - c_p (the array has its halo, wrapmask 0) or, in
 * the directions of wrapmask, from the periodic image inside the domain (the
 * halo swap is pending as well); results go to f2 */
/* blocked != 0 (pull only): f2 is a deferred state in the blocked order of
 * two distributions, [site/256][n*nvel + p][site%256]; lay of
 * lbmi_k_collide_binary as in lbmi_k_propagate_collide (0 SoA -> SoA, 1 SoA ->
 * blocked, 2 blocked -> blocked; nontemporal stores with kp->nt_store & 1) */
int lbmi_k_phi_from_g(const lbmi_kparam_t * kp, const double * f2,
		      double * phi, int pull, int wrapmask, int blocked,
		      void * stream);
int lbmi_k_collide_binary(const lbmi_kparam_t * kp, const double * src,
			  double * f2,
			  const lbmi_hydro_dev_t * h, double a, double b,
			  double kappa, double rtau2, const double * phi,
			  const double * grad, const double * delsq,
			  int wrapmask, int lay, void * stream);
/* lbmi_k_relayout for a state of ndist distributions (ndist*nvel components) */
int lbmi_k_relayout_n(const lbmi_kparam_t * kp, int ndist, const double * src,
		      double * dst, int to_blocked, void * stream);

/* Bounce-back on links (wall_bbl_kernel, wall.c:996-1107). Tables travel by
 * value; part: nblk*3 doubles of per-block momentum, added to fnet[3] (device)
 * by a one-thread epilogue in block order (deterministic). status (or NULL):
 * links whose fluid site is MAP_COLLOID only enter the accounting. err: one
 * int the host can read (pinned, mapped): a record that would address outside
 * f is skipped and its index + 1 stored there. */
typedef struct lbmi_wall_tab_s {
  int nvel;
  int ndist;
  int8_t cv[LBMI_NVEL_MAX][3];
  double wv[LBMI_NVEL_MAX];
  double rho0;
  double uw[3][3];              /* WALL_UZERO, WALL_UWTOP, WALL_UWBOT */
  double slip[19];              /* slip fraction by wall_slip_enum_t */
} lbmi_wall_tab_t;
int lbmi_k_wall_nblk(int nlink);
int lbmi_k_wall_bbl(const lbmi_kparam_t * kp, const lbmi_wall_tab_t * tab,
		    double * f, int nlink, const int * linki,
		    const int * linkj, const int * linkp, const int * linku,
		    const char * status, double * part, double * fnet,
		    int * err, void * stream);

/* wall_bbl_slip_kernel (wall.c:1118-1205): linkk, linkq, links as the
 * reference keeps them (int, int8_t, int8_t) */
int lbmi_k_wall_bbl_slip(const lbmi_kparam_t * kp, const lbmi_wall_tab_t * tab,
			 double * f, int nlink, const int * linki,
			 const int * linkj, const int * linkp,
			 const int * linkk, const int8_t * linkq,
			 const int8_t * links, const char * status,
			 double * part, double * fnet, int * err,
			 void * stream);

/* *slot_f = f, *slot_fprime = fprime on the device, in stream order */
int lbmi_k_store_pointers(double ** slot_f, double ** slot_fprime, double * f,
			  double * fprime, void * stream);

/* dst <- src at the interior sites of an SoA field of ncomp components */
int lbmi_k_interior_copy(const lbmi_kparam_t * kp, int ncomp,
			 const double * src, double * dst, void * stream);

/* rho = sum_p f_p (p order) of the interior sites, dense, (ic, jc, kc) order */
int lbmi_k_density(const lbmi_kparam_t * kp, const double * f, double * rho,
		   void * stream);

/* Moments: partial (nblk x 12 doubles workspace) then final (out_dev[9]) */
int lbmi_k_moments_nblk(void);
int lbmi_k_moments(const lbmi_kparam_t * kp, const double * f,
		   const char * status, double * work, double * out_dev,
		   void * stream);

/* Statistics of a scalar field over interior fluid sites, through the
 * workspace and the final stage of the moments: out_dev[0] volume, [2] sum of
 * squares, [3] min, [4] max, [5] Kahan-compensated sum */
int lbmi_k_field_stats(const lbmi_kparam_t * kp, const double * field,
		       const char * status, double * work, double * out_dev,
		       void * stream);

/* Host model tables (same constexpr source as the device code) */
int lbmi_k_model(int nvel, int8_t * cv, double * wv, double * na, double * ma);

#ifdef __cplusplus
}
#endif

#endif
