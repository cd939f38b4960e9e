/*****************************************************************************
 *
 *  lbmi_kernels.hip
 *
 *  Hand-written HIP kernels for gfx950 (MI355X, CDNA4) for the lattice-
 *  Boltzmann hot path of the reference (zazu29/ludwig v0.20.1):
 *
 *    k_collide            lb_collision_mrt1[_site]   collision.c:223-599
 *    k_propagate          lb_propagation_kernel      propagation.c:162-212
 *    k_propagate_collide  the two above fused: propagation(t)+collision(t+1)
 *    k_halo_copy/pack/unpack  halo_swap_pack_rank1 / unpack_rank1 and the
 *                         periodic self-wrap          halo_swap.c:709-1274
 *    k_moments_*          stats_distribution_print + distribution_gm_kernel
 *                                              stats_distribution.c:55-350
 *
 *  Design: the step is an HBM-bandwidth-bound 19/27-point
 *  stencil in FP64. One lattice site per lane, 64-wide wavefronts along the
 *  contiguous z direction of the reference's SoA layout, so that every
 *  population is one fully coalesced 512-byte access per wave. No MFMA (not
 *  a contraction), no LDS staging of f (each population value is consumed
 *  by exactly one site: there is no reuse for LDS to capture). The velocity
 *  set, weights and the mode matrix are compile-time constants folded into
 *  the instruction stream; run-time parameters travel as a by-value kernel
 *  argument. Blocks are remapped so that each XCD (private L2) owns one
 *  contiguous chunk of the lattice.
 *
 *****************************************************************************/

#include <hip/hip_runtime.h>
#include <type_traits>

#include "lbmi_kernels.h"

namespace {

enum {LBMI_M10 = 0, LBMI_BGK = 1, LBMI_TRT = 2};

/* Tunables (defaults are the measured best; the -D overrides
 * exist for the ablation runs recorded under profiles/) */
#ifndef LBMI_BLOCK
#define LBMI_BLOCK 256
#endif
#ifndef LBMI_XCD_REMAP
#define LBMI_XCD_REMAP 1
#endif
#ifndef LBMI_NT
#define LBMI_NT 0           /* bit 0: nontemporal loads, bit 1: stores */
#endif
#ifndef LBMI_ABL_NOCOLLIDE
#define LBMI_ABL_NOCOLLIDE 0
#endif
#ifndef LBMI_HALO_LANES_COPY
#define LBMI_HALO_LANES_COPY 0   /* 1: halo lanes copy in place (reference) */
#endif
#ifndef LBMI_ALIGN
#define LBMI_ALIGN 16       /* block starts at multiples of this many sites */
#endif
#ifndef LBMI_SPT
#define LBMI_SPT 1          /* sites per thread in the fused kernel */
#endif
#ifndef LBMI_WAVES
#define LBMI_WAVES 1        /* __launch_bounds__ min waves per SIMD */
#endif
#ifndef LBMI_FAST_DECODE
#define LBMI_FAST_DECODE 1
#endif
enum {BLOCK = LBMI_BLOCK, SPT = LBMI_SPT};

/* streaming accesses of f: optionally with the nontemporal hint */
__device__ __forceinline__ double ldf(const double * p) {
#if LBMI_NT & 1
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}
__device__ __forceinline__ void stf(double * p, double v) {
#if LBMI_NT & 4
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#elif LBMI_NT & 2
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}

/* ---- compile-time model description -------------------------------------
 *
 * Velocity sets and weights: lb_d3q19.h:26-38, lb_d3q27.h:28-43.
 * Mode matrix ma: lb_d3q19.c:107-153, lb_d3q27.c:150-195.
 * Normalisers na[m] = 1/sum_p w_p ma[m][p]^2: lb_d3q19.c:69-77.
 * Inverse mi[p][m] = w_p na[m] ma[m][p]: model.c:381-387.
 */

template <int NVEL> struct VSet;

template <> struct VSet<19> {
  static constexpr int cv_[19][3] = {
    { 0, 0, 0},
    { 1, 1, 0}, { 1, 0, 1}, { 1, 0, 0}, { 1, 0,-1}, { 1,-1, 0},
    { 0, 1, 1}, { 0, 1, 0}, { 0, 1,-1}, { 0, 0, 1}, { 0, 0,-1},
    { 0,-1, 1}, { 0,-1, 0}, { 0,-1,-1},
    {-1, 1, 0}, {-1, 0, 1}, {-1, 0, 0}, {-1, 0,-1}, {-1,-1, 0}};
  __host__ __device__ static constexpr int c(int p, int a) { return cv_[p][a]; }
  __host__ __device__ static constexpr double w(int p) {
    int c2 = c(p,0)*c(p,0) + c(p,1)*c(p,1) + c(p,2)*c(p,2);
    return (c2 == 0) ? 12.0/36.0 : ((c2 == 1) ? 2.0/36.0 : 1.0/36.0);
  }
  __host__ __device__ static constexpr double ma(int m, int p) {
    double cx = c(p,0), cy = c(p,1), cz = c(p,2);
    double cs2 = 1.0/3.0;
    double c2 = cx*cx + cy*cy + cz*cz;
    double chi1 = (2.0*c2 - 3.0)*(3.0*cz*cz - c2);
    double chi2 = (2.0*c2 - 3.0)*(cy*cy - cx*cx);
    double chi3 = 3.0*c2*c2 - 6.0*c2 + 1;
    switch (m) {
    case  0: return 1.0;
    case  1: return cx;
    case  2: return cy;
    case  3: return cz;
    case  4: return cx*cx - cs2;
    case  5: return cx*cy;
    case  6: return cx*cz;
    case  7: return cy*cy - cs2;
    case  8: return cy*cz;
    case  9: return cz*cz - cs2;
    case 10: return chi1;
    case 11: return chi1*cx;
    case 12: return chi1*cy;
    case 13: return chi1*cz;
    case 14: return chi2;
    case 15: return chi2*cx;
    case 16: return chi2*cy;
    case 17: return chi2*cz;
    default: return chi3;
    }
  }
};

template <> struct VSet<27> {
  /* p = 0 rest; then x slowest, z fastest from (-1,-1,-1), rest skipped */
  __host__ __device__ static constexpr int c(int p, int a) {
    if (p == 0) return 0;
    int q = p - 1;
    if (q >= 13) q += 1;
    return (a == 0) ? (q/9 - 1) : ((a == 1) ? ((q/3) % 3 - 1) : (q % 3 - 1));
  }
  __host__ __device__ static constexpr double w(int p) {
    int c2 = c(p,0)*c(p,0) + c(p,1)*c(p,1) + c(p,2)*c(p,2);
    return (c2 == 0) ? 64.0/216.0 : ((c2 == 1) ? 16.0/216.0 :
				     ((c2 == 2) ? 4.0/216.0 : 1.0/216.0));
  }
  __host__ __device__ static constexpr double ma(int m, int p) {
    double cx = c(p,0), cy = c(p,1), cz = c(p,2);
    double cs2 = 1.0/3.0;
    double hx = cx*cx - cs2, hy = cy*cy - cs2, hz = cz*cz - cs2;
    switch (m) {
    case  0: return 1.0;
    case  1: return cx;
    case  2: return cy;
    case  3: return cz;
    case  4: return hx;
    case  5: return cx*cy;
    case  6: return cx*cz;
    case  7: return hy;
    case  8: return cy*cz;
    case  9: return hz;
    case 10: return 3.0*hx*cy;
    case 11: return 3.0*hx*cz;
    case 12: return 3.0*hy*cz;
    case 13: return 3.0*hy*cx;
    case 14: return 3.0*hz*cx;
    case 15: return 3.0*hz*cy;
    case 16: return cx*cy*cz;
    case 17: return 9.0*hx*hy;
    case 18: return 9.0*hy*hz;
    case 19: return 9.0*hz*hx;
    case 20: return 9.0*hx*cy*cz;
    case 21: return 9.0*hy*cz*cx;
    case 22: return 9.0*hz*cx*cy;
    case 23: return 9.0*hx*hy*cz;
    case 24: return 9.0*hy*hz*cx;
    case 25: return 9.0*hz*hx*cy;
    default: return 27.0*hx*hy*hz;
    }
  }
};

template <int NVEL> struct Model : VSet<NVEL> {
  using V = VSet<NVEL>;
  __host__ __device__ static constexpr double na(int m) {
    double wip = 0.0;
    for (int p = 0; p < NVEL; p++) wip += V::w(p)*V::ma(m,p)*V::ma(m,p);
    return 1.0/wip;
  }
  __host__ __device__ static constexpr double mi(int p, int m) {
    return V::w(p)*na(m)*V::ma(m,p);
  }
};

/* Compile-time loop: the body sees the index as a constant expression, so
 * every table look-up above folds and zero coefficients vanish. */

template <int N> using IC = std::integral_constant<int, N>;

template <int B, int E, class F>
__device__ __forceinline__ void static_for(F && f) {
  if constexpr (B < E) {
    f(IC<B>{});
    static_for<B + 1, E>(f);
  }
}

/* ---- per-site collision ---------------------------------------------------
 *
 * lb_collision_mrt1_site, collision.c:259-599, fluctuations off, no free-
 * energy stress, constant viscosity. The reference transforms f to all NVEL
 * modes, relaxes, and transforms back with dense matrices. Here the same
 * linear map is evaluated in a form that needs only what each scheme uses:
 *
 *   f' = keep f + MI_hydro (m'_hydro - keep m_hydro) + sum_k MI[:,k] e_k
 *
 * with keep = 1 - rtau_even. M10 (all ghost rates 1): keep = 0, e = 0, so
 * f' is rebuilt from the 10 hydrodynamic moments alone. BGK: the stress
 * enters only through its equilibrium. TRT (d3q19): e_k = (rtau_even -
 * rtau_odd) m_k for the six odd ghost modes.
 *
 * REFERENCE QUIRK reproduced for parity: for d3q19 the reference's unrolled
 * d3q19_f2mode_chunk forms mode 13 with coefficient 0 instead of
 * ma[13][4] = -1 for population 4 (collision.c:2300), i.e. m13_ref =
 * m13 + f4. With ghost rate r13 the surviving part (1 - r13) f4 is
 * projected back through MI[:,13]. (No effect for M10.)
 */

struct Relax {
  double rtau_s, rtau_b, rtau_e, rtau_o;
};

/* Isothermal fluctuations (collision.c:476-518): a random stress with the
 * variance of the fluctuation-dissipation theorem joins the post-collision
 * stress, and, with ghost modes on, a random part each ghost mode
 * (lb_fluctuations_var_eta/_bulk/_ghost, _stress, _ghosts, :1745-1920).
 * The numbers are the reference's: one draw of its per-site generator gives
 * up to ten three-bit indices into Ladd's eight-entry table (noise.c:72-79,
 * 397-424, noise_uniform :467-487). D3Q19 only, as in the reference
 * (NNOISE_MAX = 10 < the 17 ghost modes of D3Q27, noise.h:18). */

struct SiteNoise {
  double shat[6];          /* xx xy xz yy yz zz */
  double ghat[9];          /* modes 10 .. 18 */
};

__device__ __forceinline__ unsigned int noise_uniform(unsigned int (&st)[4]) {
  st[0] = 69069u*st[0] + 1234567u;
  unsigned int b = st[1] ^ (st[1] << 17);
  b ^= (b >> 13);
  st[1] = b ^ (b << 5);
  st[2] = 36969u*(st[2] & 0xffffu) + (st[2] >> 16);
  st[3] = 18000u*(st[3] & 0xffffu) + (st[3] >> 16);
  b = (st[2] << 16) + st[3];
  return st[1] + (st[0] ^ b);
}

__device__ __forceinline__ double noise_table(unsigned int k) {
  const double a = sqrt(2.0 + sqrt(2.0));
  const double b = sqrt(2.0 - sqrt(2.0));
  const double v = (k == 0u || k == 7u) ? a : ((k == 1u || k == 6u) ? b : 0.0);
  return (k < 2u) ? -v : v;
}

/* STH: add the thermodynamic stress sth (xx xy xz yy yz zz) to the
 * equilibrium stress, as the two-distribution collision does
 * (collision.c:838-850) */

template <int NVEL, int SCHEME, bool STH, bool NZ = false>
__device__ __forceinline__
void collide_site_impl(double (&f)[NVEL], const double (&frc)[3],
		       const Relax & rx, const double (&sth)[6],
		       double & rho, double (&u)[3],
		       const SiteNoise * nz = nullptr) {

  static_assert(!NZ || NVEL == 19, "fluctuations: D3Q19 only (noise.h:18)");

  using M = Model<NVEL>;
  constexpr bool keepf  = (SCHEME != LBMI_M10);
  constexpr bool need_s = (SCHEME != LBMI_BGK);
  constexpr bool odd    = (SCHEME == LBMI_TRT);
  constexpr double r3 = 1.0/3.0;

  /* rho and momentum: summed in p order with exact +-1 coefficients, as the
   * reference's rows 0-3 of ma (collision.c:338-349) */
  double g[3] = {0.0, 0.0, 0.0};
  rho = 0.0;
  static_for<0, NVEL>([&](auto P) {
    constexpr int p = P;
    rho += f[p];
    static_for<0, 3>([&](auto A) {
      constexpr int a = A;
      if constexpr (M::c(p,a) ==  1) g[a] += f[p];
      if constexpr (M::c(p,a) == -1) g[a] -= f[p];
    });
  });

  double rrho = 1.0/rho;
  u[0] = rrho*(g[0] + 0.5*frc[0]);            /* collision.c:376-382 */
  u[1] = rrho*(g[1] + 0.5*frc[1]);
  u[2] = rrho*(g[2] + 0.5*frc[2]);

  /* second moment Pi_ab = sum_p f_p c_a c_b; S_ab = Pi_ab - delta_ab rho/3 */
  double s[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};   /* xx xy xz yy yz zz */
  if constexpr (need_s) {
    static_for<0, NVEL>([&](auto P) {
      constexpr int p = P;
      constexpr int cx = M::c(p,0), cy = M::c(p,1), cz = M::c(p,2);
      if constexpr (cx*cx == 1) s[0] += f[p];
      if constexpr (cx*cy == 1) s[1] += f[p];
      if constexpr (cx*cy == -1) s[1] -= f[p];
      if constexpr (cx*cz == 1) s[2] += f[p];
      if constexpr (cx*cz == -1) s[2] -= f[p];
      if constexpr (cy*cy == 1) s[3] += f[p];
      if constexpr (cy*cz == 1) s[4] += f[p];
      if constexpr (cy*cz == -1) s[4] -= f[p];
      if constexpr (cz*cz == 1) s[5] += f[p];
    });
    s[0] -= r3*rho; s[3] -= r3*rho; s[5] -= r3*rho;
  }

  /* Stress relaxation, collision.c:408-474. ds = S' - keep S */
  double ds[6];
  {
    const double keep = keepf ? (1.0 - rx.rtau_e) : 0.0;
    double seq[6] = {rho*u[0]*u[0], rho*u[0]*u[1], rho*u[0]*u[2],
		     rho*u[1]*u[1], rho*u[1]*u[2], rho*u[2]*u[2]};
    if constexpr (STH) {
      static_for<0, 6>([&](auto K) { seq[K] += sth[K]; });
    }
    const double fc = 2.0 - rx.rtau_s;
    const double uf[6] = {2.0*u[0]*frc[0], u[0]*frc[1] + frc[0]*u[1],
			  u[0]*frc[2] + frc[0]*u[2], 2.0*u[1]*frc[1],
			  u[1]*frc[2] + frc[1]*u[2], 2.0*u[2]*frc[2]};
    if constexpr (need_s) {
      double tr_s = s[0] + s[3] + s[5];
      double tr_seq = seq[0] + seq[3] + seq[5];
      double trp = r3*(tr_s - rx.rtau_b*(tr_s - tr_seq));
      constexpr int diag[6] = {1, 0, 0, 1, 0, 1};
      static_for<0, 6>([&](auto K) {
	constexpr int k = K;
	double sd  = s[k]   - (diag[k] ? r3*tr_s   : 0.0);
	double sqd = seq[k] - (diag[k] ? r3*tr_seq : 0.0);
	double sp  = sd - rx.rtau_s*(sd - sqd) + (diag[k] ? trp : 0.0)
	  + fc*uf[k];
	ds[k] = sp - keep*s[k];
      });
    }
    else {
      /* BGK: bulk = shear = ghost rate, S drops out of S' - keep S */
      static_for<0, 6>([&](auto K) {
	constexpr int k = K;
	ds[k] = rx.rtau_s*seq[k] + fc*uf[k];
      });
    }
  }

  if constexpr (NZ) {
    /* mode[4 .. 9] = S' + shat (collision.c:529-536) */
    static_for<0, 6>([&](auto K) { ds[K] += nz->shat[K]; });
  }

  /* hydrodynamic part of the back projection:
   * w_p [ (drho - 1.5 tr dS) + 3 dg.c + 4.5 dS_aa c_a^2 + 9 dS_ab c_a c_b ] */
  const double keep = keepf ? (1.0 - rx.rtau_e) : 0.0;
  const double drho = rho - keep*rho;
  const double a0 = drho - 1.5*(ds[0] + ds[3] + ds[5]);
  const double ag[3] = {3.0*(g[0] + frc[0] - keep*g[0]),
			3.0*(g[1] + frc[1] - keep*g[1]),
			3.0*(g[2] + frc[2] - keep*g[2])};
  const double bd[3] = {4.5*ds[0], 4.5*ds[3], 4.5*ds[5]};
  const double bo[3] = {9.0*ds[1], 9.0*ds[2], 9.0*ds[4]};  /* xy xz yz */

  /* odd ghost modes (TRT, d3q19) and the mode-13 quirk */
  double e[NVEL];
  if constexpr (NVEL == 19 && NZ) {
    /* ... + ghat (collision.c:540-545) */
    static_for<10, 19>([&](auto K) { e[K] = nz->ghat[K - 10]; });
  }
  else if constexpr (NVEL == 19 && keepf) {
    static_for<10, 19>([&](auto K) { e[K] = 0.0; });
  }
  if constexpr (NVEL == 19 && keepf) {
    if constexpr (odd) {
      const double dr = rx.rtau_e - rx.rtau_o;
      static_for<11, 18>([&](auto K) {
	constexpr int k = K;
	if constexpr (k != 14) {
	  double mk = 0.0;
	  static_for<0, NVEL>([&](auto P) {
	    constexpr int p = P;
	    constexpr double c = M::ma(k, p);
	    if constexpr (c == 1.0) mk += f[p];
	    else if constexpr (c == -1.0) mk -= f[p];
	    else if constexpr (c != 0.0) mk += c*f[p];
	  });
	  e[k] += dr*mk;
	}
      });
    }
    e[13] += (1.0 - rx.rtau_o)*f[4];          /* collision.c:2300 */
  }

  static_for<0, NVEL>([&](auto P) {
    constexpr int p = P;
    constexpr int cx = M::c(p,0), cy = M::c(p,1), cz = M::c(p,2);
    double t = a0;
    if constexpr (cx ==  1) t += ag[0];
    if constexpr (cx == -1) t -= ag[0];
    if constexpr (cy ==  1) t += ag[1];
    if constexpr (cy == -1) t -= ag[1];
    if constexpr (cz ==  1) t += ag[2];
    if constexpr (cz == -1) t -= ag[2];
    if constexpr (cx != 0) t += bd[0];
    if constexpr (cy != 0) t += bd[1];
    if constexpr (cz != 0) t += bd[2];
    if constexpr (cx*cy ==  1) t += bo[0];
    if constexpr (cx*cy == -1) t -= bo[0];
    if constexpr (cx*cz ==  1) t += bo[1];
    if constexpr (cx*cz == -1) t -= bo[1];
    if constexpr (cy*cz ==  1) t += bo[2];
    if constexpr (cy*cz == -1) t -= bo[2];
    /* (the coefficients must be constant expressions at the point of use:
     * a plain call of the constexpr table function is evaluated at run time,
     * na(m) loop and all -- 2.2x (BGK) and 9x (TRT) on the whole kernel) */
    constexpr double wp = M::w(p);
    double fn = wp*t;
    if constexpr (keepf) fn += keep*f[p];
    if constexpr (NVEL == 19 && NZ) {
      static_for<10, 19>([&](auto K) {
	constexpr int k = K;
	constexpr double mipk = M::mi(p,k);
	if constexpr (mipk != 0.0) fn += mipk*e[k];
      });
    }
    else if constexpr (NVEL == 19 && keepf) {
      if constexpr (odd) {
	static_for<11, 18>([&](auto K) {
	  constexpr int k = K;
	  if constexpr (k != 14) {
	    constexpr double mipk = M::mi(p,k);
	    if constexpr (mipk != 0.0) fn += mipk*e[k];
	  }
	});
      }
      else {
	constexpr double mip13 = M::mi(p,13);
	if constexpr (mip13 != 0.0) fn += mip13*e[13];
      }
    }
    f[p] = fn;
  });
}

/* Component stride of the hydro arrays force and u (hydro->nsite): the
 * lattice's own nsite unless the caller says otherwise -- with Lees-Edwards
 * planes the reference allocates them with buffer planes (lees_edw_nsites,
 * hydro.c:75-88), the distributions without (model.c:295). */

__device__ __forceinline__
size_t hstride(const lbmi_kparam_t & kp, const lbmi_hydro_dev_t & h) {
  return (h.stride > 0) ? (size_t) h.stride : (size_t) kp.nsite;
}

/* Relaxation rates of a site: the constants of lbmi_set_relaxation, or,
 * with a viscosity model, those of the local shear viscosity hydro->eta
 * (collision.c:386-404 with lb_relaxation_time_shear_v, _bulk_v, _ghosts_v,
 * :1287-1538; the bulk viscosity keeps the Newtonian ratio) */

template <int SCHEME>
__device__ __forceinline__
Relax site_relax(const lbmi_kparam_t & kp, const lbmi_hydro_dev_t & h, int i) {
  Relax rx = {kp.rtau_shear, kp.rtau_bulk, kp.rtau_even, kp.rtau_odd};
  if (h.eta) {
    const double cs2 = (1.0/3.0);
    const double eta = h.eta[i];
    const double rtau = 1.0/(0.5 + eta/(kp.rho0*cs2));
    rx.rtau_s = rtau;
    if constexpr (SCHEME == LBMI_BGK) {
      rx.rtau_b = rtau;
      rx.rtau_e = rtau;
      rx.rtau_o = rtau;
    }
    else {
      rx.rtau_b = 1.0/(0.5 + (kp.bulk_ratio*eta)/(kp.rho0*cs2));
      if constexpr (SCHEME == LBMI_TRT) {
	const double tau = eta/(kp.rho0*cs2);
	double rodd = 0.5 + 2.0*tau/(tau + 3.0/8.0);
	if (rodd > 2.0) rodd = 2.0;
	rx.rtau_e = rtau;
	rx.rtau_o = rodd;
      }
    }
  }
  return rx;
}

template <int NVEL, int SCHEME>
__device__ __forceinline__
void collide_site(double (&f)[NVEL], const double (&frc)[3], const Relax & rx,
		  double & rho, double (&u)[3]) {
  const double none[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  collide_site_impl<NVEL, SCHEME, false>(f, frc, rx, none, rho, u);
}

/* The random stress and ghost parts of site i, and its generator advanced
 * (collision.c:491-516: two draws per fluid site and step, the second only
 * with ghost modes on; solid sites draw nothing). The state is the
 * reference's array noise->state, four unsigned ints per site, SoA. */

template <int SCHEME>
__device__ __forceinline__
void site_noise(const lbmi_hydro_dev_t & h, int i, const Relax & rx,
		SiteNoise & nz) {
  using M = Model<19>;
  const size_t hs = (size_t) h.noise_stride;
  unsigned int st[4] = {h.noise[i], h.noise[hs + i], h.noise[2*hs + i],
			h.noise[3*hs + i]};
  const double kt = h.noise_kt*3.0;              /* kt*rcs2 */

  {
    /* lb_fluctuations_var_eta, _var_bulk, _stress (collision.c:1753-1887) */
    const double tau = 1.0/rx.rtau_s;
    const double taub = 1.0/rx.rtau_b;
    const double var = sqrt(kt)*sqrt(1.0/9.0)*sqrt((tau + tau - 1.0)/(tau*tau));
    const double varb = sqrt(kt)*sqrt(2.0/9.0)*sqrt((taub + taub - 1.0)/(taub*taub));
    unsigned int iu = noise_uniform(st) >> 2;
    double r[6];
    static_for<0, 6>([&](auto K) { r[K] = noise_table(iu & 7u); iu >>= 3; });
    double tr = (1.0/3.0)*(r[0] + r[3] + 1.0*r[5]);
    r[0] -= tr; r[3] -= tr; r[5] -= tr;
    const double vd = var*sqrt(2.0);
    nz.shat[0] = r[0]*vd;  nz.shat[1] = r[1]*var; nz.shat[2] = r[2]*var;
    nz.shat[3] = r[3]*vd;  nz.shat[4] = r[4]*var; nz.shat[5] = r[5]*vd;
    tr *= varb;
    nz.shat[0] += tr; nz.shat[3] += tr; nz.shat[5] += tr;
  }

  static_for<0, 9>([&](auto K) { nz.ghat[K] = 0.0; });
  if (h.noise_ghosts) {
    /* lb_fluctuations_var_ghost, _ghosts (collision.c:1800-1918): the
     * numbers go to the ghost modes in the order of the basis */
    unsigned int ig = noise_uniform(st) >> 2;
    static_for<10, 19>([&](auto K) {
      constexpr int k = K;
      constexpr bool oddk = (k >= 11 && k <= 17 && k != 14);
      double rate = oddk ? rx.rtau_o : rx.rtau_e;
      if constexpr (SCHEME == LBMI_M10) rate = 1.0;
      if constexpr (SCHEME == LBMI_BGK) rate = rx.rtau_s;
      const double taug = 1.0/rate;
      constexpr double rna = 1.0/M::na(k);
      const double varg = sqrt(kt*rna)*sqrt((taug + taug - 1.0)/(taug*taug));
      nz.ghat[k - 10] = varg*noise_table(ig & 7u);
      ig >>= 3;
    });
  }

  h.noise[i] = st[0];
  h.noise[hs + i] = st[1];
  h.noise[2*hs + i] = st[2];
  h.noise[3*hs + i] = st[3];
}

template <int NVEL, int SCHEME, bool NZ>
__device__ __forceinline__
void collide_site_nz(const lbmi_hydro_dev_t & h, int i, double (&f)[NVEL],
		     const double (&frc)[3], const Relax & rx,
		     double & rho, double (&u)[3]) {
  if constexpr (NZ) {
    const double none[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    SiteNoise nz;
    site_noise<SCHEME>(h, i, rx, nz);
    collide_site_impl<NVEL, SCHEME, false, true>(f, frc, rx, none, rho, u, &nz);
  }
  else {
    collide_site<NVEL, SCHEME>(f, frc, rx, rho, u);
  }
}

/* ---- helpers -------------------------------------------------------------- */

/* XCD-aware mapping of blockIdx to a logical block: blocks b and b + 8 run
 * on the same XCD (round-robin dispatch), so give XCD (b & 7) the contiguous
 * chunk [xcd*per, (xcd+1)*per) of the 1-d site range; neighbouring blocks
 * then share their boundary cache lines in one L2. A different hardware
 * placement only changes speed. Returns false for padding blocks. */

__device__ __forceinline__ bool logical_block(unsigned nblk, unsigned & lb,
					      unsigned group = 0) {
#if !LBMI_XCD_REMAP
  lb = blockIdx.x;
  return lb < nblk;
#endif
  if (group > 0) {
    /* XCDs interleaved at a granularity of `group` blocks: all eight work
     * in the same neighbourhood of every population array, which the HBM
     * system serves better than eight far-apart windows per array */
    unsigned xcd = blockIdx.x & 7u, j = blockIdx.x >> 3;
    unsigned grp = j/group, within = j - grp*group;
    lb = (grp*8u + xcd)*group + within;
    return lb < nblk;
  }
  unsigned per = (nblk + 7u) >> 3;
  lb = (blockIdx.x & 7u)*per + (blockIdx.x >> 3);
  return (blockIdx.x >> 3) < per && lb < nblk;
}

struct Site {
  int i;         /* site index */
  int x, y, z;   /* 0-based coordinates in nall */
  bool interior; /* y and z inside the local domain (x is by construction) */
};

__device__ __forceinline__ Site decode(const lbmi_kparam_t & kp, int i) {
  Site s;
  s.i = i;
#if LBMI_FAST_DECODE
  /* exact for 0 <= i < 2^31: rstr = (1/str)(1 + 2^-40), see lbmi_host.c */
  s.x = (int) ((double) i*kp.rstrx);
  int r = i - s.x*kp.strx;
  s.y = (int) ((double) r*kp.rstry);
  s.z = r - s.y*kp.stry;
#else
  s.x = i / kp.strx;
  int r = i - s.x*kp.strx;
  s.y = r / kp.stry;
  s.z = r - s.y*kp.stry;
#endif
  s.interior = (s.y >= kp.nhalo) && (s.y < kp.nhalo + kp.nlocal[1]) &&
    (s.z >= kp.nhalo) && (s.z < kp.nhalo + kp.nlocal[2]);
  return s;
}

/* ---- k_collide: in-place collision of interior fluid sites ---------------- */

template <int NVEL, int SCHEME, bool NZ = false>
__global__ __launch_bounds__(BLOCK)
void k_collide(lbmi_kparam_t kp, double * __restrict__ f,
	       lbmi_hydro_dev_t h, int i0, int i1, unsigned nblk) {

  unsigned lb;
  if (!logical_block(nblk, lb, (unsigned) kp.xcd_group)) return;
  int i = (i0/LBMI_ALIGN)*LBMI_ALIGN + (int) (lb*BLOCK + threadIdx.x);
  if (i < i0 || i >= i1) return;

  Site s = decode(kp, i);
  if (!s.interior) return;
  if (h.status && h.status[i] != 0) return;      /* collision.c:299-304 */

  const size_t ns = (size_t) kp.nsite;
  double fl[NVEL];
  static_for<0, NVEL>([&](auto P) { fl[P] = f[ns*P + i]; });

  double frc[3] = {kp.fbody[0], kp.fbody[1], kp.fbody[2]};
  if (h.force) {
    frc[0] += h.force[i];
    frc[1] += h.force[hstride(kp, h) + i];
    frc[2] += h.force[2*hstride(kp, h) + i];
  }

  Relax rx = site_relax<SCHEME>(kp, h, i);
  double rho, u[3];
  collide_site_nz<NVEL, SCHEME, NZ>(h, i, fl, frc, rx, rho, u);

  static_for<0, NVEL>([&](auto P) { f[ns*P + i] = fl[P]; });
  if (h.rho) h.rho[i] = rho;
  if (h.u) {
    h.u[i] = u[0];
    h.u[hstride(kp, h) + i] = u[1];
    h.u[2*hstride(kp, h) + i] = u[2];
  }
}

/* k_collide_to: the same collision OUT of place, f -> fo over the whole array:
 * sites that do not collide (halo, non-fluid) are copied. Read-modify-write
 * of the same addresses is the slow way to move these bytes on this memory
 * system (1.22 ms at 256^3 against 1.0 ms out of place with the same
 * traffic, tools/eager_probe.sh): the EAGER lb_collide writes the other array
 * and the handle swaps the two, as lb_propagation does. */

template <int NVEL, int SCHEME>
__global__ __launch_bounds__(BLOCK)
void k_collide_to(lbmi_kparam_t kp, const double * __restrict__ f,
		  double * __restrict__ fo, lbmi_hydro_dev_t h, unsigned nblk) {

  unsigned lb;
  if (!logical_block(nblk, lb, (unsigned) kp.xcd_group)) return;
  const int i = (int) (lb*BLOCK + threadIdx.x);
  if (i >= kp.nsite) return;

  const Site s = decode(kp, i);
  const int nh = kp.nhalo;
  const size_t ns = (size_t) kp.nsite;
  double fl[NVEL];
  static_for<0, NVEL>([&](auto P) { fl[P] = f[ns*P + i]; });

  if (s.interior && s.x >= nh && s.x < nh + kp.nlocal[0] &&
      !(h.status && h.status[i] != 0)) {         /* collision.c:299-304 */
    double frc[3] = {kp.fbody[0], kp.fbody[1], kp.fbody[2]};
    if (h.force) {
      frc[0] += h.force[i];
      frc[1] += h.force[hstride(kp, h) + i];
      frc[2] += h.force[2*hstride(kp, h) + i];
    }
    Relax rx = site_relax<SCHEME>(kp, h, i);
    double rho, u[3];
    collide_site_nz<NVEL, SCHEME, false>(h, i, fl, frc, rx, rho, u);
    if (h.rho) h.rho[i] = rho;
    if (h.u) {
      h.u[i] = u[0];
      h.u[hstride(kp, h) + i] = u[1];
      h.u[2*hstride(kp, h) + i] = u[2];
    }
  }
  static_for<0, NVEL>([&](auto P) { fo[ns*P + i] = fl[P]; });
}

/* k_collide_fe: the same with fe->use_stress_relaxation (collision.c:413-
 * 429) for the symmetric free energy: its stress at the site, from phi and
 * the field_grad_compute arrays, joins the equilibrium stress. */

template <int NVEL, int SCHEME>
__global__ __launch_bounds__(BLOCK)
void k_collide_fe(lbmi_kparam_t kp, double * __restrict__ f,
		  lbmi_hydro_dev_t h, double qa, double qb, double qkappa,
		  const double * __restrict__ phi,
		  const double * __restrict__ grad,
		  const double * __restrict__ delsq, int i0, int i1,
		  unsigned nblk) {

  unsigned lb;
  if (!logical_block(nblk, lb, (unsigned) kp.xcd_group)) return;
  int i = (i0/LBMI_ALIGN)*LBMI_ALIGN + (int) (lb*BLOCK + threadIdx.x);
  if (i < i0 || i >= i1) return;

  Site s = decode(kp, i);
  if (!s.interior) return;
  if (h.status && h.status[i] != 0) return;      /* collision.c:299-304 */

  const size_t ns = (size_t) kp.nsite;
  double fl[NVEL];
  static_for<0, NVEL>([&](auto P) { fl[P] = f[ns*P + i]; });

  double frc[3] = {kp.fbody[0], kp.fbody[1], kp.fbody[2]};
  if (h.force) {
    frc[0] += h.force[i];
    frc[1] += h.force[hstride(kp, h) + i];
    frc[2] += h.force[2*hstride(kp, h) + i];
  }

  Relax rx = site_relax<SCHEME>(kp, h, i);
  double rho, u[3];
  double sth[6];
  {
    /* P_ab = p0 delta_ab + kappa d_a phi d_b phi (symmetric.c:371-420) */
    const double ph = phi[i], d2 = delsq[i];
    const size_t gs = (h.gstride > 0) ? (size_t) h.gstride : ns;
    const double g0 = grad[i], g1 = grad[gs + i], g2 = grad[2*gs + i];
    const double p0 = 0.5*qa*ph*ph + 0.75*qb*ph*ph*ph*ph - qkappa*ph*d2
      - 0.5*qkappa*(g0*g0 + g1*g1 + g2*g2);
    sth[0] = p0 + qkappa*g0*g0; sth[1] = qkappa*g0*g1; sth[2] = qkappa*g0*g2;
    sth[3] = p0 + qkappa*g1*g1; sth[4] = qkappa*g1*g2;
    sth[5] = p0 + qkappa*g2*g2;
  }
  const double (&sthr)[6] = sth;
  collide_site_impl<NVEL, SCHEME, true>(fl, frc, rx, sthr, rho, u);

  static_for<0, NVEL>([&](auto P) { f[ns*P + i] = fl[P]; });
  if (h.rho) h.rho[i] = rho;
  if (h.u) {
    h.u[i] = u[0];
    h.u[hstride(kp, h) + i] = u[1];
    h.u[2*hstride(kp, h) + i] = u[2];
  }
}

/* ---- k_propagate: pull streaming ------------------------------------------ */

template <int NVEL>
__global__ __launch_bounds__(BLOCK)
void k_propagate(lbmi_kparam_t kp, const double * __restrict__ f,
		 double * __restrict__ fp, int i0, int i1, unsigned nblk) {

  using M = Model<NVEL>;
  unsigned lb;
  if (!logical_block(nblk, lb, (unsigned) kp.xcd_group)) return;
  int i = (i0/LBMI_ALIGN)*LBMI_ALIGN + (int) (lb*BLOCK + threadIdx.x);
  if (i < i0 || i >= i1) return;

  Site s = decode(kp, i);
  const size_t ns = (size_t) kp.nsite;
  const int m = s.interior ? 1 : 0;         /* kernel_mask_v, kernel.c:374 */

  static_for<0, NVEL>([&](auto P) {
    constexpr int p = P;
    int off = m*(M::c(p,0)*kp.strx + M::c(p,1)*kp.stry + M::c(p,2));
    fp[ns*p + i] = f[ns*p + (i - off)];
  });
}

/* ---- k_propagate_collide: the fused kernel --------------------------------
 *
 * fprime[i] = collide( f[i - c_p] ), one pass: 2*NVEL*8 bytes per site.
 * WRAP: directions flagged in wrapmask are periodic and local to this
 * GPU; a pull that would land in the halo is redirected to the periodic
 * image inside the domain, so no halo swap is needed for them.
 * y/z halo sites inside the processed x-planes copy in place (as
 * lb_propagation_kernel does), which keeps every store stream contiguous. */

/* Orders of a distribution array. SoA is the reference's (-DADDR_SOA,
 * memory.h:250-258): population p of site i at nsite*p + i, NVEL streams
 * nsite*8 bytes apart. The blocked order keeps LBW consecutive sites of all
 * populations together, [i/LBW][p][i%LBW]: one thread block then writes ONE
 * contiguous NVEL*LBW*8-byte chunk instead of NVEL pieces 137 MB apart, and
 * its pulls come from a few neighbouring chunks. It exists only as the
 * internal order of a deferred (FUSED) state on a single GPU: whatever a
 * caller can observe is converted back (lbmi_lb_flush). The two orders
 * occupy the same nsite*NVEL doubles; sites beyond the last whole block
 * (they lie in the last x halo plane) are not representable and never used. */

constexpr int LBW = 256;

template <int NVEL, bool BLK>
__device__ __forceinline__ size_t faddr(size_t ns, int p, int i) {
  if constexpr (BLK) {
    return (size_t) (i >> 8)*(size_t) (NVEL*LBW) + (size_t) (p*LBW + (i & (LBW - 1)));
  }
  else {
    return ns*p + i;
  }
}

/* Slabs: the boundary launch of a FUSED step takes the populations that
 * cross the X faces straight from the receive buffers of the exchange
 * ([k][plane site], k-th population with c_x = +1 from the lower neighbour,
 * c_x = -1 from the upper one, in p order: the reduced selection of
 * model.c:1192-1219) instead of from halo planes somebody has unpacked them
 * into, and leaves the populations the NEXT exchange will send in the send
 * buffers instead of having a pack kernel collect them. */

template <int NVEL> __host__ __device__ constexpr int xrank(int p, int cx) {
  int k = 0;
  for (int q = 0; q < p; q++) {
    if (Model<NVEL>::c(q, 0) == cx) k += 1;
  }
  return k;
}

template <int NVEL>
struct PulledSite {
  double fl[NVEL];
  Site s;
};

/* phase 1: issue the NVEL pulls of site i */
template <int NVEL, bool WRAP, bool RB, bool XB>
__device__ __forceinline__
void pc_pull(const lbmi_kparam_t & kp, const double * __restrict__ f,
	     int wrapmask, int i, PulledSite<NVEL> & ps,
	     const lbmi_xbuf_t & xb) {

  using M = Model<NVEL>;
  ps.s = decode(kp, i);
  const Site & s = ps.s;
  const size_t ns = (size_t) kp.nsite;
  const int m = s.interior ? 1 : 0;

  /* wrap corrections (in sites) for pulls across the low / high face */
  int wlo[3] = {0, 0, 0};
  int whi[3] = {0, 0, 0};
  if constexpr (WRAP) {
    const int nh = kp.nhalo;
    if ((wrapmask & 1) && s.x == nh) wlo[0] = kp.nlocal[0]*kp.strx;
    if ((wrapmask & 1) && s.x == nh + kp.nlocal[0] - 1) whi[0] = -kp.nlocal[0]*kp.strx;
    if ((wrapmask & 2) && s.y == nh) wlo[1] = kp.nlocal[1]*kp.stry;
    if ((wrapmask & 2) && s.y == nh + kp.nlocal[1] - 1) whi[1] = -kp.nlocal[1]*kp.stry;
    if ((wrapmask & 4) && s.z == nh) wlo[2] = kp.nlocal[2];
    if ((wrapmask & 4) && s.z == nh + kp.nlocal[2] - 1) whi[2] = -kp.nlocal[2];
  }

#if LBMI_HALO_LANES_COPY
  /* (the ablation variant has no pull against the exchange buffers of a slab:
   * it would read halo planes nobody has filled) */
  static_assert(!XB, "LBMI_HALO_LANES_COPY has no x_direct (XB) variant: build "
		"the ablation with the boundary launch off, lbmi_tune x_direct 0");
  static_for<0, NVEL>([&](auto P) {
    constexpr int p = P;
    constexpr int cx = M::c(p,0), cy = M::c(p,1), cz = M::c(p,2);
    int off = cx*kp.strx + cy*kp.stry + cz;
    if constexpr (WRAP) {
      if constexpr (cx ==  1) off -= wlo[0];
      if constexpr (cx == -1) off -= whi[0];
      if constexpr (cy ==  1) off -= wlo[1];
      if constexpr (cy == -1) off -= whi[1];
      if constexpr (cz ==  1) off -= wlo[2];
      if constexpr (cz == -1) off -= whi[2];
    }
    ps.fl[p] = ldf(&f[faddr<NVEL, RB>(ns, p, i - m*off)]);
  });
#else
  /* y/z halo lanes (2 per row) load nothing and store zeros: the halo of
   * fprime is undefined until the next halo swap in any case, and writing
   * them keeps every store stream contiguous (whole 64-byte sectors) */
  static_for<0, NVEL>([&](auto P) { ps.fl[P] = 0.0; });
  if (s.interior) {
    static_for<0, NVEL>([&](auto P) {
      constexpr int p = P;
      constexpr int cx = M::c(p,0), cy = M::c(p,1), cz = M::c(p,2);
      int off = cx*kp.strx + cy*kp.stry + cz;
      if constexpr (WRAP) {
	if constexpr (cx ==  1) off -= wlo[0];
	if constexpr (cx == -1) off -= whi[0];
	if constexpr (cy ==  1) off -= wlo[1];
	if constexpr (cy == -1) off -= whi[1];
	if constexpr (cz ==  1) off -= wlo[2];
	if constexpr (cz == -1) off -= whi[2];
      }
      if constexpr (XB && cx == 1) {
	/* first interior plane: i - off lies in the low halo plane */
	if (s.x == kp.nhalo) {
	  constexpr int k = xrank<NVEL>(p, 1);
	  ps.fl[p] = xb.recvlo[(size_t) k*kp.strx + (i - off - (kp.nhalo - 1)*kp.strx)];
	  return;
	}
      }
      if constexpr (XB && cx == -1) {
	if (s.x == kp.nhalo + kp.nlocal[0] - 1) {
	  constexpr int k = xrank<NVEL>(p, -1);
	  ps.fl[p] = xb.recvhi[(size_t) k*kp.strx + (i - off - (kp.nhalo + kp.nlocal[0])*kp.strx)];
	  return;
	}
      }
      ps.fl[p] = ldf(&f[faddr<NVEL, RB>(ns, p, i - off)]);
    });
  }
#endif
}

/* phase 2: collide (interior fluid sites) and store site i. HIO = false: the
 * step has no hydro traffic (no force field to read, rho and u not wanted
 * now): a variant of its own, so that it is also a kernel of its own name in
 * a profile */
template <int NVEL, int SCHEME, bool WB, bool NTS, bool HIO, bool XB, bool NZ = false, bool FEXT = false>
__device__ __forceinline__
void pc_collide_store(const lbmi_kparam_t & kp, double * __restrict__ fp,
		      const lbmi_hydro_dev_t & h, int i,
		      PulledSite<NVEL> & ps, const lbmi_xbuf_t & xb,
		      double fx = 0.0, double fy = 0.0, double fz = 0.0,
		      int istore = -1, bool image = false) {

  /* i: the site whose collision this is (its force, status, viscosity);
   * istore >= 0: where the populations go instead -- a halo site whose
   * periodic image i is (k_propagate_collide_halo); rho and u of an image are
   * not stored again */
  const size_t ns = (size_t) kp.nsite;
  const int is = (istore >= 0) ? istore : i;
  bool active = ps.s.interior;
  if (h.status) active = active && (h.status[i] == 0);
#if LBMI_ABL_NOCOLLIDE
  active = false;
#endif

  if (active) {
    double frc[3] = {kp.fbody[0], kp.fbody[1], kp.fbody[2]};
    if constexpr (FEXT) {
      /* the force of the site handed over in registers (k_symm_lb_step: the
       * thermodynamic force, which the reference adds into hydro->force) */
      frc[0] += fx;
      frc[1] += fy;
      frc[2] += fz;
    }
    else if constexpr (HIO) {
      if (h.force) {
	frc[0] += h.force[i];
	frc[1] += h.force[hstride(kp, h) + i];
	frc[2] += h.force[2*hstride(kp, h) + i];
      }
    }
    Relax rx = site_relax<SCHEME>(kp, h, i);
    double rho, u[3];
    collide_site_nz<NVEL, SCHEME, NZ>(h, i, ps.fl, frc, rx, rho, u);
    if constexpr (!HIO) {
      /* nothing to store */
    }
    else if (image) {
      /* the collision of the site itself stores them */
    }
    else if (kp.nt_store & 2) {
      /* rho and u are written once and not read again by this kernel */
      if (h.rho) __builtin_nontemporal_store(rho, &h.rho[i]);
      if (h.u) {
	__builtin_nontemporal_store(u[0], &h.u[i]);
	__builtin_nontemporal_store(u[1], &h.u[hstride(kp, h) + i]);
	__builtin_nontemporal_store(u[2], &h.u[2*hstride(kp, h) + i]);
      }
    }
    else {
      if (h.rho) h.rho[i] = rho;
      if (h.u) {
	h.u[i] = u[0];
	h.u[hstride(kp, h) + i] = u[1];
	h.u[2*hstride(kp, h) + i] = u[2];
      }
    }
  }

  static_for<0, NVEL>([&](auto P) {
    if constexpr (NTS) {
      __builtin_nontemporal_store(ps.fl[P], &fp[faddr<NVEL, WB>(ns, P, is)]);
    }
    else {
      stf(&fp[faddr<NVEL, WB>(ns, P, is)], ps.fl[P]);
    }
  });

  if constexpr (XB) {
    /* what the next exchange sends: of the first interior plane the
     * populations that leave downwards (-> the lower neighbour's recvhi), of
     * the last one those that leave upwards; whole planes, y/z halo lanes
     * included (nobody reads those entries: y and z wrap by index) */
    using M = Model<NVEL>;
    if (ps.s.x == kp.nhalo) {
      const int j = i - kp.nhalo*kp.strx;
      static_for<0, NVEL>([&](auto P) {
	constexpr int p = P;
	if constexpr (M::c(p,0) == -1) {
	  constexpr int k = xrank<NVEL>(p, -1);
	  xb.sendlo[(size_t) k*kp.strx + j] = ps.fl[p];
	}
      });
    }
    if (ps.s.x == kp.nhalo + kp.nlocal[0] - 1) {
      const int j = i - (kp.nhalo + kp.nlocal[0] - 1)*kp.strx;
      static_for<0, NVEL>([&](auto P) {
	constexpr int p = P;
	if constexpr (M::c(p,0) == 1) {
	  constexpr int k = xrank<NVEL>(p, 1);
	  xb.sendhi[(size_t) k*kp.strx + j] = ps.fl[p];
	}
      });
    }
  }
}

/* LAY & 3: 0 SoA -> SoA, 1 SoA -> blocked, 2 blocked -> blocked;
 * LAY & 4: nontemporal stores of fprime */

template <int NVEL, int SCHEME, bool WRAP, int LAY, bool HIO, bool XB, bool NZ = false>
__global__ __launch_bounds__(BLOCK, LBMI_WAVES)
void k_propagate_collide(lbmi_kparam_t kp, const double * __restrict__ f,
			 double * __restrict__ fp, lbmi_hydro_dev_t h,
			 int wrapmask, int i0, int i1, unsigned nblk,
			 int j0, int j1, unsigned nblk_first, lbmi_xbuf_t xb) {

  unsigned lb;
  if (!logical_block(nblk, lb, (unsigned) kp.xcd_group)) return;

  /* two site ranges in one launch (the two boundary x-planes of a slab):
   * logical blocks [0, nblk_first) cover [i0, i1), the rest [j0, j1) */
  if (lb >= nblk_first) {
    lb -= nblk_first;
    i0 = j0;
    i1 = j1;
  }

  /* SPT sites per thread, BLOCK apart: the pulls of all of them are in
   * flight before the first collision starts */
  int i[SPT];
  PulledSite<NVEL> ps[SPT];
  /* blocks start at multiples of LBMI_ALIGN sites so that every store of a
   * wave covers whole 64-byte sectors of every population array (nsite*8
   * is a multiple of 64 for the sizes of interest); lanes before i0 idle */
  constexpr int ORD = LAY & 3;
  constexpr int ALIGNV = (ORD == 0) ? LBMI_ALIGN : LBW;
  const int i0a = (i0/ALIGNV)*ALIGNV;
  static_for<0, SPT>([&](auto K) {
    constexpr int k = K;
    i[k] = i0a + (int) (lb*(BLOCK*SPT) + k*BLOCK + threadIdx.x);
    if (i[k] >= i0 && i[k] < i1) {
      pc_pull<NVEL, WRAP, ORD == 2, XB>(kp, f, wrapmask, i[k], ps[k], xb);
    }
  });
  static_for<0, SPT>([&](auto K) {
    constexpr int k = K;
    if (i[k] >= i0 && i[k] < i1) {
      pc_collide_store<NVEL, SCHEME, ORD != 0, (LAY & 4) != 0, HIO, XB, NZ>(kp, fp, h, i[k], ps[k], xb);
    }
  });
}

/* k_propagate_collide_fe: propagation(t) fused with the collision(t+1) of
 * fe->use_stress_relaxation (collision.c:413-429: the symmetric stress of the
 * site, from phi and the field_grad_compute arrays, joins the equilibrium
 * stress) -- k_collide_fe's arithmetic behind the pull of k_propagate_collide,
 * SoA -> SoA; WRAP: the periodic directions by index (FUSED on one rank),
 * else from the halo as it is (FUSED_HALO). Before it lbmi_lb_collide_fe
 * flushed every step: three passes over f where this is one. */

template <int NVEL, int SCHEME, bool WRAP>
__global__ __launch_bounds__(BLOCK, LBMI_WAVES)
void k_propagate_collide_fe(lbmi_kparam_t kp, const double * __restrict__ f,
			    double * __restrict__ fp, lbmi_hydro_dev_t h,
			    double qa, double qb, double qkappa,
			    const double * __restrict__ phi,
			    const double * __restrict__ grad,
			    const double * __restrict__ delsq,
			    int wrapmask, int i0, int i1, unsigned nblk) {

  unsigned lb;
  if (!logical_block(nblk, lb, (unsigned) kp.xcd_group)) return;
  const int i = (i0/LBMI_ALIGN)*LBMI_ALIGN + (int) (lb*BLOCK + threadIdx.x);
  if (i < i0 || i >= i1) return;

  const lbmi_xbuf_t none = {nullptr, nullptr, nullptr, nullptr};
  const size_t ns = (size_t) kp.nsite;
  PulledSite<NVEL> ps;
  pc_pull<NVEL, WRAP, false, false>(kp, f, wrapmask, i, ps, none);

  bool active = ps.s.interior;
  if (h.status) active = active && (h.status[i] == 0);      /* collision.c:299-304 */
  if (active) {
    double frc[3] = {kp.fbody[0], kp.fbody[1], kp.fbody[2]};
    if (h.force) {
      frc[0] += h.force[i];
      frc[1] += h.force[hstride(kp, h) + i];
      frc[2] += h.force[2*hstride(kp, h) + i];
    }
    Relax rx = site_relax<SCHEME>(kp, h, i);
    double rho, u[3];
    double sth[6];
    {
      /* P_ab = p0 delta_ab + kappa d_a phi d_b phi (symmetric.c:371-420) */
      const double ph = phi[i], d2 = delsq[i];
      const size_t gs = (h.gstride > 0) ? (size_t) h.gstride : ns;
      const double g0 = grad[i], g1 = grad[gs + i], g2 = grad[2*gs + i];
      const double p0 = 0.5*qa*ph*ph + 0.75*qb*ph*ph*ph*ph - qkappa*ph*d2
	- 0.5*qkappa*(g0*g0 + g1*g1 + g2*g2);
      sth[0] = p0 + qkappa*g0*g0; sth[1] = qkappa*g0*g1; sth[2] = qkappa*g0*g2;
      sth[3] = p0 + qkappa*g1*g1; sth[4] = qkappa*g1*g2;
      sth[5] = p0 + qkappa*g2*g2;
    }
    const double (&sthr)[6] = sth;
    collide_site_impl<NVEL, SCHEME, true>(ps.fl, frc, rx, sthr, rho, u);
    if (h.rho) h.rho[i] = rho;
    if (h.u) {
      h.u[i] = u[0];
      h.u[hstride(kp, h) + i] = u[1];
      h.u[2*hstride(kp, h) + i] = u[2];
    }
  }
  static_for<0, NVEL>([&](auto P) { stf(&fp[ns*P + i], ps.fl[P]); });
}

/* k_propagate_collide_halo: the step of LBMI_MODE_FUSED_HALO on one rank with
 * the halo swap of its OWN result folded in. f is the reference's
 * post-collision state with its halo (whatever bounced back into it included);
 * the kernel pulls from it as it is, SoA -> SoA. The width-1 shell around the
 * interior of fprime -- which k_propagate_collide fills with zeros and three
 * k_halo_copy launches then overwrite with the periodic images (19 or 27
 * planes read and written again, per direction, in sequence) -- is computed
 * here: a shell lane runs the collision of its periodic image (same inputs,
 * same instructions, so the same bits as the copy would have delivered) and
 * stores it at its own place. 2.3 % more collisions on lanes that were
 * storing zeros, two x planes more, no halo launch, no second pass over the
 * boundary planes. What lb_halo means afterwards: nothing left to do, as long
 * as nobody has written to f in between (lbmi_host.c: halo_fresh). */

template <int NVEL, int SCHEME, bool HIO>
__global__ __launch_bounds__(BLOCK, LBMI_WAVES)
void k_propagate_collide_halo(lbmi_kparam_t kp, const double * __restrict__ f,
			      double * __restrict__ fp, lbmi_hydro_dev_t h,
			      int i0, int i1, unsigned nblk) {

  static_assert(SPT == 1, "one site per thread");
  unsigned lb;
  if (!logical_block(nblk, lb, (unsigned) kp.xcd_group)) return;
  const int i = (i0/LBMI_ALIGN)*LBMI_ALIGN + (int) (lb*BLOCK + threadIdx.x);
  if (i < i0 || i >= i1) return;

  const lbmi_xbuf_t none = {nullptr, nullptr, nullptr, nullptr};
  const Site s = decode(kp, i);
  const int nh = kp.nhalo;
  /* inside the interior or the width-1 shell around it? (an outer halo layer
   * of a wider allocation keeps the zero fill) */
  const bool near = (s.x >= nh - 1 && s.x <= nh + kp.nlocal[0] &&
		     s.y >= nh - 1 && s.y <= nh + kp.nlocal[1] &&
		     s.z >= nh - 1 && s.z <= nh + kp.nlocal[2]);
  const bool inside = s.interior && s.x >= nh && s.x < nh + kp.nlocal[0];
  int isrc = i;
  if (near && !inside) {
    /* the periodic image: every coordinate back into the local domain */
    int dx = 0, dy = 0, dz = 0;
    if (s.x < nh) dx = kp.nlocal[0]; else if (s.x >= nh + kp.nlocal[0]) dx = -kp.nlocal[0];
    if (s.y < nh) dy = kp.nlocal[1]; else if (s.y >= nh + kp.nlocal[1]) dy = -kp.nlocal[1];
    if (s.z < nh) dz = kp.nlocal[2]; else if (s.z >= nh + kp.nlocal[2]) dz = -kp.nlocal[2];
    isrc = i + dx*kp.strx + dy*kp.stry + dz;
  }
  PulledSite<NVEL> ps;
  if (near) {
    pc_pull<NVEL, false, false, false>(kp, f, 0, isrc, ps, none);
  }
  else {
    ps.s = s;
    ps.s.interior = false;
    static_for<0, NVEL>([&](auto P) { ps.fl[P] = 0.0; });
  }
  pc_collide_store<NVEL, SCHEME, false, false, HIO, false>(kp, fp, h, isrc, ps, none,
							    0.0, 0.0, 0.0, i, isrc != i);
}

/* k_relayout: the whole array from one order to the other (every whole
 * block of LBW sites, halo included), src != dst. Used when a deferred
 * blocked state has to become observable again (lbmi_lb_flush). */

/* (NVEL here: the number of components per site: nvel, or 2 nvel for the
 * two distributions of a symmetric_lb state) */
template <int NVEL, bool TO_BLK>
__global__ __launch_bounds__(BLOCK)
void k_relayout(lbmi_kparam_t kp, const double * __restrict__ src,
		double * __restrict__ dst, int nfull, unsigned nblk) {
  unsigned lb;
  if (!logical_block(nblk, lb, (unsigned) kp.xcd_group)) return;
  int i = (int) (lb*BLOCK + threadIdx.x);
  if (i >= nfull) return;
  const size_t ns = (size_t) kp.nsite;
  double v[NVEL];
  static_for<0, NVEL>([&](auto P) { v[P] = src[faddr<NVEL, !TO_BLK>(ns, P, i)]; });
  static_for<0, NVEL>([&](auto P) { dst[faddr<NVEL, TO_BLK>(ns, P, i)] = v[P]; });
}

/* k_hydro_from_f: hydro->rho and hydro->u of the LAST collision from the
 * post-collision distributions it left (either order), for a caller that
 * does not want them written every step (lbmi_tune "hydro_lazy"). The
 * collision conserves rho and adds F to the momentum, so with g' = sum f'_p c_p
 * the velocity it used, u = (g + F/2)/rho (collision.c:376-382), is
 * (g' - F/2)/rho. Fluid interior sites only, as lb_collision_mrt1_site
 * (collision.c:299-304, 571-579). The sums run in p order like the
 * collision's; the values agree with the ones it would have stored to a few
 * ulp (different operands: f' instead of f), well inside the 1e-12 of the
 * parity criterion. */

template <int NVEL, bool BLK>
__global__ __launch_bounds__(BLOCK)
void k_hydro_from_f(lbmi_kparam_t kp, const double * __restrict__ f,
		    lbmi_hydro_dev_t h, int i0, int i1, unsigned nblk) {

  using M = Model<NVEL>;
  unsigned lb;
  if (!logical_block(nblk, lb, (unsigned) kp.xcd_group)) return;
  int i = (i0/LBMI_ALIGN)*LBMI_ALIGN + (int) (lb*BLOCK + threadIdx.x);
  if (i < i0 || i >= i1) return;

  Site s = decode(kp, i);
  if (!s.interior) return;
  if (h.status && h.status[i] != 0) return;

  const size_t ns = (size_t) kp.nsite;
  double fl[NVEL];
  static_for<0, NVEL>([&](auto P) { fl[P] = f[faddr<NVEL, BLK>(ns, P, i)]; });

  double rho = 0.0;
  double g[3] = {0.0, 0.0, 0.0};
  static_for<0, NVEL>([&](auto P) {
    constexpr int p = P;
    rho += fl[p];
    static_for<0, 3>([&](auto A) {
      constexpr int a = A;
      if constexpr (M::c(p,a) ==  1) g[a] += fl[p];
      if constexpr (M::c(p,a) == -1) g[a] -= fl[p];
    });
  });

  double frc[3] = {kp.fbody[0], kp.fbody[1], kp.fbody[2]};
  if (h.force) {
    frc[0] += h.force[i];
    frc[1] += h.force[hstride(kp, h) + i];
    frc[2] += h.force[2*hstride(kp, h) + i];
  }
  const double rrho = 1.0/rho;
  if (h.rho) h.rho[i] = rho;
  if (h.u) {
    h.u[i] = rrho*(g[0] - 0.5*frc[0]);
    h.u[hstride(kp, h) + i] = rrho*(g[1] - 0.5*frc[1]);
    h.u[2*hstride(kp, h) + i] = rrho*(g[2] - 0.5*frc[2]);
  }
}

/* ---- in-place streaming: the AA pattern --------------------------------------
 *
 * One array, two alternating kernels (Bailey et al. 2009). Every thread reads
 * and writes the SAME set of addresses, so f is updated in place: HBM sees
 * nvel read-modify-write streams instead of nvel read + nvel write streams,
 * which the memory system serves ~10 % faster (profiles/r01_probes.txt).
 *
 *   k_aa_even:  C(t)                 read (x, slot p), collide, write the
 *                                    post-collision p to (x, slot opp(p)).
 *   k_aa_odd :  P(t) C(t+1) P(t+1)   read p from (x - c_p, slot opp(p))
 *                                    [= pull-propagation of the swapped
 *                                    layout], collide, write p to
 *                                    (x + c_p, slot p) [= push-propagation
 *                                    into the normal layout].
 *
 * With R(x) = {(x - c_p, opp(p))} and W(x) = {(x + c_p, p)} the substitution
 * q = opp(p) shows W(x) = R(x): no location is shared between threads.
 * opp(p) = nvel - p for p >= 1 (cv[p] = -cv[nvel - p] in both velocity sets,
 * tests/unit/test_lb_model.c:103-140). Periodic directions are wrapped by
 * index arithmetic on both the pull and the push; y/z halo lanes idle (their
 * lines are resident from the neighbouring lanes' loads, so the gaps in the
 * store streams cost nothing here). Pulling from the NORMAL order instead
 * (template argument SWAPPED_IN = false) is not race-free in place -- a thread
 * would push into locations its neighbours still have to pull from -- and is
 * never launched: after a flush the launcher swaps the slots first
 * (k_aa_unswap, a per-site involution).
 */

template <int NVEL> __host__ __device__ constexpr int opp(int p) {
  return (p == 0) ? 0 : NVEL - p;
}

struct WrapAdj { int wlo[3]; int whi[3]; };

__device__ __forceinline__
WrapAdj wrap_adjust(const lbmi_kparam_t & kp, const Site & s, int wrapmask) {
  WrapAdj w = {{0, 0, 0}, {0, 0, 0}};
  const int nh = kp.nhalo;
  if ((wrapmask & 1) && s.x == nh) w.wlo[0] = kp.nlocal[0]*kp.strx;
  if ((wrapmask & 1) && s.x == nh + kp.nlocal[0] - 1) w.whi[0] = -kp.nlocal[0]*kp.strx;
  if ((wrapmask & 2) && s.y == nh) w.wlo[1] = kp.nlocal[1]*kp.stry;
  if ((wrapmask & 2) && s.y == nh + kp.nlocal[1] - 1) w.whi[1] = -kp.nlocal[1]*kp.stry;
  if ((wrapmask & 4) && s.z == nh) w.wlo[2] = kp.nlocal[2];
  if ((wrapmask & 4) && s.z == nh + kp.nlocal[2] - 1) w.whi[2] = -kp.nlocal[2];
  return w;
}

template <int NVEL, int SCHEME>
__device__ __forceinline__
void site_collide_hydro(const lbmi_kparam_t & kp, const lbmi_hydro_dev_t & h,
			int i, double (&fl)[NVEL]) {
  const size_t ns = (size_t) kp.nsite;
  double frc[3] = {kp.fbody[0], kp.fbody[1], kp.fbody[2]};
  if (h.force) {
    frc[0] += h.force[i];
    frc[1] += h.force[hstride(kp, h) + i];
    frc[2] += h.force[2*hstride(kp, h) + i];
  }
  Relax rx = site_relax<SCHEME>(kp, h, i);
  double rho, u[3];
  collide_site<NVEL, SCHEME>(fl, frc, rx, rho, u);
  if (h.rho) h.rho[i] = rho;
  if (h.u) {
    h.u[i] = u[0];
    h.u[hstride(kp, h) + i] = u[1];
    h.u[2*hstride(kp, h) + i] = u[2];
  }
}

template <int NVEL, int SCHEME>
__global__ __launch_bounds__(BLOCK, LBMI_WAVES)
void k_aa_even(lbmi_kparam_t kp, double * __restrict__ f, lbmi_hydro_dev_t h,
	       int i0, int i1, unsigned nblk) {

  unsigned lb;
  if (!logical_block(nblk, lb, (unsigned) kp.xcd_group)) return;
  int i = (i0/LBMI_ALIGN)*LBMI_ALIGN + (int) (lb*BLOCK + threadIdx.x);
  if (i < i0 || i >= i1) return;

  Site s = decode(kp, i);
  if (!s.interior) return;

  const size_t ns = (size_t) kp.nsite;
  double fl[NVEL];
  static_for<0, NVEL>([&](auto P) { fl[P] = f[ns*P + i]; });

  bool fluid = true;
  if (h.status) fluid = (h.status[i] == 0);
#if LBMI_ABL_NOCOLLIDE
  fluid = false;
#endif
  if (fluid) site_collide_hydro<NVEL, SCHEME>(kp, h, i, fl);

  static_for<0, NVEL>([&](auto P) {
    constexpr int p = P;
    f[ns*opp<NVEL>(p) + i] = fl[p];
  });
}

template <int NVEL, int SCHEME, bool SWAPPED_IN>
__global__ __launch_bounds__(BLOCK, LBMI_WAVES)
void k_aa_odd(lbmi_kparam_t kp, double * __restrict__ f, lbmi_hydro_dev_t h,
	      int wrapmask, int i0, int i1, unsigned nblk) {

  using M = Model<NVEL>;
  unsigned lb;
  if (!logical_block(nblk, lb, (unsigned) kp.xcd_group)) return;
  int i = (i0/LBMI_ALIGN)*LBMI_ALIGN + (int) (lb*BLOCK + threadIdx.x);
  if (i < i0 || i >= i1) return;

  Site s = decode(kp, i);
  if (!s.interior) return;

  const size_t ns = (size_t) kp.nsite;
  const WrapAdj w = wrap_adjust(kp, s, wrapmask);

  /* pull: population p of this site sits at x - c_p (periodic image if that
   * is across a wrapped face), in slot opp(p) if the layout is swapped */
  double fl[NVEL];
  static_for<0, NVEL>([&](auto P) {
    constexpr int p = P;
    constexpr int cx = M::c(p,0), cy = M::c(p,1), cz = M::c(p,2);
    constexpr int slot = SWAPPED_IN ? opp<NVEL>(p) : p;
    int off = cx*kp.strx + cy*kp.stry + cz;
    if constexpr (cx ==  1) off -= w.wlo[0];
    if constexpr (cx == -1) off -= w.whi[0];
    if constexpr (cy ==  1) off -= w.wlo[1];
    if constexpr (cy == -1) off -= w.whi[1];
    if constexpr (cz ==  1) off -= w.wlo[2];
    if constexpr (cz == -1) off -= w.whi[2];
    fl[p] = f[ns*slot + (i - off)];
  });

  bool fluid = true;
  if (h.status) fluid = (h.status[i] == 0);
#if LBMI_ABL_NOCOLLIDE
  fluid = false;
#endif
  if (fluid) site_collide_hydro<NVEL, SCHEME>(kp, h, i, fl);

  /* push: post-collision p goes to x + c_p, slot p. Across the high face
   * (c = +1 at the last site) subtract the extent, across the low face
   * (c = -1 at the first site) add it: whi = -extent, wlo = +extent. */
  static_for<0, NVEL>([&](auto P) {
    constexpr int p = P;
    constexpr int cx = M::c(p,0), cy = M::c(p,1), cz = M::c(p,2);
    int off = cx*kp.strx + cy*kp.stry + cz;
    if constexpr (cx ==  1) off += w.whi[0];
    if constexpr (cx == -1) off += w.wlo[0];
    if constexpr (cy ==  1) off += w.whi[1];
    if constexpr (cy == -1) off += w.wlo[1];
    if constexpr (cz ==  1) off += w.whi[2];
    if constexpr (cz == -1) off += w.wlo[2];
    f[ns*p + (i + off)] = fl[p];
  });
}

/* swapped <-> normal slot layout, in place (its own inverse) */

template <int NVEL>
__global__ __launch_bounds__(BLOCK)
void k_aa_unswap(lbmi_kparam_t kp, double * __restrict__ f, int i0, int i1,
		 unsigned nblk) {
  unsigned lb;
  if (!logical_block(nblk, lb, (unsigned) kp.xcd_group)) return;
  int i = (i0/LBMI_ALIGN)*LBMI_ALIGN + (int) (lb*BLOCK + threadIdx.x);
  if (i < i0 || i >= i1) return;
  Site s = decode(kp, i);
  if (!s.interior) return;
  const size_t ns = (size_t) kp.nsite;
  static_for<1, (NVEL + 1)/2>([&](auto P) {
    constexpr int p = P;
    constexpr int q = opp<NVEL>(p);
    double a = f[ns*p + i];
    double b = f[ns*q + i];
    f[ns*p + i] = b;
    f[ns*q + i] = a;
  });
}

/* fprime[p][x] = f[p][x + c_p] with periodic wrap by index: the inverse of
 * the propagation (used to hand back a post-collision state when a flush
 * arrives after k_aa_odd has already propagated) */

template <int NVEL>
__global__ __launch_bounds__(BLOCK)
void k_unpropagate_wrap(lbmi_kparam_t kp, const double * __restrict__ f,
			double * __restrict__ fp, int wrapmask, int i0,
			int i1, unsigned nblk) {
  using M = Model<NVEL>;
  unsigned lb;
  if (!logical_block(nblk, lb, (unsigned) kp.xcd_group)) return;
  int i = (i0/LBMI_ALIGN)*LBMI_ALIGN + (int) (lb*BLOCK + threadIdx.x);
  if (i < i0 || i >= i1) return;
  Site s = decode(kp, i);
  const size_t ns = (size_t) kp.nsite;
  if (!s.interior) {
    static_for<0, NVEL>([&](auto P) { fp[ns*P + i] = f[ns*P + i]; });
    return;
  }
  const WrapAdj w = wrap_adjust(kp, s, wrapmask);
  static_for<0, NVEL>([&](auto P) {
    constexpr int p = P;
    constexpr int cx = M::c(p,0), cy = M::c(p,1), cz = M::c(p,2);
    int off = cx*kp.strx + cy*kp.stry + cz;
    if constexpr (cx ==  1) off += w.whi[0];
    if constexpr (cx == -1) off += w.wlo[0];
    if constexpr (cy ==  1) off += w.whi[1];
    if constexpr (cy == -1) off += w.wlo[1];
    if constexpr (cz ==  1) off += w.whi[2];
    if constexpr (cz == -1) off += w.wlo[2];
    fp[ns*p + i] = f[ns*p + (i + off)];
  });
}

/* ---- halo kernels ----------------------------------------------------------
 *
 * One pass of halo_swap_packed (halo_swap.c:709-1063) for direction dir on
 * one rank with periodic wrap: the first/last interior plane is copied to
 * the opposite width-1 halo plane next to the interior, over the FULL
 * extent (halo included) of the two other directions, so that running the
 * passes in the order X, Y, Z completes edges and corners.
 * grid.y enumerates (side, component); threads enumerate plane sites. */

__device__ __forceinline__
size_t plane_site(const lbmi_kparam_t & kp, int dir, int j, int coord) {
  /* j-th site of the plane dir = coord, fastest index = fastest memory */
  if (dir == 0) return (size_t) coord*kp.strx + j;
  if (dir == 1) {
    int x = j / kp.nall[2];
    int z = j - x*kp.nall[2];
    return (size_t) x*kp.strx + (size_t) coord*kp.stry + z;
  }
  /* dir == 2: j runs over (x, y), y fastest */
  return (size_t) j*kp.stry + coord;
}

__device__ __forceinline__ int plane_size(const lbmi_kparam_t & kp, int dir) {
  if (dir == 0) return kp.nall[1]*kp.nall[2];
  if (dir == 1) return kp.nall[0]*kp.nall[2];
  return kp.nall[0]*kp.nall[1];
}

__global__ __launch_bounds__(BLOCK)
void k_halo_copy(lbmi_kparam_t kp, int dir, lbmi_halo_sel_t sel,
		 double * __restrict__ data) {

  int j = blockIdx.x*BLOCK + threadIdx.x;
  if (j >= plane_size(kp, dir)) return;

  int k = blockIdx.y;
  const int w = blockIdx.z;                /* halo layer 0 .. nswap-1 */
  const int nh = kp.nhalo;
  int comp, src, dst;
  if (k < sel.nlo) {
    comp = sel.lo[k];                      /* low halo <- last interior */
    src = nh + kp.nlocal[dir] - 1 - w;
    dst = nh - 1 - w;
  }
  else {
    comp = sel.hi[k - sel.nlo];            /* high halo <- first interior */
    src = nh + w;
    dst = nh + kp.nlocal[dir] + w;
  }
  double * d = data + (size_t) kp.nsite*comp;
  d[plane_site(kp, dir, j, dst)] = d[plane_site(kp, dir, j, src)];
}

/* X planes to/from contiguous buffers [k][plane site] for the RCCL ring:
 * buf_lo carries the FIRST interior plane (wanted by the lower neighbour's
 * high halo: components sel.hi), buf_hi the LAST interior plane
 * (components sel.lo). */

/* address of component p of site i in an nvel-component array: SoA, or the
 * blocked order of a deferred distribution state (faddr above) */
__device__ __forceinline__
size_t xaddr(const lbmi_kparam_t & kp, int blocked, int p, size_t i) {
  if (blocked) {
    return (i >> 8)*(size_t) (kp.nvel*LBW) + (size_t) (p*LBW) + (i & (size_t) (LBW - 1));
  }
  return (size_t) kp.nsite*p + i;
}

/* dir: the decomposed direction (0: X, the planes are contiguous runs; 1, 2:
 * slabs along Y or Z, plane_site gathers rows of nall[Z] values or single
 * values nall[Z] apart -- halo_swap.c:1074-1274 packs the same sets) */

__global__ __launch_bounds__(BLOCK)
void k_halo_pack_x(lbmi_kparam_t kp, int dir, lbmi_halo_sel_t sel,
		   const double * __restrict__ data,
		   double * __restrict__ buf_lo, double * __restrict__ buf_hi,
		   int blocked, int layer) {

  /* layer l of a swap of several layers: the (l+1)-th interior plane from
   * either end */
  int j = blockIdx.x*BLOCK + threadIdx.x;
  int psz = plane_size(kp, dir);
  if (j >= psz) return;
  int k = blockIdx.y;
  const int nh = kp.nhalo;
  if (k < sel.nhi) {
    size_t i = plane_site(kp, dir, j, nh + layer);
    buf_lo[(size_t) k*psz + j] = data[xaddr(kp, blocked, sel.hi[k], i)];
  }
  else {
    int kk = k - sel.nhi;
    size_t i = plane_site(kp, dir, j, nh + kp.nlocal[dir] - 1 - layer);
    buf_hi[(size_t) kk*psz + j] = data[xaddr(kp, blocked, sel.lo[kk], i)];
  }
}

/* buf_lo here is what arrived from the LOWER neighbour (its last interior
 * plane, components sel.lo) -> our low halo plane; buf_hi arrived from the
 * UPPER neighbour (its first interior plane, components sel.hi). */

__global__ __launch_bounds__(BLOCK)
void k_halo_unpack_x(lbmi_kparam_t kp, int dir, lbmi_halo_sel_t sel,
		     double * __restrict__ data,
		     const double * __restrict__ buf_lo,
		     const double * __restrict__ buf_hi, int blocked,
		     int layer) {

  int j = blockIdx.x*BLOCK + threadIdx.x;
  int psz = plane_size(kp, dir);
  if (j >= psz) return;
  int k = blockIdx.y;
  const int nh = kp.nhalo;
  /* blocked order: the sites past the last whole block (the very end of the
   * high halo plane, y/z halo rows nobody pulls from) do not exist */
  const size_t nfull = blocked ? ((size_t) kp.nsite/LBW)*LBW : (size_t) kp.nsite;
  if (k < sel.nlo) {
    size_t i = plane_site(kp, dir, j, nh - 1 - layer);
    if (i < nfull) data[xaddr(kp, blocked, sel.lo[k], i)] = buf_lo[(size_t) k*psz + j];
  }
  else {
    int kk = k - sel.nlo;
    size_t i = plane_site(kp, dir, j, nh + kp.nlocal[dir] + layer);
    if (i < nfull) data[xaddr(kp, blocked, sel.hi[kk], i)] = buf_hi[(size_t) kk*psz + j];
  }
}

/* ---- record stream (the distribution file format) -------------------------
 *
 * lb_io_aggr_pack with lb_write_buf (model.c:1385-1402, 1479-1510): one
 * record of nvel doubles (p order) per INTERIOR site, sites in (ic,jc,kc)
 * order, independent of the memory order. On the device this is an
 * SoA-with-halo <-> dense AoS transposition: a block takes RB consecutive
 * interior sites, moves them through an LDS tile so that BOTH sides are
 * coalesced (population-major reads of 512 B per wave, contiguous record
 * writes), 2*nvel*8 bytes per site of HBM traffic. */

enum {RB1 = 256, RB2 = 128};

__device__ __forceinline__
size_t interior_site(const lbmi_kparam_t & kp, long long ib) {
  /* ib-th interior site in (x, y, z) order -> site index */
  int nz = kp.nlocal[2], ny = kp.nlocal[1];
  long long xy = ib / nz;
  int z = (int) (ib - xy*nz);
  int x = (int) (xy / ny);
  int y = (int) (xy - (long long) x*ny);
  return (size_t) (x + kp.nhalo)*kp.strx + (size_t) (y + kp.nhalo)*kp.stry
    + (size_t) (z + kp.nhalo);
}

/* NVEL here = values per site = ndist*nvel (19, 27, 38, 54): the record of a
 * site is [n][p], which is the component order of f. RB sites per block
 * (128 for the two-distribution records: the tile stays below 64 KB). */
template <int NVEL, bool PACK, int RB>
__global__ __launch_bounds__(RB)
void k_records(lbmi_kparam_t kp, double * __restrict__ f,
	       double * __restrict__ rec, long long ninterior) {

  /* tile[site][p], row length NVEL + 1 doubles against bank conflicts */
  __shared__ double tile[RB*(NVEL + 1)];
  const size_t ns = (size_t) kp.nsite;
  const long long ib0 = (long long) blockIdx.x*RB;
  const int nhere = (int) ((ninterior - ib0 < RB) ? (ninterior - ib0) : RB);
  const int t = threadIdx.x;

  if constexpr (PACK) {
    if (t < nhere) {
      size_t i = interior_site(kp, ib0 + t);
      static_for<0, NVEL>([&](auto P) {
	tile[t*(NVEL + 1) + P] = f[ns*P + i];
      });
    }
    __syncthreads();
    for (int k = t; k < nhere*NVEL; k += RB) {
      int site = k / NVEL;
      int p = k - site*NVEL;
      rec[ib0*NVEL + k] = tile[site*(NVEL + 1) + p];
    }
  }
  else {
    for (int k = t; k < nhere*NVEL; k += RB) {
      int site = k / NVEL;
      int p = k - site*NVEL;
      tile[site*(NVEL + 1) + p] = rec[ib0*NVEL + k];
    }
    __syncthreads();
    if (t < nhere) {
      size_t i = interior_site(kp, ib0 + t);
      static_for<0, NVEL>([&](auto P) {
	f[ns*P + i] = tile[t*(NVEL + 1) + P];
      });
    }
  }
}

/* hydro_field_set (hydro.c:279-330: hydro_u_zero, hydro_f_zero): every site,
 * halo included, of an SoA field of ncomp components gets a constant */

__global__ __launch_bounds__(BLOCK)
void k_field_set(long long nsite, int ncomp, double * __restrict__ field,
		 double v0, double v1, double v2) {
  long long i = (long long) blockIdx.x*BLOCK + threadIdx.x;
  if (i >= nsite) return;
  field[i] = v0;
  if (ncomp > 1) field[nsite + i] = v1;
  if (ncomp > 2) field[2*nsite + i] = v2;
}

/* ---- symmetric free energy: gradients and thermodynamic force (row f2) -----
 *
 * k_grad_7pt: grad_3d_7pt_fluid_d2 (gradient_3d_7pt_fluid.c:232-320):
 *   grad_a = (phi(+e_a) - phi(-e_a))/2, delsq = 7-point Laplacian, for the
 *   sites 1-nextra .. nlocal+nextra (nextra = nhalo - 1).
 * k_symm_force<FROM_GRAD>: pth_stress_compute with fe_symm_str_v
 *   (phi_force_stress.c:171-300, symmetric.c:371-420) followed by
 *   pth_force_fluid_kernel_v (phi_force_colloid.c:324-480):
 *     P_ab = p0 d_ab + kappa d_a phi d_b phi,
 *     p0   = a phi^2/2 + 3 b phi^4/4 - kappa phi delsq - kappa |grad|^2/2,
 *     F_a -= sum_b [ (P_ab(+e_b) + P_ab)/2 - (P_ab(-e_b) + P_ab)/2 ],
 *   added to hydro->force at the interior sites.
 *   FROM_GRAD = true reads the grad/delsq arrays (drop-in for the reference's
 *   two stages without its 9-component stress array: 72 B/site never
 *   written or re-read); FROM_GRAD = false is the MI355X-native form: the
 *   force straight from phi (a 25-point stencil; phi is 137 MB at 256^3
 *   and lives in L2/Infinity Cache, so HBM sees 8 B read + 48 B RMW per
 *   site instead of ~270 B in the reference's three kernels).
 */

struct Symm { double a, b, kappa; };

__device__ __forceinline__
void symm_stress(const Symm & q, double phi, const double (&g)[3],
		 double delsq, double (&s)[3][3]) {
  double p0 = 0.5*q.a*phi*phi + 0.75*q.b*phi*phi*phi*phi - q.kappa*phi*delsq
    - 0.5*q.kappa*(g[0]*g[0] + g[1]*g[1] + g[2]*g[2]);
  for (int ia = 0; ia < 3; ia++) {
    for (int ib = 0; ib < 3; ib++) {
      s[ia][ib] = ((ia == ib) ? p0 : 0.0) + q.kappa*g[ia]*g[ib];
    }
  }
}

__device__ __forceinline__
void grad7(const double * __restrict__ phi, size_t j, int strx, int stry,
	   double (&g)[3], double & delsq) {
  double xp = phi[j + strx], xm = phi[j - strx];
  double yp = phi[j + stry], ym = phi[j - stry];
  double zp = phi[j + 1], zm = phi[j - 1];
  g[0] = 0.5*(xp - xm);
  g[1] = 0.5*(yp - ym);
  g[2] = 0.5*(zp - zm);
  delsq = xp + xm + yp + ym + zp + zm - 6.0*phi[j];
}

/* grad_3d_27pt_kernel, GRAD_DEL2 (gradient_3d_27pt_fluid.c:216-364):
 * grad_a = (1/18) sum of the nine differences across the 3x3 plane normal to
 * a, delsq = (1/9)(sum of the 26 neighbours - 26 phi); summation orders of
 * the reference (z fastest), so the result is the reference's bit for bit
 * when the compiler does not contract (it has nothing to contract here). */

__device__ __forceinline__
void grad27(const double * __restrict__ phi, size_t j, int strx, int stry,
	    double (&g)[3], double & delsq) {
  double v[3][3][3];
  static_for<0, 27>([&](auto N) {
    constexpr int n = N;
    constexpr int a = n/9, b = (n/3) % 3, c = n % 3;
    v[a][b][c] = phi[(ptrdiff_t) j + (a - 1)*strx + (b - 1)*stry + (c - 1)];
  });
  const double r9 = (1.0/9.0);
  double sx = 0.0, sy = 0.0, sz = 0.0, d2 = 0.0;
  static_for<0, 9>([&](auto N) {
    constexpr int n = N;
    constexpr int b = n/3, c = n % 3;
    sx += v[2][b][c]; sx -= v[0][b][c];
    sy += v[b][2][c]; sy -= v[b][0][c];
    sz += v[b][c][2]; sz -= v[b][c][0];
  });
  static_for<0, 27>([&](auto N) {
    constexpr int n = N;
    if constexpr (n != 13) d2 += v[n/9][(n/3) % 3][n % 3];
  });
  d2 -= 26.0*v[1][1][1];
  g[0] = 0.5*r9*sx;
  g[1] = 0.5*r9*sy;
  g[2] = 0.5*r9*sz;
  delsq = r9*d2;
}

template <int NPT>
__device__ __forceinline__
void grad_at(const double * __restrict__ phi, size_t j, int strx, int stry,
	     double (&g)[3], double & delsq) {
  if constexpr (NPT == 27) grad27(phi, j, strx, stry, g, delsq);
  else grad7(phi, j, strx, stry, g, delsq);
}

/* Advective flux through the face between the sites l and r = l + s, face
 * velocity uf (advection_x, advection.c:433-482): order 1 upwind (:542-640),
 * 2 mean (:790-916), 3 three-point upwind-biased (:977-1176), 4 four-point
 * centred (:1188-1296). west selects the form the reference uses for the
 * "west" face of a site (it differs only in the side a zero velocity takes).
 * The order is uniform over the launch. */

__device__ __forceinline__
double adv_flux(int order, bool west, double uf, double pm, double pl,
		double pr, double pp) {
  /* pm, pl | pr, pp: phi at l - s, l, r, r + s */
  if (order == 1) {
    if (west) return uf*((uf > 0.0) ? pl : pr);
    return uf*((uf < 0.0) ? pr : pl);
  }
  if (order == 2) return uf*0.5*(pl + pr);
  if (order == 3) {
    const double a1 = -0.213933;
    const double a2 =  0.927865;
    const double a3 =  0.286067;
    const bool down = west ? !(uf > 0.0) : (uf < 0.0);
    if (down) return uf*(a1*pp + a2*pr + a3*pl);
    return uf*(a1*pm + a2*pl + a3*pr);
  }
  const double a1 = (1.0/16.0);
  const double a2 = (9.0/16.0);
  return uf*(- a1*pm + a2*pl + a2*pr - a1*pp);
}

template <int NPT>
__global__ __launch_bounds__(BLOCK)
void k_grad(lbmi_kparam_t kp, const double * __restrict__ phi,
	    double * __restrict__ grad, double * __restrict__ delsq,
	    int i0, int i1, unsigned nblk) {
  /* XCD-aware block order: neighbouring blocks (which share the y+-1 rows
   * and x+-1 planes of phi) on the same XCD, i.e. behind the same L2 */
  unsigned lb;
  if (!logical_block(nblk, lb, (unsigned) kp.xcd_group)) return;
  int i = (i0/LBMI_ALIGN)*LBMI_ALIGN + (int) (lb*BLOCK + threadIdx.x);
  if (i < i0 || i >= i1) return;
  Site s = decode(kp, i);
  const int ne = kp.nhalo - 1;
  const int nh = kp.nhalo;
  if (s.y < nh - ne || s.y >= nh + kp.nlocal[1] + ne) return;
  if (s.z < nh - ne || s.z >= nh + kp.nlocal[2] + ne) return;
  const size_t ns = (size_t) kp.nsite;
  double g[3], d2;
  grad_at<NPT>(phi, (size_t) i, kp.strx, kp.stry, g, d2);
  grad[i] = g[0];
  grad[ns + i] = g[1];
  grad[2*ns + i] = g[2];
  delsq[i] = d2;
}

template <bool FROM_GRAD, int NPT>
__global__ __launch_bounds__(BLOCK)
void k_symm_force(lbmi_kparam_t kp, Symm q, const double * __restrict__ phi,
		  const double * __restrict__ grad,
		  const double * __restrict__ delsq,
		  double * __restrict__ force, int i0, int i1, unsigned nblk) {

  unsigned lb;
  if (!logical_block(nblk, lb, (unsigned) kp.xcd_group)) return;
  int i = (i0/LBMI_ALIGN)*LBMI_ALIGN + (int) (lb*BLOCK + threadIdx.x);
  if (i < i0 || i >= i1) return;
  Site s = decode(kp, i);
  if (!s.interior) return;

  const size_t ns = (size_t) kp.nsite;
  const int str[3] = {kp.strx, kp.stry, 1};

  auto stress_at = [&](size_t j, double (&st)[3][3]) {
    double g[3], d2;
    if constexpr (FROM_GRAD) {
      g[0] = grad[j]; g[1] = grad[ns + j]; g[2] = grad[2*ns + j];
      d2 = delsq[j];
    }
    else {
      grad_at<NPT>(phi, j, kp.strx, kp.stry, g, d2);
    }
    symm_stress(q, phi[j], g, d2, st);
  };

  double pth0[3][3], pth1[3][3];
  double f[3] = {0.0, 0.0, 0.0};
  stress_at((size_t) i, pth0);

  static_for<0, 3>([&](auto D) {
    constexpr int id = D;
    stress_at((size_t) (i + str[id]), pth1);
    for (int ia = 0; ia < 3; ia++) f[ia] -= 0.5*(pth1[ia][id] + pth0[ia][id]);
    stress_at((size_t) (i - str[id]), pth1);
    for (int ia = 0; ia < 3; ia++) f[ia] += 0.5*(pth1[ia][id] + pth0[ia][id]);
  });

  force[i] += f[0];
  force[ns + i] += f[1];
  force[2*ns + i] += f[2];
}

/* k_cahn_hilliard<FROM_DELSQ>: one Cahn-Hilliard step of the symmetric
 * binary fluid as phi_cahn_hilliard runs it without noise, walls or LE planes
 * (phi_cahn_hilliard.c:195-284): first-order upwind advective fluxes
 * (advection.c:542-640), diffusive fluxes -M (mu_1 - mu_0) with
 * mu = a phi + b phi^3 - kappa delsq (phi_cahn_hilliard.c:349-402,
 * symmetric.c:303-316) and the forward step (:1026-1060). The reference
 * stores four flux arrays (fw, fe, fy, fz: 32 B/site written and re-read)
 * and updates phi in place; here each site evaluates its six face fluxes
 * in registers and writes phi_out (phi_out != phi), 8 B + 24 B (u) read and
 * 8 B written per site. FROM_DELSQ = false takes delsq from phi itself. */

template <bool FROM_DELSQ, int NPT>
__global__ __launch_bounds__(BLOCK)
void k_cahn_hilliard(lbmi_kparam_t kp, Symm q, double mobility, int order,
		     const double * __restrict__ phi,
		     const double * __restrict__ delsq,
		     const double * __restrict__ u,
		     double * __restrict__ phi_out, int i0, int i1,
		     unsigned nblk) {

  unsigned lb;
  if (!logical_block(nblk, lb, (unsigned) kp.xcd_group)) return;
  int i = (i0/LBMI_ALIGN)*LBMI_ALIGN + (int) (lb*BLOCK + threadIdx.x);
  if (i < i0 || i >= i1) return;
  Site s = decode(kp, i);
  if (!s.interior) return;

  const size_t ns = (size_t) kp.nsite;
  const int str[3] = {kp.strx, kp.stry, 1};

  auto mu_at = [&](size_t j) {
    double d2;
    if constexpr (FROM_DELSQ) {
      d2 = delsq[j];
    }
    else {
      double g[3];
      grad_at<NPT>(phi, j, kp.strx, kp.stry, g, d2);
    }
    double ph = phi[j];
    return q.a*ph + q.b*ph*ph*ph - q.kappa*d2;
  };

  const double phi0 = phi[i];
  const double mu0 = mu_at((size_t) i);
  double fhi[3], flo[3];               /* fluxes through the +d and -d faces */

  static_for<0, 3>([&](auto D) {
    constexpr int id = D;
    const double ud0 = u[ns*id + i];
    /* phi along the line through the site; two sites away only for the
     * schemes that reach there (uniform branch) */
    const double pm1 = phi[i - str[id]], pp1 = phi[i + str[id]];
    double pm2 = 0.0, pp2 = 0.0;
    if (order > 2) {
      pm2 = phi[i - 2*str[id]];
      pp2 = phi[i + 2*str[id]];
    }
    {
      size_t j = (size_t) (i + str[id]);       /* face (i, i + e_d) */
      double uf = 0.5*(ud0 + u[ns*id + j]);
      double f = adv_flux(order, false, uf, pm1, phi0, pp1, pp2);
      f -= mobility*(mu_at(j) - mu0);
      fhi[id] = f;
    }
    {
      /* x: the "west" flux of this site; y, z: the +d flux of the site
       * below, as the reference's update reads them */
      size_t j = (size_t) (i - str[id]);       /* face (i - e_d, i) */
      double uf = 0.5*(ud0 + u[ns*id + j]);
      double f = adv_flux(order, id == 0, uf, pm2, pm1, phi0, pp1);
      f -= mobility*(mu0 - mu_at(j));
      flo[id] = f;
    }
  });

  const double wz = (kp.nlocal[2] == 1) ? 0.0 : 1.0;
  phi_out[i] = phi0 - (+ fhi[0] - flo[0] + fhi[1] - flo[1]
		       + wz*fhi[2] - wz*flo[2]);
}

/* k_symm_fe_step: thermodynamic force AND Cahn-Hilliard step of the symmetric
 * binary fluid in one pass over phi (k_symm_force<false> + k_cahn_hilliard
 * <false> share the seven (grad, delsq) evaluations around a site): reads
 * phi (25-point, cache resident) and u, adds F to force, writes phi_out.
 * u is the velocity of the previous collision with a valid halo; the force
 * does not depend on u and the update of phi does not depend on the force,
 * so the two results are exactly those of the separate kernels. */

/* One site of the fused pass. phi may be the global array (ip = i, strides
 * of the lattice) or an LDS tile (ip = index in the tile, its strides);
 * grad, delsq (NPT = 0 only), u, force, phi_out are global, indexed by i. */

/* up[d], um[d]: the global site offsets to the +d and -d neighbours for the
 * reads of u (the strides, or, for a kernel that wraps the periodic box by
 * index instead of relying on a halo swap of u, the wrapped offsets) */

template <bool ACCUMULATE, int NPT>
__device__ __forceinline__
void fe_step_site(const lbmi_kparam_t & kp, const Symm & q, double mobility,
		  int order, const double * __restrict__ phi, int ip, int psx,
		  int psy, const double * __restrict__ grad,
		  const double * __restrict__ delsq,
		  const double * __restrict__ u, double * __restrict__ force,
		  double * __restrict__ phi_out, int i, const int (&up)[3],
		  const int (&um)[3]) {

  const size_t ns = (size_t) kp.nsite;
  const int pstr[3] = {psx, psy, 1};

  /* stress and chemical potential at a site: from phi alone (NPT = 7, 27)
   * or from the gradient arrays (NPT = 0) */
  auto at = [&](int jp, size_t j, double (&st)[3][3], double & mu, double & ph) {
    double g[3], d2;
    if constexpr (NPT == 0) {
      g[0] = grad[j]; g[1] = grad[ns + j]; g[2] = grad[2*ns + j];
      d2 = delsq[j];
    }
    else {
      grad_at<NPT>(phi, (size_t) jp, psx, psy, g, d2);
    }
    ph = phi[jp];
    symm_stress(q, ph, g, d2, st);
    mu = q.a*ph + q.b*ph*ph*ph - q.kappa*d2;
  };

  double pth0[3][3], pth1[3][3];
  double mu0, mu1, phi0, phi1;
  double f[3] = {0.0, 0.0, 0.0};
  double fhi[3], flo[3];
  at(ip, (size_t) i, pth0, mu0, phi0);

  static_for<0, 3>([&](auto D) {
    constexpr int id = D;
    const double ud0 = u[ns*id + i];
    double pm2 = 0.0, pp2 = 0.0;
    if (order > 2) {
      pm2 = phi[ip - 2*pstr[id]];
      pp2 = phi[ip + 2*pstr[id]];
    }
    const double pm1 = phi[ip - pstr[id]], pp1 = phi[ip + pstr[id]];
    {
      size_t j = (size_t) (i + up[id]);
      at(ip + pstr[id], j, pth1, mu1, phi1);
      for (int ia = 0; ia < 3; ia++) f[ia] -= 0.5*(pth1[ia][id] + pth0[ia][id]);
      double uf = 0.5*(ud0 + u[ns*id + j]);
      double fl = adv_flux(order, false, uf, pm1, phi0, pp1, pp2);
      fl -= mobility*(mu1 - mu0);
      fhi[id] = fl;
    }
    {
      size_t j = (size_t) (i + um[id]);
      at(ip - pstr[id], j, pth1, mu1, phi1);
      for (int ia = 0; ia < 3; ia++) f[ia] += 0.5*(pth1[ia][id] + pth0[ia][id]);
      double uf = 0.5*(ud0 + u[ns*id + j]);
      double fl = adv_flux(order, id == 0, uf, pm2, pm1, phi0, pp1);
      fl -= mobility*(mu0 - mu1);
      flo[id] = fl;
    }
  });

  if constexpr (ACCUMULATE) {
    force[i] += f[0];
    force[ns + i] += f[1];
    force[2*ns + i] += f[2];
  }
  else {
    /* hydro_f_zero + add in one store: the caller asserts that nothing
     * else has contributed to the force field this step */
    force[i] = f[0];
    force[ns + i] = f[1];
    force[2*ns + i] = f[2];
  }

  const double wz = (kp.nlocal[2] == 1) ? 0.0 : 1.0;
  phi_out[i] = phi0 - (+ fhi[0] - flo[0] + fhi[1] - flo[1]
		       + wz*fhi[2] - wz*flo[2]);
}

template <bool ACCUMULATE, int NPT>
__global__ __launch_bounds__(BLOCK)
void k_symm_fe_step(lbmi_kparam_t kp, Symm q, double mobility, int order,
		    const double * __restrict__ phi,
		    const double * __restrict__ grad,
		    const double * __restrict__ delsq,
		    const double * __restrict__ u,
		    double * __restrict__ force,
		    double * __restrict__ phi_out, int i0, int i1,
		    unsigned nblk) {

  unsigned lb;
  if (!logical_block(nblk, lb, (unsigned) kp.xcd_group)) return;
  int i = (i0/LBMI_ALIGN)*LBMI_ALIGN + (int) (lb*BLOCK + threadIdx.x);
  if (i < i0 || i >= i1) return;
  Site s = decode(kp, i);
  if (!s.interior) return;

  const int up[3] = {kp.strx, kp.stry, 1};
  const int um[3] = {-kp.strx, -kp.stry, -1};
  fe_step_site<ACCUMULATE, NPT>(kp, q, mobility, order, phi, i, kp.strx,
				kp.stry, grad, delsq, u, force, phi_out, i,
				up, um);
}

/* The same pass with phi staged through LDS: a block of 256 threads owns a
 * tile of FT_X x FT_Y x FT_Z sites, loads phi for the tile and the two
 * layers around it once (the stencil of the pass: gradients at the site and
 * its six neighbours, advection up to two sites away), and every thread
 * then works FT_X sites from the tile. The 49 (7-point) phi reads per site
 * become LDS reads; HBM and L2 see phi ~3.4 times per site instead of being
 * asked 25 distinct addresses per site. u, force and phi_out stay global. */

enum {FT_X = 4, FT_Y = 8, FT_Z = 32, FT_H = 2,
      FT_LX = FT_X + 2*FT_H, FT_LY = FT_Y + 2*FT_H, FT_LZ = FT_Z + 2*FT_H};

/* WRAP: the periodic box is wrapped by index -- the tile takes phi of the
 * halo layers from the periodic images inside the domain and u of a
 * neighbour across a face from the opposite face -- so that neither phi nor
 * u needs a halo swap beforehand (single rank). */

/* What the force and the fluxes need from one site: grad phi, the isotropic
 * part p0 of the pressure tensor and the chemical potential (symmetric.c:
 * 371-420: P_ab = p0 delta_ab + kappa d_a phi d_b phi, mu = a phi + b phi^3 -
 * kappa delsq phi), from the phi tile */

struct FeSite {
  double g[3];
  double p0;
  double mu;
};

template <int NPT>
__device__ __forceinline__
FeSite fe_eval(const Symm & q, const double * __restrict__ tile, int lp) {
  FeSite e;
  double d2;
  grad_at<NPT>(tile, (size_t) lp, FT_LY*FT_LZ, FT_LZ, e.g, d2);
  const double ph = tile[lp];
  e.p0 = 0.5*q.a*ph*ph + 0.75*q.b*ph*ph*ph*ph - q.kappa*ph*d2
    - 0.5*q.kappa*(e.g[0]*e.g[0] + e.g[1]*e.g[1] + e.g[2]*e.g[2]);
  e.mu = q.a*ph + q.b*ph*ph*ph - q.kappa*d2;
  return e;
}

template <bool ACCUMULATE, int NPT, bool WRAP>
__global__ __launch_bounds__(FT_Y*FT_Z)
void k_symm_fe_step_tiled(lbmi_kparam_t kp, Symm q, double mobility, int order,
			  const double * __restrict__ phi,
			  const double * __restrict__ u,
			  double * __restrict__ force,
			  double * __restrict__ phi_out, int nty, int ntz,
			  unsigned nblk) {

  __shared__ double tile[FT_LX*FT_LY*FT_LZ];

  unsigned lb;
  if (!logical_block(nblk, lb, (unsigned) kp.xcd_group)) return;
  const int bz = (int) (lb % (unsigned) ntz);
  const int by = (int) ((lb / (unsigned) ntz) % (unsigned) nty);
  const int bx = (int) (lb / (unsigned) (ntz*nty));
  const int nh = kp.nhalo;
  const int x0 = nh + bx*FT_X, y0 = nh + by*FT_Y, z0 = nh + bz*FT_Z;

  /* This thread's column, and the velocities of its first site: issued before
   * the tile is loaded, so that their latency is behind the loader's (the
   * pass is bound by the latency of dependent loads, not by HBM or FP64
   * issue: profiles/r02_fe_pass_counters.txt) */
  const int tz = (int) threadIdx.x % FT_Z, ty = (int) threadIdx.x / FT_Z;
  const int gy = y0 + ty, gz = z0 + tz;
  const bool mine = (gy < nh + kp.nlocal[1] && gz < nh + kp.nlocal[2]);
  const size_t ns = (size_t) kp.nsite;
  int up[3] = {kp.strx, kp.stry, 1};
  int um[3] = {-kp.strx, -kp.stry, -1};
  if constexpr (WRAP) {
    if (gy == nh + kp.nlocal[1] - 1) up[1] = -(kp.nlocal[1] - 1)*kp.stry;
    if (gy == nh) um[1] = (kp.nlocal[1] - 1)*kp.stry;
    if (gz == nh + kp.nlocal[2] - 1) up[2] = -(kp.nlocal[2] - 1);
    if (gz == nh) um[2] = kp.nlocal[2] - 1;
  }
  /* un[id]: u_id at the site, at its + and at its - neighbour in direction id */
  double un[3][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
  auto load_u = [&](int gx) {
    const int i = gx*kp.strx + gy*kp.stry + gz;
    if constexpr (WRAP) {
      up[0] = (gx == nh + kp.nlocal[0] - 1) ? -(kp.nlocal[0] - 1)*kp.strx : kp.strx;
      um[0] = (gx == nh) ? (kp.nlocal[0] - 1)*kp.strx : -kp.strx;
    }
    static_for<0, 3>([&](auto D) {
      constexpr int id = D;
      un[id][0] = u[ns*id + i];
      un[id][1] = u[ns*id + (size_t) (i + up[id])];
      un[id][2] = u[ns*id + (size_t) (i + um[id])];
    });
  };
  /* (27-point gradients: the evaluations already take every register there
   * is; requesting the velocities a site ahead costs more than it hides) */
  constexpr bool AHEAD = (NPT == 7);
  if (AHEAD && mine && x0 < nh + kp.nlocal[0]) load_u(x0);

  /* The tile, a row of FT_LZ values per thread group: the thread's z column
   * and its (at most two) rows are fixed, so that the loop over the x planes
   * carries no division and the wrap of x is scalar work. */
  {
    enum {ROWS = (FT_Y*FT_Z)/FT_LZ};                 /* rows loaded at a time */
    const int lz = (int) threadIdx.x % FT_LZ;
    const int r0 = (int) threadIdx.x / FT_LZ;        /* ROWS*FT_LZ <= threads */
    int gz = z0 - FT_H + lz;
    bool okz = (r0 < ROWS) && (gz < kp.nall[2]);
    if constexpr (WRAP) {
      /* a halo coordinate (at most nhalo <= nlocal away) -> its image */
      if (gz < nh) gz += kp.nlocal[2];
      else if (gz >= nh + kp.nlocal[2]) gz -= kp.nlocal[2];
    }
    for (int ly = r0; ly < FT_LY; ly += ROWS) {
      int gy = y0 - FT_H + ly;
      const bool ok = okz && (gy < kp.nall[1]);
      if constexpr (WRAP) {
	if (gy < nh) gy += kp.nlocal[1];
	else if (gy >= nh + kp.nlocal[1]) gy -= kp.nlocal[1];
      }
      const size_t rowoff = (size_t) gy*kp.stry + gz;
      for (int lx = 0; lx < FT_LX; lx++) {
	int gx = x0 - FT_H + lx;
	const bool okx = (gx < kp.nall[0]);
	if constexpr (WRAP) {
	  if (gx < nh) gx += kp.nlocal[0];
	  else if (gx >= nh + kp.nlocal[0]) gx -= kp.nlocal[0];
	}
	if (r0 < ROWS) {
	  tile[(lx*FT_LY + ly)*FT_LZ + lz] =
	    (ok && okx) ? phi[(size_t) gx*kp.strx + rowoff] : 0.0;
	}
      }
    }
  }
  __syncthreads();

  if (!mine) return;

  constexpr int pstr[3] = {FT_LY*FT_LZ, FT_LZ, 1};
  const double wz = (kp.nlocal[2] == 1) ? 0.0 : 1.0;

  /* the thread walks FT_X sites along x: what it evaluated at x + 1 is the
   * centre of the next site and the x - 1 neighbour of the one after, so a
   * site costs five evaluations (x + 1, y +- 1, z +- 1), not seven */
  int ip = ((0 + FT_H)*FT_LY + (ty + FT_H))*FT_LZ + (tz + FT_H);
  FeSite em = fe_eval<NPT>(q, tile, ip - pstr[0]);
  FeSite ec = fe_eval<NPT>(q, tile, ip);

  for (int tx = 0; tx < FT_X; tx++, ip += pstr[0]) {
    const int gx = x0 + tx;
    if (gx >= nh + kp.nlocal[0]) break;
    const int i = gx*kp.strx + gy*kp.stry + gz;
    /* the velocities of this site have arrived by now; those of the next
     * one are requested before anything is computed with these */
    double uc[3][3];
    if constexpr (!AHEAD) load_u(gx);
    static_for<0, 9>([&](auto N) { uc[N/3][N % 3] = un[N/3][N % 3]; });
    if (AHEAD && tx + 1 < FT_X && gx + 1 < nh + kp.nlocal[0]) load_u(gx + 1);
    const FeSite ep = fe_eval<NPT>(q, tile, ip + pstr[0]);
    const double phi0 = tile[ip];
    double f[3] = {0.0, 0.0, 0.0};
    double fhi[3], flo[3];

    static_for<0, 3>([&](auto D) {
      constexpr int id = D;
      FeSite hi, lo;
      if constexpr (id == 0) {
	hi = ep;
	lo = em;
      }
      else {
	hi = fe_eval<NPT>(q, tile, ip + pstr[id]);
	lo = fe_eval<NPT>(q, tile, ip - pstr[id]);
      }
      /* F_a = -d_b P_ab: -(1/2)(P(+) + P(0)) + (1/2)(P(-) + P(0)), column id
       * (phi_force.c / pth_force_fluid_kernel, the reference's association) */
      static_for<0, 3>([&](auto A) {
	constexpr int ia = A;
	const double c0 = ((ia == id) ? ec.p0 : 0.0) + q.kappa*ec.g[ia]*ec.g[id];
	const double cp = ((ia == id) ? hi.p0 : 0.0) + q.kappa*hi.g[ia]*hi.g[id];
	const double cm = ((ia == id) ? lo.p0 : 0.0) + q.kappa*lo.g[ia]*lo.g[id];
	f[ia] -= 0.5*(cp + c0);
	f[ia] += 0.5*(cm + c0);
      });
      const double ud0 = uc[id][0];
      double pm2 = 0.0, pp2 = 0.0;
      if (order > 2) {
	pm2 = tile[ip - 2*pstr[id]];
	pp2 = tile[ip + 2*pstr[id]];
      }
      const double pm1 = tile[ip - pstr[id]], pp1 = tile[ip + pstr[id]];
      {
	const double uf = 0.5*(ud0 + uc[id][1]);
	double fl = adv_flux(order, false, uf, pm1, phi0, pp1, pp2);
	fl -= mobility*(hi.mu - ec.mu);
	fhi[id] = fl;
      }
      {
	const double uf = 0.5*(ud0 + uc[id][2]);
	double fl = adv_flux(order, id == 0, uf, pm2, pm1, phi0, pp1);
	fl -= mobility*(ec.mu - lo.mu);
	flo[id] = fl;
      }
    });

    if constexpr (ACCUMULATE) {
      force[i] += f[0];
      force[ns + i] += f[1];
      force[2*ns + i] += f[2];
    }
    else {
      /* hydro_f_zero + add in one store: the caller asserts that nothing
       * else has contributed to the force field this step */
      force[i] = f[0];
      force[ns + i] = f[1];
      force[2*ns + i] = f[2];
    }
    phi_out[i] = phi0 - (+ fhi[0] - flo[0] + fhi[1] - flo[1]
			 + wz*fhi[2] - wz*flo[2]);
    em = ec;
    ec = ep;
  }
}

/* ---- k_symm_lb_step: the whole binary-fluid step of BASELINE config 4 in ONE pass ----
 *
 * ludwig.c:537-860 with free_energy symmetric (finite difference), per site:
 *   hydro_f_zero, phi_force_calculation   F = -d_b P_ab(phi_t)      (phi_force.c:74-136)
 *   phi_cahn_hilliard                     phi_t+1 from phi_t, u_t-1 (phi_cahn_hilliard.c:206-284)
 *   lb_collide (+ lb_halo, lb_propagation of the step before, as k_propagate_collide)
 * The separate passes move, beside the 304 B of the distributions, the force
 * twice (24 B written, 24 B read) and u twice (24 B written by the collision,
 * 24 B read by the advection), phi twice: 424 B per site and step, and the
 * free-energy pass is bound by the latency of its dependent loads, not by
 * HBM (profiles/r02_fe_pass_counters.txt). Here the thermodynamic force never
 * leaves the registers: the thread that collides a site has just evaluated
 * it from the 25 values of phi around the site (|dx| + |dy| + |dz| <= 2, from
 * L2: neighbouring lanes share all but one of them), and the same values give
 * the six Cahn-Hilliard fluxes. Their latency hides behind the 19 pulls of the
 * distributions, which are in flight meanwhile. 304 + 32 (rho, u) + 24 (u of
 * the previous step) + 16 (phi in, out) = 376 B per site.
 *
 * The advection reads u_t-1 of the six neighbours while the collisions of this
 * launch write u_t: two arrays, u_prev (read) and h.u (written), swapped by the
 * caller like phi and phi_out. Single rank: every access across a periodic
 * face goes to the image inside the domain (no halo of f, phi or u needed).
 * Arithmetic: that of k_symm_fe_step_tiled<false, 7, true> and of the
 * collision, expression for expression. */

__host__ __device__ constexpr int iabs_c(int a) { return (a < 0) ? -a : a; }

/* position of (dx, dy, dz) among the 25 points, and its inverse */
__host__ __device__ constexpr int p25(int dx, int dy, int dz) {
  int n = 0;
  for (int a = -2; a <= 2; a++) {
    for (int b = -2; b <= 2; b++) {
      for (int c = -2; c <= 2; c++) {
	if (iabs_c(a) + iabs_c(b) + iabs_c(c) > 2) continue;
	if (a == dx && b == dy && c == dz) return n;
	n += 1;
      }
    }
  }
  return -1;
}

__host__ __device__ constexpr int q25(int m, int comp) {
  int n = 0;
  for (int a = -2; a <= 2; a++) {
    for (int b = -2; b <= 2; b++) {
      for (int c = -2; c <= 2; c++) {
	if (iabs_c(a) + iabs_c(b) + iabs_c(c) > 2) continue;
	if (n == m) return (comp == 0) ? a : ((comp == 1) ? b : c);
	n += 1;
      }
    }
  }
  return 0;
}

static_assert(p25(2, 0, 0) == 24 && p25(0, 0, 0) == 12 && q25(24, 0) == 2, "25 points");

/* fe_eval at the site (OX, OY, OZ) away, from the 25 values in registers
 * (7-point gradients: grad7 + fe_eval, same expressions) */
template <int OX, int OY, int OZ>
__device__ __forceinline__
FeSite fe_eval_reg(const Symm & q, const double (&ph)[25]) {
#pragma clang fp contract(fast)
  constexpr int ic = p25(OX, OY, OZ);
  constexpr int ixp = p25(OX + 1, OY, OZ), ixm = p25(OX - 1, OY, OZ);
  constexpr int iyp = p25(OX, OY + 1, OZ), iym = p25(OX, OY - 1, OZ);
  constexpr int izp = p25(OX, OY, OZ + 1), izm = p25(OX, OY, OZ - 1);
  static_assert(ic >= 0 && ixp >= 0 && ixm >= 0 && iyp >= 0 && iym >= 0 &&
		izp >= 0 && izm >= 0, "site not inside the 25 points");
  FeSite e;
  const double xp = ph[ixp], xm = ph[ixm];
  const double yp = ph[iyp], ym = ph[iym];
  const double zp = ph[izp], zm = ph[izm];
  e.g[0] = 0.5*(xp - xm);
  e.g[1] = 0.5*(yp - ym);
  e.g[2] = 0.5*(zp - zm);
  const double d2 = xp + xm + yp + ym + zp + zm - 6.0*ph[ic];
  const double p = ph[ic];
  e.p0 = 0.5*q.a*p*p + 0.75*q.b*p*p*p*p - q.kappa*p*d2
    - 0.5*q.kappa*(e.g[0]*e.g[0] + e.g[1]*e.g[1] + e.g[2]*e.g[2]);
  e.mu = q.a*p + q.b*p*p*p - q.kappa*d2;
  return e;
}

#ifndef LBMI_FEWAVES
#define LBMI_FEWAVES 1      /* k_symm_lb_step: __launch_bounds__ min waves per SIMD */
#endif

template <int NVEL, int SCHEME, int LAY>
__global__ __launch_bounds__(BLOCK, LBMI_FEWAVES)
void k_symm_lb_step(lbmi_kparam_t kp, const double * __restrict__ f,
		    double * __restrict__ fp, lbmi_hydro_dev_t h, Symm q,
		    double mobility, int order,
		    const double * __restrict__ phi,
		    const double * __restrict__ uprev,
		    double * __restrict__ phi_out, int i0, int i1,
		    unsigned nblk, int gstripe) {

  static_assert(SPT == 1, "one site per thread");
  constexpr int ORD = LAY & 3;
  int i;
  if (gstripe > 0) {
    /* Stripes: the hardware hands block b to XCD b mod 8; here XCD k works on
     * the k-th eighth of EVERY x plane (gstripe blocks of it), plane after
     * plane. The 25 values of phi and the velocities a site needs from the
     * planes x - 2 .. x + 2 are then behind the L2 that has just had them, or
     * is about to: with blocks dealt out in runs of the 1-d site order every
     * XCD fetches its own copy of five planes per run (the counters showed
     * 1.29 x the algorithmic reads). The last stripe of a plane is short:
     * blocks past its end have nothing to do. (Measured: 3 - 6 % SLOWER --
     * the distribution streams want all XCDs in one neighbourhood of the 1-d
     * order more than phi wants one L2: lbmi_tune "fe_stripes", default 0.) */
    const int xcd = (int) (blockIdx.x & 7u), j = (int) (blockIdx.x >> 3);
    const int m = j/gstripe, w = j - m*gstripe;
    const int plane0 = i0 + m*kp.strx;
    i = plane0 + (xcd*gstripe + w)*BLOCK + (int) threadIdx.x;
    if (plane0 >= i1 || i >= plane0 + kp.strx) return;
  }
  else {
    unsigned lb;
    if (!logical_block(nblk, lb, (unsigned) kp.fe_xcd_group)) return;
    constexpr int ALIGNV = (ORD == 0) ? LBMI_ALIGN : LBW;
    i = (i0/ALIGNV)*ALIGNV + (int) (lb*BLOCK + threadIdx.x);
    if (i < i0 || i >= i1) return;
  }

  const lbmi_xbuf_t none = {nullptr, nullptr, nullptr, nullptr};
  const size_t ns = (size_t) kp.nsite;
  const Site s = decode(kp, i);

  /* the 25 values of phi and the nine of u first, then the pulls: the
   * free-energy arithmetic runs while the distributions are on their way */
  double ph[25];
  double uc[3][3];
  int up[3] = {0, 0, 0}, um[3] = {0, 0, 0};
  if (s.interior) {
    /* offsets (in sites) of the neighbours at distance -2 .. +2 in each
     * direction, across a periodic face: the image inside the domain */
    int ox[5], oy[5], oz[5];
    auto offsets = [&](int c, int n, int str, int (&o)[5]) {
      /* c: 0 .. n - 1, the coordinate inside the local domain (n >= 4) */
      static_for<0, 5>([&](auto K) {
	constexpr int k = K - 2;
	int t = k;
	if (c + k < 0) t += n;
	else if (c + k >= n) t -= n;
	o[K] = t*str;
      });
    };
    offsets(s.x - kp.nhalo, kp.nlocal[0], kp.strx, ox);
    offsets(s.y - kp.nhalo, kp.nlocal[1], kp.stry, oy);
    offsets(s.z - kp.nhalo, kp.nlocal[2], 1, oz);
    static_for<0, 25>([&](auto N) {
      constexpr int m = N;
      constexpr int a = q25(m, 0), b = q25(m, 1), c = q25(m, 2);
      ph[m] = phi[i + ox[a + 2] + oy[b + 2] + oz[c + 2]];
    });
    up[0] = ox[3]; up[1] = oy[3]; up[2] = oz[3];
    um[0] = ox[1]; um[1] = oy[1]; um[2] = oz[1];
  }

  PulledSite<NVEL> ps;
  pc_pull<NVEL, true, ORD == 2, false>(kp, f, 7, i, ps, none);

  if (s.interior) {
    /* uc[id][0..2]: u_id at the site, at its + and at its - neighbour in id:
     * requested last (the fluxes are the last thing evaluated, and nine
     * values fewer are alive while the stresses are) */
    static_for<0, 3>([&](auto D) {
      constexpr int id = D;
      uc[id][0] = uprev[ns*id + i];
      uc[id][1] = uprev[ns*id + (size_t) (i + up[id])];
      uc[id][2] = uprev[ns*id + (size_t) (i + um[id])];
    });
  }

  double frc[3] = {0.0, 0.0, 0.0};
  if (s.interior) {
#pragma clang fp contract(fast)
    const FeSite ec = fe_eval_reg<0, 0, 0>(q, ph);
    constexpr int icentre = p25(0, 0, 0);
    const double phi0 = ph[icentre];
    double fhi[3], flo[3];
    static_for<0, 3>([&](auto D) {
      constexpr int id = D;
      constexpr int ex = (id == 0), ey = (id == 1), ez = (id == 2);
      const FeSite hi = fe_eval_reg<ex, ey, ez>(q, ph);
      const FeSite lo = fe_eval_reg<-ex, -ey, -ez>(q, ph);
      /* F_a = -d_b P_ab, column id: pth_force_fluid_kernel forms
       * -(P(+) + P(0))/2 + (P(-) + P(0))/2; the stress of the site itself
       * cancels, and is not evaluated here (the difference to the reference's
       * association is a rounding of |P|, 1e-16 of a stress whose divergence
       * is taken: far inside the 1e-12 of the parity criterion) */
      static_for<0, 3>([&](auto A) {
	constexpr int ia = A;
	const double cp = ((ia == id) ? hi.p0 : 0.0) + q.kappa*hi.g[ia]*hi.g[id];
	const double cm = ((ia == id) ? lo.p0 : 0.0) + q.kappa*lo.g[ia]*lo.g[id];
	frc[ia] += 0.5*(cm - cp);
      });
      constexpr int im2 = p25(-2*ex, -2*ey, -2*ez), ip2 = p25(2*ex, 2*ey, 2*ez);
      constexpr int im1 = p25(-ex, -ey, -ez), ip1 = p25(ex, ey, ez);
      const double ud0 = uc[id][0];
      double pm2 = 0.0, pp2 = 0.0;
      if (order > 2) {
	pm2 = ph[im2];
	pp2 = ph[ip2];
      }
      const double pm1 = ph[im1], pp1 = ph[ip1];
      {
	const double uf = 0.5*(ud0 + uc[id][1]);
	double fl = adv_flux(order, false, uf, pm1, phi0, pp1, pp2);
	fl -= mobility*(hi.mu - ec.mu);
	fhi[id] = fl;
      }
      {
	const double uf = 0.5*(ud0 + uc[id][2]);
	double fl = adv_flux(order, id == 0, uf, pm2, pm1, phi0, pp1);
	fl -= mobility*(ec.mu - lo.mu);
	flo[id] = fl;
      }
    });
    phi_out[i] = phi0 - (+ fhi[0] - flo[0] + fhi[1] - flo[1]
			 + fhi[2] - flo[2]);
  }

  pc_collide_store<NVEL, SCHEME, ORD != 0, (LAY & 4) != 0, true, false, false, true>(
      kp, fp, h, i, ps, none, frc[0], frc[1], frc[2]);
}

/* ---- two distributions: the symmetric_lb step (row f4) ------------------------
 *
 * k_phi_from_g: phi_lb_to_field (phi_lb_coupler.c:39-112), phi = sum_p g_p.
 * k_collide_binary: lb_collision_mrt2_site (collision.c:720-1027) without
 * noise: the density distribution (n = 0) relaxes as in k_collide with the
 * thermodynamic stress of the symmetric free energy in the equilibrium
 * stress, at every interior site; the order-parameter distribution (n = 1)
 * is re-projected from phi, jphi (relaxed towards phi u at rtau2) and
 * sphi = phi u u + mu delta. f2[(n*NVEL + p)*nsite + i]. */

/* offset (in sites) of the source of population p for a pull at a site whose
 * wrap corrections are w (directions wrapped by index: the source of a pull
 * across a periodic face is the image inside the domain, not the halo) */
template <int NVEL, int P>
__device__ __forceinline__ int pull_offset(const lbmi_kparam_t & kp,
					    const WrapAdj & w) {
  using M = Model<NVEL>;
  constexpr int cx = M::c(P,0), cy = M::c(P,1), cz = M::c(P,2);
  int off = cx*kp.strx + cy*kp.stry + cz;
  if constexpr (cx ==  1) off -= w.wlo[0];
  if constexpr (cx == -1) off -= w.whi[0];
  if constexpr (cy ==  1) off -= w.wlo[1];
  if constexpr (cy == -1) off -= w.whi[1];
  if constexpr (cz ==  1) off -= w.wlo[2];
  if constexpr (cz == -1) off -= w.whi[2];
  return off;
}

/* ---- k_propagate_collide_face: the boundary planes of a slab along Y or Z ------------
 *
 * The FUSED step of such a slab (lbmi_fused_step): ONE launch of
 * k_propagate_collide over all interior x planes while the messages travel --
 * its results in the first and the last plane of the decomposed direction are
 * made from halo planes nobody has filled and do not count -- then this
 * kernel for those two planes: a thread per plane site, the populations that
 * cross the face taken straight from the receive buffers ([k][plane site], k
 * the rank of p among the populations with c_dim = +1 or -1, plane sites in
 * the order of the two remaining coordinates), everything else pulled from f
 * with the two local directions wrapped by index; it collides, overwrites
 * what the big launch left at the site, and leaves the populations the NEXT
 * exchange sends in the send buffers. The planes of Y and Z slabs are not
 * contiguous (rows of nall[Z] values; single values nall[Z] apart): the
 * accesses of this kernel are gathers, two planes' worth per step. */

template <int NVEL, int DIM>
__host__ __device__ constexpr int drank(int p, int c) {
  int k = 0;
  for (int q = 0; q < p; q++) {
    if (Model<NVEL>::c(q, DIM) == c) k += 1;
  }
  return k;
}

template <int NVEL, int SCHEME, int DIM, bool HIO>
__global__ __launch_bounds__(BLOCK)
void k_propagate_collide_face(lbmi_kparam_t kp, const double * __restrict__ f,
			      double * __restrict__ fp, lbmi_hydro_dev_t h,
			      int wrapmask, lbmi_xbuf_t xb) {

  using M = Model<NVEL>;
  const int psz = plane_size(kp, DIM);
  const int j = (int) (blockIdx.x*BLOCK + threadIdx.x);
  if (j >= psz) return;
  const int nh = kp.nhalo;
  const int nd = kp.nlocal[DIM];
  const int coord = (blockIdx.y == 0) ? nh : nh + nd - 1;
  const bool atlo = (coord == nh), athi = (coord == nh + nd - 1);
  const int i = (int) plane_site(kp, DIM, j, coord);
  Site s = decode(kp, i);
  if (!(s.interior && s.x >= nh && s.x < nh + kp.nlocal[0])) return;

  const size_t ns = (size_t) kp.nsite;
  const lbmi_xbuf_t none = {nullptr, nullptr, nullptr, nullptr};
  const WrapAdj w = wrap_adjust(kp, s, wrapmask);      /* the two local directions */
  const int c3[3] = {s.x, s.y, s.z};
  PulledSite<NVEL> ps;
  ps.s = s;

  static_for<0, NVEL>([&](auto P) {
    constexpr int p = P;
    constexpr int cd = M::c(p, DIM);
    bool buffered = false;
    if constexpr (cd != 0) {
      if ((cd == 1 && atlo) || (cd == -1 && athi)) {
	/* the source lies in the neighbour's plane: entry of the receive
	 * buffer at the source's two other coordinates (wrapped) */
	int o[3];
	static_for<0, 3>([&](auto D) {
	  constexpr int d = D;
	  int v = c3[d] - M::c(p, d);
	  if (d != DIM) {
	    if (v < nh) v += kp.nlocal[d];
	    else if (v >= nh + kp.nlocal[d]) v -= kp.nlocal[d];
	  }
	  o[d] = v;
	});
	int jj;
	if constexpr (DIM == 0) jj = o[1]*kp.nall[2] + o[2];
	else if constexpr (DIM == 1) jj = o[0]*kp.nall[2] + o[2];
	else jj = o[0]*kp.nall[1] + o[1];
	if constexpr (cd == 1) {
	  constexpr int k = drank<NVEL, DIM>(p, 1);
	  ps.fl[p] = xb.recvlo[(size_t) k*psz + jj];
	}
	else {
	  constexpr int k = drank<NVEL, DIM>(p, -1);
	  ps.fl[p] = xb.recvhi[(size_t) k*psz + jj];
	}
	buffered = true;
      }
    }
    if (!buffered) {
      const int off = pull_offset<NVEL, p>(kp, w);
      ps.fl[p] = f[ns*p + (i - off)];
    }
  });

  pc_collide_store<NVEL, SCHEME, false, false, HIO, false>(kp, fp, h, i, ps, none);

  /* what the next exchange sends: of the first plane the populations that
   * leave downwards, of the last one those that leave upwards */
  static_for<0, NVEL>([&](auto P) {
    constexpr int p = P;
    constexpr int cd = M::c(p, DIM);
    if constexpr (cd == -1) {
      constexpr int k = drank<NVEL, DIM>(p, -1);
      if (atlo) xb.sendlo[(size_t) k*psz + j] = ps.fl[p];
    }
    if constexpr (cd == 1) {
      constexpr int k = drank<NVEL, DIM>(p, 1);
      if (athi) xb.sendhi[(size_t) k*psz + j] = ps.fl[p];
    }
  });
}

/* RB: f2 is in the blocked order of a deferred two-distribution state,
 * [site/256][n*NVEL + p][site%256] (faddr with 2 NVEL components) */
template <int NVEL, bool PULL, bool RB = false>
__global__ __launch_bounds__(BLOCK)
void k_phi_from_g(lbmi_kparam_t kp, const double * __restrict__ f2,
		  double * __restrict__ phi, int wrapmask, int i0, int i1,
		  unsigned nblk) {
  unsigned lb;
  if (!logical_block(nblk, lb, (unsigned) kp.xcd_group)) return;
  int i = (i0/LBMI_ALIGN)*LBMI_ALIGN + (int) (lb*BLOCK + threadIdx.x);
  if (i < i0 || i >= i1) return;
  Site s = decode(kp, i);
  if (!s.interior) return;
  const size_t ns = (size_t) kp.nsite;
  double sum = 0.0;
  /* PULL: the propagation is pending: population p of this site still sits
   * at i - c_p of the post-collision array -- in its halo where that has
   * been swapped (FUSED_HALO, wrapmask 0), at the periodic image where the
   * halo swap is pending too (FUSED on one GPU) */
  const WrapAdj w = wrap_adjust(kp, s, PULL ? wrapmask : 0);
  static_for<0, NVEL>([&](auto P) {
    constexpr int p = P;
    const int off = PULL ? pull_offset<NVEL, p>(kp, w) : 0;
    sum += f2[faddr<2*NVEL, RB>(ns, NVEL + p, i - off)];
  });
  phi[i] = sum;
}

/* PULL = false: in place on f2 (src == f2). PULL = true: propagation(t)
 * fused in: populations pulled from src (the post-collision array with its
 * halo), results written to f2 (the other array), as k_propagate_collide
 * does for one distribution. */
/* LAY as k_propagate_collide (PULL only): 0 SoA -> SoA, 1 SoA -> blocked,
 * 2 blocked -> blocked, + 4 nontemporal stores; the blocked order of two
 * distributions is [site/256][n*NVEL + p][site%256]: a block writes ONE
 * contiguous 2*NVEL*2 KiB chunk instead of 2*NVEL pieces nsite*8 B apart */
template <int NVEL, int SCHEME, bool PULL, bool NZ = false, int LAY = 0>
__global__ __launch_bounds__(BLOCK)
void k_collide_binary(lbmi_kparam_t kp, const double * src, double * f2,
		      lbmi_hydro_dev_t h, Symm q, double rtau2,
		      const double * __restrict__ phi,
		      const double * __restrict__ grad,
		      const double * __restrict__ delsq, int wrapmask,
		      int i0, int i1, unsigned nblk) {

  using M = Model<NVEL>;
  constexpr bool RB = ((LAY & 3) == 2), WB = ((LAY & 3) != 0), NTS = ((LAY & 4) != 0);
  constexpr int ALIGNV = ((LAY & 3) == 0) ? LBMI_ALIGN : LBW;
  static_assert(LAY == 0 || PULL, "the blocked order is a deferred state");
  unsigned lb;
  if (!logical_block(nblk, lb, (unsigned) kp.xcd_group)) return;
  int i = (i0/ALIGNV)*ALIGNV + (int) (lb*BLOCK + threadIdx.x);
  if (i < i0 || i >= i1) return;
  Site s = decode(kp, i);
  if (!s.interior) return;

  const size_t ns = (size_t) kp.nsite;

  double fl[NVEL];
  double gl[NVEL];
  const WrapAdj w = wrap_adjust(kp, s, PULL ? wrapmask : 0);
  static_for<0, NVEL>([&](auto P) {
    constexpr int p = P;
    const int off = PULL ? pull_offset<NVEL, p>(kp, w) : 0;
    fl[p] = src[faddr<2*NVEL, RB>(ns, p, i - off)];
    if constexpr (p > 0) gl[p] = src[faddr<2*NVEL, RB>(ns, NVEL + p, i - off)];
  });

  double frc[3] = {kp.fbody[0], kp.fbody[1], kp.fbody[2]};
  if (h.force) {
    frc[0] += h.force[i];
    frc[1] += h.force[hstride(kp, h) + i];
    frc[2] += h.force[2*hstride(kp, h) + i];
  }

  const double ph = phi[i];
  const double d2 = delsq[i];
  const size_t gs = (h.gstride > 0) ? (size_t) h.gstride : ns;
  const double gr[3] = {grad[i], grad[gs + i], grad[2*gs + i]};
  double pth[3][3];
  symm_stress(q, ph, gr, d2, pth);
  const double sth[6] = {pth[0][0], pth[0][1], pth[0][2],
			 pth[1][1], pth[1][2], pth[2][2]};

  Relax rx = {kp.rtau_shear, kp.rtau_bulk, kp.rtau_even, kp.rtau_odd};
  double rho, u[3];
  if constexpr (NZ) {
    /* lb_collision_fluctuations (collision.c:884-900, 1663-1745): every
     * site draws (no status test in lb_collision_mrt2_site); the variances
     * are those of the global relaxation times (lb_collision_noise_var_set) */
    SiteNoise nz;
    site_noise<SCHEME>(h, i, rx, nz);
    collide_site_impl<NVEL, SCHEME, true, true>(fl, frc, rx, sth, rho, u, &nz);
  }
  else {
    collide_site_impl<NVEL, SCHEME, true>(fl, frc, rx, sth, rho, u);
  }

  static_for<0, NVEL>([&](auto P) {
    if constexpr (NTS) __builtin_nontemporal_store(fl[P], &f2[faddr<2*NVEL, WB>(ns, P, i)]);
    else f2[faddr<2*NVEL, WB>(ns, P, i)] = fl[P];
  });
  if (h.u) {
    h.u[i] = u[0];
    h.u[hstride(kp, h) + i] = u[1];
    h.u[2*hstride(kp, h) + i] = u[2];
  }

  /* order-parameter distribution, collision.c:955-1024 */
  const double mu = q.a*ph + q.b*ph*ph*ph - q.kappa*d2;
  double jphi[3] = {0.0, 0.0, 0.0};
  static_for<1, NVEL>([&](auto P) {
    constexpr int p = P;
    static_for<0, 3>([&](auto A) {
      constexpr int a = A;
      if constexpr (M::c(p,a) ==  1) jphi[a] += gl[p];
      if constexpr (M::c(p,a) == -1) jphi[a] -= gl[p];
    });
  });
  for (int ia = 0; ia < 3; ia++) {
    jphi[ia] = jphi[ia] - rtau2*(jphi[ia] - ph*u[ia]);
  }
  /* sphi_ab = phi u_a u_b + mu delta_ab */
  const double sp[6] = {ph*u[0]*u[0] + mu, ph*u[0]*u[1], ph*u[0]*u[2],
			ph*u[1]*u[1] + mu, ph*u[1]*u[2], ph*u[2]*u[2] + mu};
  const double r3 = 1.0/3.0;
  const double trsp = r3*(sp[0] + sp[3] + sp[5]);

  static_for<0, NVEL>([&](auto P) {
    constexpr int p = P;
    constexpr int cx = M::c(p,0), cy = M::c(p,1), cz = M::c(p,2);
    double jdotc = 0.0;
    if constexpr (cx ==  1) jdotc += jphi[0];
    if constexpr (cx == -1) jdotc -= jphi[0];
    if constexpr (cy ==  1) jdotc += jphi[1];
    if constexpr (cy == -1) jdotc -= jphi[1];
    if constexpr (cz ==  1) jdotc += jphi[2];
    if constexpr (cz == -1) jdotc -= jphi[2];
    /* sphi : (c c - delta/3) */
    double sq = -trsp;
    if constexpr (cx != 0) sq += sp[0];
    if constexpr (cy != 0) sq += sp[3];
    if constexpr (cz != 0) sq += sp[5];
    if constexpr (cx*cy ==  1) sq += 2.0*sp[1];
    if constexpr (cx*cy == -1) sq -= 2.0*sp[1];
    if constexpr (cx*cz ==  1) sq += 2.0*sp[2];
    if constexpr (cx*cz == -1) sq -= 2.0*sp[2];
    if constexpr (cy*cz ==  1) sq += 2.0*sp[4];
    if constexpr (cy*cz == -1) sq -= 2.0*sp[4];
    constexpr double wp = M::w(p);
    double gn = wp*(jdotc*3.0 + sq*4.5);
    if constexpr (p == 0) gn += ph;
    if constexpr (NTS) __builtin_nontemporal_store(gn, &f2[faddr<2*NVEL, WB>(ns, NVEL + p, i)]);
    else f2[faddr<2*NVEL, WB>(ns, NVEL + p, i)] = gn;
  });
}

/* ---- moments ----------------------------------------------------------------
 *
 * Per interior fluid site: rho = sum_p f_p in p order (lb_0th_moment,
 * model.c:819-833); volume, sum rho, sum rho^2, min, max
 * (stats_distribution.c:82-98); momentum sum_p f_p c_p for p >= 1 with
 * Kahan compensation (kahan_add_double, util_sum.c:30-40;
 * distribution_gm_kernel, stats_distribution.c:311-322).
 * Reduction: per-lane accumulators -> wavefront reduction by cross-lane
 * shuffles (64 lanes) -> one LDS slot per wave -> per-block partial in
 * global memory -> a single-block second kernel. No atomics: the result is
 * bitwise reproducible for a given launch geometry. */

struct Kahan { double sum, cs; };

__device__ __forceinline__ void kadd(Kahan & k, double val) {
  double y = val + k.cs;
  double t = k.sum + y;
  k.cs = y - (t - k.sum);
  k.sum = t;
}

__device__ __forceinline__ void kmerge(Kahan & k, const Kahan & o) {
  kadd(k, o.sum);                       /* kahan_add, util_sum.c:142-150 */
  kadd(k, o.cs);
}

struct Partial {
  double vol, srho, srho2, rmin, rmax;
  Kahan g[3];
};

__device__ __forceinline__ double shfl_down_d(double v, int d) {
  return __shfl_down(v, d, 64);
}

__device__ __forceinline__ void partial_merge_shfl(Partial & a, int d) {
  Partial o;
  o.vol = shfl_down_d(a.vol, d);
  o.srho = shfl_down_d(a.srho, d);
  o.srho2 = shfl_down_d(a.srho2, d);
  o.rmin = shfl_down_d(a.rmin, d);
  o.rmax = shfl_down_d(a.rmax, d);
  for (int ia = 0; ia < 3; ia++) {
    o.g[ia].sum = shfl_down_d(a.g[ia].sum, d);
    o.g[ia].cs = shfl_down_d(a.g[ia].cs, d);
  }
  a.vol += o.vol;
  a.srho += o.srho;
  a.srho2 += o.srho2;
  a.rmin = fmin(a.rmin, o.rmin);
  a.rmax = fmax(a.rmax, o.rmax);
  for (int ia = 0; ia < 3; ia++) kmerge(a.g[ia], o.g[ia]);
}

__device__ __forceinline__ void partial_zero(Partial & a) {
  a.vol = 0.0; a.srho = 0.0; a.srho2 = 0.0;
  a.rmin = 1.7976931348623157e308; a.rmax = -1.7976931348623157e308;
  for (int ia = 0; ia < 3; ia++) { a.g[ia].sum = 0.0; a.g[ia].cs = 0.0; }
}

enum {MOM_NBLK = 1024, MOM_NW = 11};   /* doubles per partial record */

__device__ __forceinline__ void partial_store(const Partial & a, double * w) {
  w[0] = a.vol; w[1] = a.srho; w[2] = a.srho2; w[3] = a.rmin; w[4] = a.rmax;
  for (int ia = 0; ia < 3; ia++) {
    w[5 + 2*ia] = a.g[ia].sum; w[6 + 2*ia] = a.g[ia].cs;
  }
}

__device__ __forceinline__ void partial_load(Partial & a, const double * w) {
  a.vol = w[0]; a.srho = w[1]; a.srho2 = w[2]; a.rmin = w[3]; a.rmax = w[4];
  for (int ia = 0; ia < 3; ia++) {
    a.g[ia].sum = w[5 + 2*ia]; a.g[ia].cs = w[6 + 2*ia];
  }
}

/* Block-level reduction of per-lane partials; result valid in thread 0 */

__device__ __forceinline__ void block_reduce(Partial & acc) {
  __shared__ double lds[(BLOCK/64)*MOM_NW];
  for (int d = 32; d >= 1; d >>= 1) partial_merge_shfl(acc, d);
  int wave = threadIdx.x >> 6;
  int lane = threadIdx.x & 63;
  if (lane == 0) partial_store(acc, lds + wave*MOM_NW);
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < BLOCK/64; w++) {
      Partial o;
      partial_load(o, lds + w*MOM_NW);
      acc.vol += o.vol; acc.srho += o.srho; acc.srho2 += o.srho2;
      acc.rmin = fmin(acc.rmin, o.rmin); acc.rmax = fmax(acc.rmax, o.rmax);
      for (int ia = 0; ia < 3; ia++) kmerge(acc.g[ia], o.g[ia]);
    }
  }
}

template <int NVEL>
__global__ __launch_bounds__(BLOCK)
void k_moments_partial(lbmi_kparam_t kp, const double * __restrict__ f,
		       const char * __restrict__ status,
		       double * __restrict__ work) {

  using M = Model<NVEL>;
  const size_t ns = (size_t) kp.nsite;
  const int nh = kp.nhalo;
  const int i0 = nh*kp.strx;
  const int i1 = (nh + kp.nlocal[0])*kp.strx;

  Partial acc;
  partial_zero(acc);

  for (int i = i0 + blockIdx.x*BLOCK + threadIdx.x; i < i1;
       i += gridDim.x*BLOCK) {
    Site s = decode(kp, i);
    if (!s.interior) continue;
    if (status && status[i] != 0) continue;
    double fl[NVEL];
    static_for<0, NVEL>([&](auto P) { fl[P] = f[ns*P + i]; });
    double rho = 0.0;
    static_for<0, NVEL>([&](auto P) { rho += fl[P]; });
    acc.vol += 1.0;
    acc.srho += rho;
    acc.srho2 += rho*rho;
    acc.rmin = fmin(acc.rmin, rho);
    acc.rmax = fmax(acc.rmax, rho);
    static_for<1, NVEL>([&](auto P) {
      constexpr int p = P;
      static_for<0, 3>([&](auto A) {
	constexpr int a = A;
	if constexpr (M::c(p,a) ==  1) kadd(acc.g[a],  fl[p]);
	if constexpr (M::c(p,a) == -1) kadd(acc.g[a], -fl[p]);
      });
    });
  }

  block_reduce(acc);
  if (threadIdx.x == 0) partial_store(acc, work + (size_t) blockIdx.x*MOM_NW);
}

/* k_field_stats_partial: the statistics of a scalar field over interior fluid
 * sites that cahn_stats_reduce collects with three kernels, each a serial
 * loop under a compare-and-swap lock per block (cahn_hilliard_stats.c:
 * 226-480): volume, Kahan-compensated sum, sum of squares, minimum, maximum
 * -- one pass, the reduction tree of the moments (shuffles, LDS, two stages,
 * no atomics). The partial uses the slots of the moments: g[0] carries the
 * compensated sum, srho2 the squares. */

__global__ __launch_bounds__(BLOCK)
void k_field_stats_partial(lbmi_kparam_t kp, const double * __restrict__ field,
			   const char * __restrict__ status,
			   double * __restrict__ work) {
  const int nh = kp.nhalo;
  const int i0 = nh*kp.strx;
  const int i1 = (nh + kp.nlocal[0])*kp.strx;

  Partial acc;
  partial_zero(acc);

  for (int i = i0 + blockIdx.x*BLOCK + threadIdx.x; i < i1;
       i += gridDim.x*BLOCK) {
    Site s = decode(kp, i);
    if (!s.interior) continue;
    if (status && status[i] != 0) continue;
    const double v = field[i];
    acc.vol += 1.0;
    acc.srho2 += v*v;
    acc.rmin = fmin(acc.rmin, v);
    acc.rmax = fmax(acc.rmax, v);
    kadd(acc.g[0], v);
  }

  block_reduce(acc);
  if (threadIdx.x == 0) partial_store(acc, work + (size_t) blockIdx.x*MOM_NW);
}

__global__ __launch_bounds__(BLOCK)
void k_moments_final(int npartial, const double * __restrict__ work,
		     double * __restrict__ out) {
  Partial acc;
  partial_zero(acc);
  for (int b = threadIdx.x; b < npartial; b += BLOCK) {
    Partial o;
    partial_load(o, work + (size_t) b*MOM_NW);
    acc.vol += o.vol; acc.srho += o.srho; acc.srho2 += o.srho2;
    acc.rmin = fmin(acc.rmin, o.rmin); acc.rmax = fmax(acc.rmax, o.rmax);
    for (int ia = 0; ia < 3; ia++) kmerge(acc.g[ia], o.g[ia]);
  }
  block_reduce(acc);
  if (threadIdx.x == 0) {
    out[0] = acc.vol; out[1] = acc.srho; out[2] = acc.srho2;
    out[3] = acc.rmin; out[4] = acc.rmax;
    out[5] = acc.g[0].sum + acc.g[0].cs;    /* kahan_sum, util_sum.c:61 */
    out[6] = acc.g[1].sum + acc.g[1].cs;
    out[7] = acc.g[2].sum + acc.g[2].cs;
    out[8] = 0.0;
  }
}

/* k_interior_copy: dst <- src at the interior sites of an SoA field of ncomp
 * components (the halo of dst stays as it is) */

__global__ __launch_bounds__(BLOCK)
void k_interior_copy(lbmi_kparam_t kp, int ncomp, const double * __restrict__ src,
		     double * __restrict__ dst) {
  const int i0 = kp.nhalo*kp.strx;
  const int i1 = (kp.nhalo + kp.nlocal[0])*kp.strx;
  int i = i0 + blockIdx.x*BLOCK + threadIdx.x;
  if (i >= i1) return;
  Site s = decode(kp, i);
  if (!s.interior) return;
  for (int n = 0; n < ncomp; n++) {
    dst[(size_t) kp.nsite*n + i] = src[(size_t) kp.nsite*n + i];
  }
}

/* k_density: lb_0th_moment (model.c:817-832) of every interior site, summed in
 * p order as there, into a dense array in (ic, jc, kc) order: what
 * stats_distribution_print walks over (stats_distribution.c:73-88). The host
 * adds the values up in the reference's own order, so the printed sums agree
 * with its CPU path to the last digit. */

template <int NVEL>
__global__ __launch_bounds__(BLOCK)
void k_density(lbmi_kparam_t kp, const double * __restrict__ f,
	       double * __restrict__ rho, long long ninterior) {
  long long ib = (long long) blockIdx.x*BLOCK + threadIdx.x;
  if (ib >= ninterior) return;
  const size_t i = interior_site(kp, ib);
  const size_t ns = (size_t) kp.nsite;
  double r = 0.0;
  static_for<0, NVEL>([&](auto P) { r += f[ns*P + i]; });
  rho[ib] = r;
}

/* ---- launch helpers --------------------------------------------------------- */

inline unsigned grid_for(unsigned nblk, unsigned group = 0) {
  unsigned q = 8u*(group > 0 ? group : 1u);
  return ((nblk + q - 1u)/q)*q;
}

template <int NVEL>
int launch_collide(const lbmi_kparam_t & kp, double * f,
		   const lbmi_hydro_dev_t & h, hipStream_t st) {
  int i0 = kp.nhalo*kp.strx;
  int i1 = (kp.nhalo + kp.nlocal[0])*kp.strx;
  int i0a = (i0/LBMI_ALIGN)*LBMI_ALIGN;
  unsigned nblk = (unsigned) ((i1 - i0a + BLOCK - 1)/BLOCK);
  dim3 grid(grid_for(nblk, (unsigned) kp.xcd_group)), block(BLOCK);
  unsigned lds = (kp.lds_cap <= 65536 && nblk > 4096u) ? (unsigned) kp.lds_cap : 0u;
  if (h.noise != nullptr) {
    if constexpr (NVEL == 19) {
      switch (kp.scheme) {
      case LBMI_M10:
	hipLaunchKernelGGL((k_collide<19, LBMI_M10, true>), grid, block, 0, st, kp, f, h, i0, i1, nblk);
	break;
      case LBMI_BGK:
	hipLaunchKernelGGL((k_collide<19, LBMI_BGK, true>), grid, block, 0, st, kp, f, h, i0, i1, nblk);
	break;
      case LBMI_TRT:
	hipLaunchKernelGGL((k_collide<19, LBMI_TRT, true>), grid, block, 0, st, kp, f, h, i0, i1, nblk);
	break;
      default:
	return (int) hipErrorInvalidValue;
      }
      return (int) hipGetLastError();
    }
    return (int) hipErrorInvalidValue;
  }
  switch (kp.scheme) {
  case LBMI_M10:
    hipLaunchKernelGGL((k_collide<NVEL, LBMI_M10>), grid, block, lds, st,
		       kp, f, h, i0, i1, nblk);
    break;
  case LBMI_BGK:
    hipLaunchKernelGGL((k_collide<NVEL, LBMI_BGK>), grid, block, lds, st,
		       kp, f, h, i0, i1, nblk);
    break;
  case LBMI_TRT:
    if constexpr (NVEL == 19) {
      hipLaunchKernelGGL((k_collide<NVEL, LBMI_TRT>), grid, block, lds, st,
			 kp, f, h, i0, i1, nblk);
      break;
    }
    return (int) hipErrorInvalidValue;
  default:
    return (int) hipErrorInvalidValue;
  }
  return (int) hipGetLastError();
}

template <int NVEL>
int launch_collide_to(const lbmi_kparam_t & kp, const double * f, double * fo,
		      const lbmi_hydro_dev_t & h, hipStream_t st) {
  unsigned nblk = (unsigned) ((kp.nsite + BLOCK - 1)/BLOCK);
  dim3 grid(grid_for(nblk, (unsigned) kp.xcd_group)), block(BLOCK);
  unsigned lds = (kp.lds_cap <= 65536 && nblk > 4096u) ? (unsigned) kp.lds_cap : 0u;
  switch (kp.scheme) {
  case LBMI_M10:
    hipLaunchKernelGGL((k_collide_to<NVEL, LBMI_M10>), grid, block, lds, st, kp, f, fo, h, nblk);
    break;
  case LBMI_BGK:
    hipLaunchKernelGGL((k_collide_to<NVEL, LBMI_BGK>), grid, block, lds, st, kp, f, fo, h, nblk);
    break;
  case LBMI_TRT:
    if constexpr (NVEL == 19) {
      hipLaunchKernelGGL((k_collide_to<NVEL, LBMI_TRT>), grid, block, lds, st, kp, f, fo, h, nblk);
      break;
    }
    return (int) hipErrorInvalidValue;
  default:
    return (int) hipErrorInvalidValue;
  }
  return (int) hipGetLastError();
}

template <int NVEL, bool WRAP, int LAY, bool HIO, bool XB>
int launch_pc_hio(const lbmi_kparam_t & kp, const double * f, double * fp,
		  const lbmi_hydro_dev_t & h, int wrapmask, int i0, int i1,
		  int j0, int j1, hipStream_t st, const lbmi_xbuf_t & xb) {
  constexpr int ALIGNV = ((LAY & 3) == 0) ? LBMI_ALIGN : LBW;
  static_assert((LAY & 3) == 0 || BLOCK*SPT == LBW, "blocked order: one thread block per layout block");
  const int i0a = (i0/ALIGNV)*ALIGNV;
  const int j0a = (j0/ALIGNV)*ALIGNV;
  unsigned nblk_first = (unsigned) ((i1 - i0a + BLOCK*SPT - 1)/(BLOCK*SPT));
  unsigned nblk = nblk_first;
  if (j1 > j0) nblk += (unsigned) ((j1 - j0a + BLOCK*SPT - 1)/(BLOCK*SPT));
  dim3 grid(grid_for(nblk, (unsigned) kp.xcd_group)), block(BLOCK);
  /* dynamic LDS is not used by the kernel: it only caps the number of
   * resident blocks per CU (160 KiB / lds_cap), see DESIGN.md */
  unsigned lds = (unsigned) kp.lds_cap;
  /* a launch that fits on the chip in two rounds anyway (the boundary
   * planes of a slab, a 64^3 lattice) is latency-bound and L2/MALL
   * resident: give it the full occupancy */
  if (nblk <= 4096u) lds = 0u;
  if (lds > 65536u) {
    /* above 64 KiB the limit must be raised per kernel */
    const void * fn = nullptr;
    if (kp.scheme == LBMI_M10) fn = (const void *) k_propagate_collide<NVEL, LBMI_M10, WRAP, LAY, HIO, XB>;
    if (kp.scheme == LBMI_BGK) fn = (const void *) k_propagate_collide<NVEL, LBMI_BGK, WRAP, LAY, HIO, XB>;
    if constexpr (NVEL == 19) {
      if (kp.scheme == LBMI_TRT) fn = (const void *) k_propagate_collide<NVEL, LBMI_TRT, WRAP, LAY, HIO, XB>;
    }
    if (fn) {
      hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
      if (e != hipSuccess) return (int) e;
    }
  }
  switch (kp.scheme) {
  case LBMI_M10:
    hipLaunchKernelGGL((k_propagate_collide<NVEL, LBMI_M10, WRAP, LAY, HIO, XB>), grid,
		       block, lds, st, kp, f, fp, h, wrapmask, i0, i1, nblk,
		       j0, j1, nblk_first, xb);
    break;
  case LBMI_BGK:
    hipLaunchKernelGGL((k_propagate_collide<NVEL, LBMI_BGK, WRAP, LAY, HIO, XB>), grid,
		       block, lds, st, kp, f, fp, h, wrapmask, i0, i1, nblk,
		       j0, j1, nblk_first, xb);
    break;
  case LBMI_TRT:
    if constexpr (NVEL == 19) {
      hipLaunchKernelGGL((k_propagate_collide<NVEL, LBMI_TRT, WRAP, LAY, HIO, XB>), grid,
			 block, lds, st, kp, f, fp, h, wrapmask, i0, i1, nblk,
			 j0, j1, nblk_first, xb);
      break;
    }
    return (int) hipErrorInvalidValue;
  default:
    return (int) hipErrorInvalidValue;
  }
  return (int) hipGetLastError();
}

template <int NVEL, bool WRAP, int LAY>
int launch_pc(const lbmi_kparam_t & kp, const double * f, double * fp,
	      const lbmi_hydro_dev_t & h, int wrapmask, int i0, int i1,
	      int j0, int j1, hipStream_t st, const lbmi_xbuf_t * xb) {
  const bool hio = (h.force != nullptr || h.rho != nullptr || h.u != nullptr);
  const lbmi_xbuf_t none = {nullptr, nullptr, nullptr, nullptr};
  if constexpr (WRAP && (LAY & 4) == 0) {
    /* the boundary planes of a slab against the exchange buffers */
    if (xb != nullptr) {
      if (hio) return launch_pc_hio<NVEL, WRAP, LAY, true, true>(kp, f, fp, h, wrapmask, i0, i1, j0, j1, st, *xb);
      return launch_pc_hio<NVEL, WRAP, LAY, false, true>(kp, f, fp, h, wrapmask, i0, i1, j0, j1, st, *xb);
    }
  }
  if (xb != nullptr) return (int) hipErrorInvalidValue;
  if (hio) return launch_pc_hio<NVEL, WRAP, LAY, true, false>(kp, f, fp, h, wrapmask, i0, i1, j0, j1, st, none);
  return launch_pc_hio<NVEL, WRAP, LAY, false, false>(kp, f, fp, h, wrapmask, i0, i1, j0, j1, st, none);
}

/* with fluctuations: D3Q19, the hydro arrays as they are (the variant with
 * hydro traffic tests every pointer), no exchange buffers */
template <bool WRAP, int LAY>
int launch_pc_noise(const lbmi_kparam_t & kp, const double * f, double * fp,
		    const lbmi_hydro_dev_t & h, int wrapmask, int i0, int i1,
		    int j0, int j1, hipStream_t st) {
  const lbmi_xbuf_t none = {nullptr, nullptr, nullptr, nullptr};
  constexpr int ALIGNV = ((LAY & 3) == 0) ? LBMI_ALIGN : LBW;
  const int i0a = (i0/ALIGNV)*ALIGNV;
  const int j0a = (j0/ALIGNV)*ALIGNV;
  unsigned nblk_first = (unsigned) ((i1 - i0a + BLOCK*SPT - 1)/(BLOCK*SPT));
  unsigned nblk = nblk_first;
  if (j1 > j0) nblk += (unsigned) ((j1 - j0a + BLOCK*SPT - 1)/(BLOCK*SPT));
  dim3 grid(grid_for(nblk, (unsigned) kp.xcd_group)), block(BLOCK);
  switch (kp.scheme) {
  case LBMI_M10:
    hipLaunchKernelGGL((k_propagate_collide<19, LBMI_M10, WRAP, LAY, true, false, true>), grid,
		       block, 0, st, kp, f, fp, h, wrapmask, i0, i1, nblk, j0, j1, nblk_first, none);
    break;
  case LBMI_BGK:
    hipLaunchKernelGGL((k_propagate_collide<19, LBMI_BGK, WRAP, LAY, true, false, true>), grid,
		       block, 0, st, kp, f, fp, h, wrapmask, i0, i1, nblk, j0, j1, nblk_first, none);
    break;
  case LBMI_TRT:
    hipLaunchKernelGGL((k_propagate_collide<19, LBMI_TRT, WRAP, LAY, true, false, true>), grid,
		       block, 0, st, kp, f, fp, h, wrapmask, i0, i1, nblk, j0, j1, nblk_first, none);
    break;
  default:
    return (int) hipErrorInvalidValue;
  }
  return (int) hipGetLastError();
}

template <int NVEL>
int launch_pc_any(const lbmi_kparam_t & kp, const double * f, double * fp,
		  const lbmi_hydro_dev_t & h, int wrapmask, int lay, int i0,
		  int i1, int j0, int j1, hipStream_t st,
		  const lbmi_xbuf_t * xb) {
  if (h.noise != nullptr) {
    if (NVEL != 19 || xb != nullptr) return (int) hipErrorInvalidValue;
    if (!wrapmask) {
      if (lay != 0) return (int) hipErrorInvalidValue;
      return launch_pc_noise<false, 0>(kp, f, fp, h, 0, i0, i1, j0, j1, st);
    }
    if (lay == 0) return launch_pc_noise<true, 0>(kp, f, fp, h, wrapmask, i0, i1, j0, j1, st);
#if LBMI_BLOCK*LBMI_SPT == 256
    if (lay == 1) return launch_pc_noise<true, 1>(kp, f, fp, h, wrapmask, i0, i1, j0, j1, st);
    if (lay == 2) return launch_pc_noise<true, 2>(kp, f, fp, h, wrapmask, i0, i1, j0, j1, st);
#endif
    return (int) hipErrorInvalidValue;
  }
  if (!wrapmask) {
    if (lay != 0 || xb != nullptr) return (int) hipErrorInvalidValue;
    return launch_pc<NVEL, false, 0>(kp, f, fp, h, 0, i0, i1, j0, j1, st, nullptr);
  }
  if (lay == 0) return launch_pc<NVEL, true, 0>(kp, f, fp, h, wrapmask, i0, i1, j0, j1, st, xb);
#if LBMI_BLOCK*LBMI_SPT == 256
  if ((kp.nt_store & 1) && xb == nullptr) {
    if (lay == 1) return launch_pc<NVEL, true, 5>(kp, f, fp, h, wrapmask, i0, i1, j0, j1, st, nullptr);
    if (lay == 2) return launch_pc<NVEL, true, 6>(kp, f, fp, h, wrapmask, i0, i1, j0, j1, st, nullptr);
  }
  if (lay == 1) return launch_pc<NVEL, true, 1>(kp, f, fp, h, wrapmask, i0, i1, j0, j1, st, xb);
  if (lay == 2) return launch_pc<NVEL, true, 2>(kp, f, fp, h, wrapmask, i0, i1, j0, j1, st, xb);
#endif
  return (int) hipErrorInvalidValue;
}

struct Range1D {
  int i0, i1;
  unsigned nblk, grid, lds;
};

inline Range1D interior_range(const lbmi_kparam_t & kp) {
  Range1D r;
  r.i0 = kp.nhalo*kp.strx;
  r.i1 = (kp.nhalo + kp.nlocal[0])*kp.strx;
  int i0a = (r.i0/LBMI_ALIGN)*LBMI_ALIGN;
  r.nblk = (unsigned) ((r.i1 - i0a + BLOCK - 1)/BLOCK);
  r.grid = grid_for(r.nblk, (unsigned) kp.xcd_group);
  r.lds = (kp.lds_cap <= 65536 && r.nblk > 4096u) ? (unsigned) kp.lds_cap : 0u;
  return r;
}

template <int NVEL>
int launch_aa_even(const lbmi_kparam_t & kp, double * f,
		   const lbmi_hydro_dev_t & h, hipStream_t st) {
  Range1D r = interior_range(kp);
  dim3 grid(r.grid), block(BLOCK);
  switch (kp.scheme) {
  case LBMI_M10:
    hipLaunchKernelGGL((k_aa_even<NVEL, LBMI_M10>), grid, block, r.lds, st,
		       kp, f, h, r.i0, r.i1, r.nblk);
    break;
  case LBMI_BGK:
    hipLaunchKernelGGL((k_aa_even<NVEL, LBMI_BGK>), grid, block, r.lds, st,
		       kp, f, h, r.i0, r.i1, r.nblk);
    break;
  case LBMI_TRT:
    if constexpr (NVEL == 19) {
      hipLaunchKernelGGL((k_aa_even<NVEL, LBMI_TRT>), grid, block, r.lds, st,
			 kp, f, h, r.i0, r.i1, r.nblk);
      break;
    }
    return (int) hipErrorInvalidValue;
  default:
    return (int) hipErrorInvalidValue;
  }
  return (int) hipGetLastError();
}

template <int NVEL, bool SWAPPED_IN>
int launch_aa_odd(const lbmi_kparam_t & kp, double * f,
		  const lbmi_hydro_dev_t & h, int wrapmask, hipStream_t st) {
  Range1D r = interior_range(kp);
  dim3 grid(r.grid), block(BLOCK);
  switch (kp.scheme) {
  case LBMI_M10:
    hipLaunchKernelGGL((k_aa_odd<NVEL, LBMI_M10, SWAPPED_IN>), grid, block,
		       r.lds, st, kp, f, h, wrapmask, r.i0, r.i1, r.nblk);
    break;
  case LBMI_BGK:
    hipLaunchKernelGGL((k_aa_odd<NVEL, LBMI_BGK, SWAPPED_IN>), grid, block,
		       r.lds, st, kp, f, h, wrapmask, r.i0, r.i1, r.nblk);
    break;
  case LBMI_TRT:
    if constexpr (NVEL == 19) {
      hipLaunchKernelGGL((k_aa_odd<NVEL, LBMI_TRT, SWAPPED_IN>), grid, block,
			 r.lds, st, kp, f, h, wrapmask, r.i0, r.i1, r.nblk);
      break;
    }
    return (int) hipErrorInvalidValue;
  default:
    return (int) hipErrorInvalidValue;
  }
  return (int) hipGetLastError();
}

template <int NVEL>
int model_tables(int8_t * cv, double * wv, double * na, double * ma) {
  using M = Model<NVEL>;
  for (int p = 0; p < NVEL; p++) {
    for (int a = 0; a < 3; a++) cv[3*p + a] = (int8_t) M::c(p, a);
    wv[p] = M::w(p);
  }
  for (int m = 0; m < NVEL; m++) {
    na[m] = M::na(m);
    for (int p = 0; p < NVEL; p++) ma[m*NVEL + p] = M::ma(m, p);
  }
  return 0;
}

} /* namespace */

/* ---- C launchers ------------------------------------------------------------ */

/* ---- bounce-back on links (row f4) --------------------------------------------
 *
 * wall_bbl_kernel (wall.c:996-1107) without colloids: link n joins the fluid
 * site i to the solid site j = i + c_p; the post-collision population p at i
 * comes back as population nvel - p at j (from where lb_propagation pulls it
 * into i), minus 2 rcs2 w_p rho0 c_p.u_w for a moving wall; both
 * distributions when ndist = 2. The momentum given to the wall,
 * (2 f - 2 rcs2 w_p rho0 c_p.u_w - 2 w_p) c_p, is reduced per block
 * (shuffles + LDS) and added to fnet in block order. */

enum {WALL_BLOCK = 256, WALL_NBLK_MAX = 1024};
enum {MAP_COLLOID = 2};                                    /* map.h:23 */

/* block reduction: 64-lane shuffles, then one LDS slot per wave */
__device__ __forceinline__ void wall_block_sum(double fx, double fy, double fz,
					       double * __restrict__ part) {
  for (int d = 32; d > 0; d >>= 1) {
    fx += shfl_down_d(fx, d);
    fy += shfl_down_d(fy, d);
    fz += shfl_down_d(fz, d);
  }
  __shared__ double red[WALL_BLOCK/64][3];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) { red[wave][0] = fx; red[wave][1] = fy; red[wave][2] = fz; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int w = 0; w < WALL_BLOCK/64; w++) {
      s0 += red[w][0]; s1 += red[w][1]; s2 += red[w][2];
    }
    part[3*blockIdx.x + 0] = s0;
    part[3*blockIdx.x + 1] = s1;
    part[3*blockIdx.x + 2] = s2;
  }
}

/* A link record is usable if both sites lie inside the arrays and p is a
 * moving direction (wall.c:440-451 stores p = 1 .. nvel - 1) */
__device__ __forceinline__ bool wall_link_ok(size_t ns, int nvel, int i, int j,
					     int p) {
  return (size_t) (unsigned) i < ns && (size_t) (unsigned) j < ns &&
    p >= 1 && p < nvel;
}

__global__ __launch_bounds__(WALL_BLOCK)
void k_wall_bbl(lbmi_kparam_t kp, lbmi_wall_tab_t tab, double * __restrict__ f,
		int nlink, const int * __restrict__ linki,
		const int * __restrict__ linkj, const int * __restrict__ linkp,
		const int * __restrict__ linku,
		const char * __restrict__ status, double * __restrict__ part,
		int * __restrict__ err) {

  const size_t ns = (size_t) kp.nsite;
  const double rcs2 = 3.0;
  double fx = 0.0, fy = 0.0, fz = 0.0;

  for (int n = blockIdx.x*WALL_BLOCK + threadIdx.x; n < nlink;
       n += gridDim.x*WALL_BLOCK) {
    const int i = linki[n], j = linkj[n];
    const int ij = linkp[n], ji = tab.nvel - ij, ia = linku[n];
    /* a record that would address outside f (link arrays the caller owns may
     * hold anything) is never dereferenced: sticky error, read by the host */
    if (!wall_link_ok(ns, tab.nvel, i, j, ij) || (unsigned) ia > 2u) {
      *err = n + 1;
      continue;
    }
    const double cx = tab.cv[ij][0], cy = tab.cv[ij][1], cz = tab.cv[ij][2];
    const double cdotu = cx*tab.uw[ia][0] + cy*tab.uw[ia][1] + cz*tab.uw[ia][2];
    const double wall = 2.0*rcs2*tab.wv[ij]*tab.rho0*cdotu;
    double fp = f[ns*ij + i];
    if (status != nullptr && status[i] == MAP_COLLOID) {
      /* a colloid sits on the fluid side: its own bounce-back moves the
       * populations; here only the accounting (wall.c:1048-1061) */
      fp += f[ns*ji + j];
      fx += (fp - 2.0*tab.wv[ij])*cx;
      fy += (fp - 2.0*tab.wv[ij])*cy;
      fz += (fp - 2.0*tab.wv[ij])*cz;
      continue;
    }
    const double force = 2.0*fp - wall;
    fx += (force - 2.0*tab.wv[ij])*cx;
    fy += (force - 2.0*tab.wv[ij])*cy;
    fz += (force - 2.0*tab.wv[ij])*cz;
    f[ns*ji + j] = fp - wall;
    if (tab.ndist > 1) {
      const size_t off = ns*(size_t) tab.nvel;
      f[off + ns*ji + j] = f[off + ns*ij + i] - wall;
    }
  }

  wall_block_sum(fx, fy, fz, part);
}

/* wall_bbl_slip_kernel (wall.c:1118-1205) without colloids: a fraction s of
 * what comes back along link n is the population q of the fluid site k one
 * step along the wall from i (specular reflection), 1 - s is bounced back;
 * walls at rest, LB_RHO only, as there. The normal factor w = -(c_p + c_q)/2
 * is formed in integers, as there. */

__global__ __launch_bounds__(WALL_BLOCK)
void k_wall_bbl_slip(lbmi_kparam_t kp, lbmi_wall_tab_t tab,
		     double * __restrict__ f, int nlink,
		     const int * __restrict__ linki,
		     const int * __restrict__ linkj,
		     const int * __restrict__ linkp,
		     const int * __restrict__ linkk,
		     const int8_t * __restrict__ linkq,
		     const int8_t * __restrict__ links,
		     const char * __restrict__ status,
		     double * __restrict__ part, int * __restrict__ err) {

  const size_t ns = (size_t) kp.nsite;
  double fsum[3] = {0.0, 0.0, 0.0};

  for (int n = blockIdx.x*WALL_BLOCK + threadIdx.x; n < nlink;
       n += gridDim.x*WALL_BLOCK) {
    const int i = linki[n], j = linkj[n], k = linkk[n];
    const int ij = linkp[n], ji = tab.nvel - ij, q = linkq[n];
    const int is = links[n];
    if (!wall_link_ok(ns, tab.nvel, i, j, ij) || (size_t) (unsigned) k >= ns ||
	(unsigned) q >= (unsigned) tab.nvel || (unsigned) is >= 19u) {
      *err = n + 1;
      continue;
    }
    const double s = tab.slip[is];
    const double fi = f[ns*ij + i];
    const double fk = f[ns*q + k];
    if (status != nullptr && status[i] == MAP_COLLOID) {     /* wall.c:1148-1161 */
      const double fp = fi + f[ns*ji + j];
      for (int a = 0; a < 3; a++) fsum[a] += (fp - 2.0*tab.wv[ij])*tab.cv[ij][a];
      continue;
    }
    f[ns*ji + j] = (1.0 - s)*fi + s*fk;
    for (int a = 0; a < 3; a++) {
      const int iw = -((int) tab.cv[ij][a] + (int) tab.cv[q][a])/2;
      const double w = iw;
      fsum[a] += 2.0*(1.0 - s)*(fi - tab.wv[ij])*tab.cv[ij][a];
      fsum[a] += 2.0*w*w*s*(fk - tab.wv[q])*tab.cv[q][a];
    }
  }
  wall_block_sum(fsum[0], fsum[1], fsum[2], part);
}

/* One wavefront: lane l adds the partials of blocks l, l + 64, ... in that
 * order, then a shuffle tree: a fixed order for a given number of blocks (no
 * atomics), and 16 dependent steps instead of 1024 for the largest grid. */
__global__ __launch_bounds__(64)
void k_wall_fnet(int nblk, const double * __restrict__ part,
		 double * __restrict__ fnet) {
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  for (int b = threadIdx.x; b < nblk; b += 64) {
    s0 += part[3*b]; s1 += part[3*b + 1]; s2 += part[3*b + 2];
  }
  for (int d = 32; d > 0; d >>= 1) {
    s0 += shfl_down_d(s0, d);
    s1 += shfl_down_d(s1, d);
    s2 += shfl_down_d(s2, d);
  }
  if (threadIdx.x == 0) {
    fnet[0] += s0; fnet[1] += s1; fnet[2] += s2;
  }
}

extern "C" int lbmi_k_collide_to(const lbmi_kparam_t * kp, const double * f,
				 double * fo, const lbmi_hydro_dev_t * h,
				 void * stream) {
  hipStream_t st = (hipStream_t) stream;
  if (h->noise != nullptr || f == fo) return (int) hipErrorInvalidValue;
  if (kp->nvel == 19) return launch_collide_to<19>(*kp, f, fo, *h, st);
  if (kp->nvel == 27) return launch_collide_to<27>(*kp, f, fo, *h, st);
  return (int) hipErrorInvalidValue;
}

extern "C" int lbmi_k_collide(const lbmi_kparam_t * kp, double * f,
			      const lbmi_hydro_dev_t * h, void * stream) {
  hipStream_t st = (hipStream_t) stream;
  if (kp->nvel == 19) return launch_collide<19>(*kp, f, *h, st);
  if (kp->nvel == 27) return launch_collide<27>(*kp, f, *h, st);
  return (int) hipErrorInvalidValue;
}

template <int NVEL>
static int launch_collide_fe(const lbmi_kparam_t & kp, double * f,
			     const lbmi_hydro_dev_t & h, double a, double b,
			     double kappa, const double * phi,
			     const double * grad, const double * delsq,
			     hipStream_t st) {
  Range1D r = interior_range(kp);
  dim3 grid(r.grid), block(BLOCK);
  switch (kp.scheme) {
  case LBMI_M10:
    hipLaunchKernelGGL((k_collide_fe<NVEL, LBMI_M10>), grid, block, 0, st, kp,
		       f, h, a, b, kappa, phi, grad, delsq, r.i0, r.i1, r.nblk);
    break;
  case LBMI_BGK:
    hipLaunchKernelGGL((k_collide_fe<NVEL, LBMI_BGK>), grid, block, 0, st, kp,
		       f, h, a, b, kappa, phi, grad, delsq, r.i0, r.i1, r.nblk);
    break;
  case LBMI_TRT:
    if constexpr (NVEL == 19) {
      hipLaunchKernelGGL((k_collide_fe<NVEL, LBMI_TRT>), grid, block, 0, st,
			 kp, f, h, a, b, kappa, phi, grad, delsq, r.i0, r.i1,
			 r.nblk);
      break;
    }
    return (int) hipErrorInvalidValue;
  default:
    return (int) hipErrorInvalidValue;
  }
  return (int) hipGetLastError();
}

extern "C" int lbmi_k_collide_fe(const lbmi_kparam_t * kp, double * f,
				 const lbmi_hydro_dev_t * h, double a,
				 double b, double kappa, const double * phi,
				 const double * grad, const double * delsq,
				 void * stream) {
  hipStream_t st = (hipStream_t) stream;
  if (kp->nvel == 19) return launch_collide_fe<19>(*kp, f, *h, a, b, kappa, phi, grad, delsq, st);
  if (kp->nvel == 27) return launch_collide_fe<27>(*kp, f, *h, a, b, kappa, phi, grad, delsq, st);
  return (int) hipErrorInvalidValue;
}

template <int NVEL, bool WRAP>
static int launch_pc_fe(const lbmi_kparam_t & kp, const double * f, double * fp,
			const lbmi_hydro_dev_t & h, double a, double b, double kappa,
			const double * phi, const double * grad, const double * delsq,
			int wrapmask, hipStream_t st) {
  Range1D r = interior_range(kp);
  dim3 grid(r.grid), block(BLOCK);
  switch (kp.scheme) {
  case LBMI_M10:
    hipLaunchKernelGGL((k_propagate_collide_fe<NVEL, LBMI_M10, WRAP>), grid, block, r.lds, st,
		       kp, f, fp, h, a, b, kappa, phi, grad, delsq, wrapmask, r.i0, r.i1, r.nblk);
    break;
  case LBMI_BGK:
    hipLaunchKernelGGL((k_propagate_collide_fe<NVEL, LBMI_BGK, WRAP>), grid, block, r.lds, st,
		       kp, f, fp, h, a, b, kappa, phi, grad, delsq, wrapmask, r.i0, r.i1, r.nblk);
    break;
  case LBMI_TRT:
    if constexpr (NVEL == 19) {
      hipLaunchKernelGGL((k_propagate_collide_fe<NVEL, LBMI_TRT, WRAP>), grid, block, r.lds, st,
			 kp, f, fp, h, a, b, kappa, phi, grad, delsq, wrapmask, r.i0, r.i1, r.nblk);
      break;
    }
    return (int) hipErrorInvalidValue;
  default:
    return (int) hipErrorInvalidValue;
  }
  return (int) hipGetLastError();
}

extern "C" int lbmi_k_propagate_collide_fe(const lbmi_kparam_t * kp, const double * f,
					   double * fp, const lbmi_hydro_dev_t * h,
					   double a, double b, double kappa,
					   const double * phi, const double * grad,
					   const double * delsq, int wrapmask,
					   void * stream) {
  hipStream_t st = (hipStream_t) stream;
  if (f == fp) return (int) hipErrorInvalidValue;
  if (kp->nvel == 19) {
    return wrapmask ? launch_pc_fe<19, true>(*kp, f, fp, *h, a, b, kappa, phi, grad, delsq, wrapmask, st)
      : launch_pc_fe<19, false>(*kp, f, fp, *h, a, b, kappa, phi, grad, delsq, 0, st);
  }
  if (kp->nvel == 27) {
    return wrapmask ? launch_pc_fe<27, true>(*kp, f, fp, *h, a, b, kappa, phi, grad, delsq, wrapmask, st)
      : launch_pc_fe<27, false>(*kp, f, fp, *h, a, b, kappa, phi, grad, delsq, 0, st);
  }
  return (int) hipErrorInvalidValue;
}

extern "C" int lbmi_k_propagate(const lbmi_kparam_t * kp, const double * f,
				double * fprime, void * stream) {
  hipStream_t st = (hipStream_t) stream;
  int i0 = kp->nhalo*kp->strx;
  int i1 = (kp->nhalo + kp->nlocal[0])*kp->strx;
  int i0a = (i0/LBMI_ALIGN)*LBMI_ALIGN;
  unsigned nblk = (unsigned) ((i1 - i0a + BLOCK - 1)/BLOCK);
  dim3 grid(grid_for(nblk, (unsigned) kp->xcd_group)), block(BLOCK);
  unsigned lds = (kp->lds_cap <= 65536 && nblk > 4096u) ? (unsigned) kp->lds_cap : 0u;
  if (kp->nvel == 19) {
    hipLaunchKernelGGL((k_propagate<19>), grid, block, lds, st, *kp, f, fprime,
		       i0, i1, nblk);
  }
  else if (kp->nvel == 27) {
    hipLaunchKernelGGL((k_propagate<27>), grid, block, lds, st, *kp, f, fprime,
		       i0, i1, nblk);
  }
  else {
    return (int) hipErrorInvalidValue;
  }
  return (int) hipGetLastError();
}

extern "C" int lbmi_k_propagate_collide(const lbmi_kparam_t * kp,
					const double * f, double * fprime,
					const lbmi_hydro_dev_t * h,
					int wrapmask, int lay, int xlo,
					int xhi, int xlo2, int xhi2,
					const lbmi_xbuf_t * xb,
					void * stream) {
  hipStream_t st = (hipStream_t) stream;
  if (xhi < xlo) {
    if (xhi2 < xlo2) return 0;
    xlo = xlo2; xhi = xhi2; xlo2 = 0; xhi2 = -1;
  }
  int i0 = xlo*kp->strx;
  int i1 = (xhi + 1)*kp->strx;
  int j0 = 0, j1 = 0;
  if (xhi2 >= xlo2) {
    j0 = xlo2*kp->strx;
    j1 = (xhi2 + 1)*kp->strx;
  }
  if (kp->nvel == 19) {
    return launch_pc_any<19>(*kp, f, fprime, *h, wrapmask, lay, i0, i1, j0, j1, st, xb);
  }
  if (kp->nvel == 27) {
    return launch_pc_any<27>(*kp, f, fprime, *h, wrapmask, lay, i0, i1, j0, j1, st, xb);
  }
  return (int) hipErrorInvalidValue;
}

template <int NVEL, bool HIO>
static int launch_pc_halo(const lbmi_kparam_t & kp, const double * f, double * fp,
			  const lbmi_hydro_dev_t & h, hipStream_t st) {
  /* the x planes of the interior and the halo plane on either side */
  const int i0 = (kp.nhalo - 1)*kp.strx;
  const int i1 = (kp.nhalo + kp.nlocal[0] + 1)*kp.strx;
  const int i0a = (i0/LBMI_ALIGN)*LBMI_ALIGN;
  const unsigned nblk = (unsigned) ((i1 - i0a + BLOCK - 1)/BLOCK);
  dim3 grid(grid_for(nblk, (unsigned) kp.xcd_group)), block(BLOCK);
  unsigned lds = (kp.lds_cap <= 65536 && nblk > 4096u) ? (unsigned) kp.lds_cap : 0u;
  switch (kp.scheme) {
  case LBMI_M10:
    hipLaunchKernelGGL((k_propagate_collide_halo<NVEL, LBMI_M10, HIO>), grid, block,
		       lds, st, kp, f, fp, h, i0, i1, nblk);
    break;
  case LBMI_BGK:
    hipLaunchKernelGGL((k_propagate_collide_halo<NVEL, LBMI_BGK, HIO>), grid, block,
		       lds, st, kp, f, fp, h, i0, i1, nblk);
    break;
  case LBMI_TRT:
    if constexpr (NVEL == 19) {
      hipLaunchKernelGGL((k_propagate_collide_halo<NVEL, LBMI_TRT, HIO>), grid, block,
			 lds, st, kp, f, fp, h, i0, i1, nblk);
      break;
    }
    return (int) hipErrorInvalidValue;
  default:
    return (int) hipErrorInvalidValue;
  }
  return (int) hipGetLastError();
}

static int host_plane_size(const lbmi_kparam_t * kp, int dir);

template <int NVEL, int DIM, bool HIO>
static int launch_pc_face(const lbmi_kparam_t & kp, const double * f, double * fp,
			  const lbmi_hydro_dev_t & h, int wrapmask,
			  const lbmi_xbuf_t & xb, hipStream_t st) {
  const int psz = host_plane_size(&kp, DIM);
  dim3 grid((unsigned) ((psz + BLOCK - 1)/BLOCK), (kp.nlocal[DIM] > 1) ? 2u : 1u), block(BLOCK);
  switch (kp.scheme) {
  case LBMI_M10:
    hipLaunchKernelGGL((k_propagate_collide_face<NVEL, LBMI_M10, DIM, HIO>), grid, block, 0, st,
		       kp, f, fp, h, wrapmask, xb);
    break;
  case LBMI_BGK:
    hipLaunchKernelGGL((k_propagate_collide_face<NVEL, LBMI_BGK, DIM, HIO>), grid, block, 0, st,
		       kp, f, fp, h, wrapmask, xb);
    break;
  case LBMI_TRT:
    if constexpr (NVEL == 19) {
      hipLaunchKernelGGL((k_propagate_collide_face<NVEL, LBMI_TRT, DIM, HIO>), grid, block, 0, st,
			 kp, f, fp, h, wrapmask, xb);
      break;
    }
    return (int) hipErrorInvalidValue;
  default:
    return (int) hipErrorInvalidValue;
  }
  return (int) hipGetLastError();
}

extern "C" int lbmi_k_propagate_collide_face(const lbmi_kparam_t * kp, int dim,
					     const double * f, double * fprime,
					     const lbmi_hydro_dev_t * h,
					     int wrapmask, const lbmi_xbuf_t * xb,
					     void * stream) {
  hipStream_t st = (hipStream_t) stream;
  const bool hio = (h->force != nullptr || h->rho != nullptr || h->u != nullptr);
  if (h->noise != nullptr || xb == nullptr || (dim != 1 && dim != 2)) return (int) hipErrorInvalidValue;
  if (kp->nvel == 19) {
    if (dim == 1) return hio ? launch_pc_face<19, 1, true>(*kp, f, fprime, *h, wrapmask, *xb, st)
      : launch_pc_face<19, 1, false>(*kp, f, fprime, *h, wrapmask, *xb, st);
    return hio ? launch_pc_face<19, 2, true>(*kp, f, fprime, *h, wrapmask, *xb, st)
      : launch_pc_face<19, 2, false>(*kp, f, fprime, *h, wrapmask, *xb, st);
  }
  if (kp->nvel == 27) {
    if (dim == 1) return hio ? launch_pc_face<27, 1, true>(*kp, f, fprime, *h, wrapmask, *xb, st)
      : launch_pc_face<27, 1, false>(*kp, f, fprime, *h, wrapmask, *xb, st);
    return hio ? launch_pc_face<27, 2, true>(*kp, f, fprime, *h, wrapmask, *xb, st)
      : launch_pc_face<27, 2, false>(*kp, f, fprime, *h, wrapmask, *xb, st);
  }
  return (int) hipErrorInvalidValue;
}

extern "C" int lbmi_k_propagate_collide_halo(const lbmi_kparam_t * kp,
					     const double * f, double * fprime,
					     const lbmi_hydro_dev_t * h,
					     void * stream) {
  hipStream_t st = (hipStream_t) stream;
  const bool hio = (h->force != nullptr || h->rho != nullptr || h->u != nullptr);
  if (h->noise != nullptr) return (int) hipErrorInvalidValue;
  if (kp->nvel == 19) {
    return hio ? launch_pc_halo<19, true>(*kp, f, fprime, *h, st)
      : launch_pc_halo<19, false>(*kp, f, fprime, *h, st);
  }
  if (kp->nvel == 27) {
    return hio ? launch_pc_halo<27, true>(*kp, f, fprime, *h, st)
      : launch_pc_halo<27, false>(*kp, f, fprime, *h, st);
  }
  return (int) hipErrorInvalidValue;
}

extern "C" int lbmi_k_hydro_from_f(const lbmi_kparam_t * kp, const double * f,
				   const lbmi_hydro_dev_t * h, int blocked,
				   void * stream) {
  hipStream_t st = (hipStream_t) stream;
  Range1D r = interior_range(*kp);
  dim3 grid(r.grid), block(BLOCK);
  if (kp->nvel == 19 && !blocked) hipLaunchKernelGGL((k_hydro_from_f<19, false>), grid, block, 0, st, *kp, f, *h, r.i0, r.i1, r.nblk);
  else if (kp->nvel == 19) hipLaunchKernelGGL((k_hydro_from_f<19, true>), grid, block, 0, st, *kp, f, *h, r.i0, r.i1, r.nblk);
  else if (kp->nvel == 27 && !blocked) hipLaunchKernelGGL((k_hydro_from_f<27, false>), grid, block, 0, st, *kp, f, *h, r.i0, r.i1, r.nblk);
  else if (kp->nvel == 27) hipLaunchKernelGGL((k_hydro_from_f<27, true>), grid, block, 0, st, *kp, f, *h, r.i0, r.i1, r.nblk);
  else return (int) hipErrorInvalidValue;
  return (int) hipGetLastError();
}

extern "C" int lbmi_k_blocked_sites(const lbmi_kparam_t * kp) {
  /* sites representable in the blocked order: whole blocks only */
  return (int) ((kp->nsite/LBW)*LBW);
}

extern "C" int lbmi_k_relayout_n(const lbmi_kparam_t * kp, int ndist,
				 const double * src, double * dst,
				 int to_blocked, void * stream) {
  hipStream_t st = (hipStream_t) stream;
  int nfull = lbmi_k_blocked_sites(kp);
  unsigned nblk = (unsigned) ((nfull + BLOCK - 1)/BLOCK);
  dim3 grid(grid_for(nblk, (unsigned) kp->xcd_group)), block(BLOCK);
  if (ndist == 1) return lbmi_k_relayout(kp, src, dst, to_blocked, stream);
  if (ndist != 2) return (int) hipErrorInvalidValue;
  if (kp->nvel == 19) {
    if (to_blocked) hipLaunchKernelGGL((k_relayout<38, true>), grid, block, 0, st, *kp, src, dst, nfull, nblk);
    else hipLaunchKernelGGL((k_relayout<38, false>), grid, block, 0, st, *kp, src, dst, nfull, nblk);
  }
  else if (kp->nvel == 27) {
    if (to_blocked) hipLaunchKernelGGL((k_relayout<54, true>), grid, block, 0, st, *kp, src, dst, nfull, nblk);
    else hipLaunchKernelGGL((k_relayout<54, false>), grid, block, 0, st, *kp, src, dst, nfull, nblk);
  }
  else {
    return (int) hipErrorInvalidValue;
  }
  return (int) hipGetLastError();
}

extern "C" int lbmi_k_relayout(const lbmi_kparam_t * kp, const double * src,
			       double * dst, int to_blocked, void * stream) {
  hipStream_t st = (hipStream_t) stream;
  int nfull = lbmi_k_blocked_sites(kp);
  unsigned nblk = (unsigned) ((nfull + BLOCK - 1)/BLOCK);
  dim3 grid(grid_for(nblk, (unsigned) kp->xcd_group)), block(BLOCK);
  if (kp->nvel == 19) {
    if (to_blocked) hipLaunchKernelGGL((k_relayout<19, true>), grid, block, 0, st, *kp, src, dst, nfull, nblk);
    else hipLaunchKernelGGL((k_relayout<19, false>), grid, block, 0, st, *kp, src, dst, nfull, nblk);
  }
  else if (kp->nvel == 27) {
    if (to_blocked) hipLaunchKernelGGL((k_relayout<27, true>), grid, block, 0, st, *kp, src, dst, nfull, nblk);
    else hipLaunchKernelGGL((k_relayout<27, false>), grid, block, 0, st, *kp, src, dst, nfull, nblk);
  }
  else {
    return (int) hipErrorInvalidValue;
  }
  return (int) hipGetLastError();
}

extern "C" int lbmi_k_aa_even(const lbmi_kparam_t * kp, double * f,
			      const lbmi_hydro_dev_t * h, void * stream) {
  hipStream_t st = (hipStream_t) stream;
  if (kp->nvel == 19) return launch_aa_even<19>(*kp, f, *h, st);
  if (kp->nvel == 27) return launch_aa_even<27>(*kp, f, *h, st);
  return (int) hipErrorInvalidValue;
}

extern "C" int lbmi_k_aa_unswap(const lbmi_kparam_t * kp, double * f,
				void * stream);

extern "C" int lbmi_k_aa_odd(const lbmi_kparam_t * kp, double * f,
			     const lbmi_hydro_dev_t * h, int wrapmask,
			     int swapped_in, void * stream) {
  hipStream_t st = (hipStream_t) stream;
  if (!swapped_in) {
    /* The pull-collide-push is race-free only from the slot-swapped order
     * (every thread then reads and writes the SAME addresses): from the
     * normal order a thread would push into locations its neighbours still
     * have to pull from. After a flush, swap the slots first (a per-site
     * permutation, in place). */
    int ifail = lbmi_k_aa_unswap(kp, f, stream);
    if (ifail) return ifail;
  }
  if (kp->nvel == 19) return launch_aa_odd<19, true>(*kp, f, *h, wrapmask, st);
  if (kp->nvel == 27) return launch_aa_odd<27, true>(*kp, f, *h, wrapmask, st);
  return (int) hipErrorInvalidValue;
}

extern "C" int lbmi_k_aa_unswap(const lbmi_kparam_t * kp, double * f,
				void * stream) {
  hipStream_t st = (hipStream_t) stream;
  Range1D r = interior_range(*kp);
  dim3 grid(r.grid), block(BLOCK);
  if (kp->nvel == 19) {
    hipLaunchKernelGGL((k_aa_unswap<19>), grid, block, r.lds, st, *kp, f,
		       r.i0, r.i1, r.nblk);
  }
  else if (kp->nvel == 27) {
    hipLaunchKernelGGL((k_aa_unswap<27>), grid, block, r.lds, st, *kp, f,
		       r.i0, r.i1, r.nblk);
  }
  else {
    return (int) hipErrorInvalidValue;
  }
  return (int) hipGetLastError();
}

extern "C" int lbmi_k_unpropagate_wrap(const lbmi_kparam_t * kp,
				       const double * f, double * fprime,
				       int wrapmask, void * stream) {
  hipStream_t st = (hipStream_t) stream;
  Range1D r = interior_range(*kp);
  dim3 grid(r.grid), block(BLOCK);
  if (kp->nvel == 19) {
    hipLaunchKernelGGL((k_unpropagate_wrap<19>), grid, block, r.lds, st, *kp,
		       f, fprime, wrapmask, r.i0, r.i1, r.nblk);
  }
  else if (kp->nvel == 27) {
    hipLaunchKernelGGL((k_unpropagate_wrap<27>), grid, block, r.lds, st, *kp,
		       f, fprime, wrapmask, r.i0, r.i1, r.nblk);
  }
  else {
    return (int) hipErrorInvalidValue;
  }
  return (int) hipGetLastError();
}

extern "C" int lbmi_k_halo_copy(const lbmi_kparam_t * kp, int dir,
				const lbmi_halo_sel_t * sel, double * data,
				int nswap, void * stream) {
  hipStream_t st = (hipStream_t) stream;
  int psz = (dir == 0) ? kp->nall[1]*kp->nall[2]
    : ((dir == 1) ? kp->nall[0]*kp->nall[2] : kp->nall[0]*kp->nall[1]);
  int ncomp = sel->nlo + sel->nhi;
  if (ncomp == 0) return 0;
  if (nswap < 1 || nswap > kp->nhalo || nswap > kp->nlocal[dir]) {
    return (int) hipErrorInvalidValue;
  }
  dim3 grid((psz + BLOCK - 1)/BLOCK, ncomp, nswap), block(BLOCK);
  hipLaunchKernelGGL(k_halo_copy, grid, block, 0, st, *kp, dir, *sel, data);
  return (int) hipGetLastError();
}

static int host_plane_size(const lbmi_kparam_t * kp, int dir) {
  return (dir == 0) ? kp->nall[1]*kp->nall[2]
    : ((dir == 1) ? kp->nall[0]*kp->nall[2] : kp->nall[0]*kp->nall[1]);
}

extern "C" int lbmi_k_halo_pack(const lbmi_kparam_t * kp, int dir,
				const lbmi_halo_sel_t * sel,
				const double * data, double * buf_lo,
				double * buf_hi, int blocked, int layer,
				void * stream) {
  hipStream_t st = (hipStream_t) stream;
  int ncomp = sel->nlo + sel->nhi;
  if (ncomp == 0) return 0;
  if (dir < 0 || dir > 2) return (int) hipErrorInvalidValue;
  dim3 grid((host_plane_size(kp, dir) + BLOCK - 1)/BLOCK, ncomp), block(BLOCK);
  hipLaunchKernelGGL(k_halo_pack_x, grid, block, 0, st, *kp, dir, *sel, data,
		     buf_lo, buf_hi, blocked, layer);
  return (int) hipGetLastError();
}

extern "C" int lbmi_k_halo_unpack(const lbmi_kparam_t * kp, int dir,
				  const lbmi_halo_sel_t * sel,
				  double * data, const double * buf_lo,
				  const double * buf_hi, int blocked,
				  int layer, void * stream) {
  hipStream_t st = (hipStream_t) stream;
  int ncomp = sel->nlo + sel->nhi;
  if (ncomp == 0) return 0;
  if (dir < 0 || dir > 2) return (int) hipErrorInvalidValue;
  dim3 grid((host_plane_size(kp, dir) + BLOCK - 1)/BLOCK, ncomp), block(BLOCK);
  hipLaunchKernelGGL(k_halo_unpack_x, grid, block, 0, st, *kp, dir, *sel, data,
		     buf_lo, buf_hi, blocked, layer);
  return (int) hipGetLastError();
}

extern "C" int lbmi_k_halo_pack_x(const lbmi_kparam_t * kp,
				  const lbmi_halo_sel_t * sel,
				  const double * data, double * buf_lo,
				  double * buf_hi, int blocked, int layer,
				  void * stream) {
  return lbmi_k_halo_pack(kp, 0, sel, data, buf_lo, buf_hi, blocked, layer, stream);
}

extern "C" int lbmi_k_halo_unpack_x(const lbmi_kparam_t * kp,
				    const lbmi_halo_sel_t * sel,
				    double * data, const double * buf_lo,
				    const double * buf_hi, int blocked,
				    int layer, void * stream) {
  return lbmi_k_halo_unpack(kp, 0, sel, data, buf_lo, buf_hi, blocked, layer, stream);
}

template <int NCOMP, int RBT>
static int launch_records(const lbmi_kparam_t & kp, double * f, double * rec,
			  int pack, hipStream_t st) {
  long long nint = (long long) kp.nlocal[0]*kp.nlocal[1]*kp.nlocal[2];
  dim3 grid((unsigned) ((nint + RBT - 1)/RBT)), block(RBT);
  if (pack) hipLaunchKernelGGL((k_records<NCOMP, true, RBT>), grid, block, 0, st, kp, f, rec, nint);
  else hipLaunchKernelGGL((k_records<NCOMP, false, RBT>), grid, block, 0, st, kp, f, rec, nint);
  return (int) hipGetLastError();
}

extern "C" int lbmi_k_records(const lbmi_kparam_t * kp, int ndist, double * f,
			      double * rec, int pack, void * stream) {
  hipStream_t st = (hipStream_t) stream;
  switch (kp->nvel*ndist) {
  case 19: return launch_records<19, RB1>(*kp, f, rec, pack, st);
  case 27: return launch_records<27, RB1>(*kp, f, rec, pack, st);
  case 38: return launch_records<38, RB2>(*kp, f, rec, pack, st);
  case 54: return launch_records<54, RB2>(*kp, f, rec, pack, st);
  default: break;
  }
  return (int) hipErrorInvalidValue;
}

extern "C" int lbmi_k_field_set(const lbmi_kparam_t * kp, int ncomp,
				double * field, const double * v,
				void * stream) {
  hipStream_t st = (hipStream_t) stream;
  if (ncomp < 1 || ncomp > 3) return (int) hipErrorInvalidValue;
  dim3 grid((unsigned) ((kp->nsite + BLOCK - 1)/BLOCK)), block(BLOCK);
  hipLaunchKernelGGL(k_field_set, grid, block, 0, st, kp->nsite, ncomp, field,
		     v[0], (ncomp > 1) ? v[1] : 0.0, (ncomp > 2) ? v[2] : 0.0);
  return (int) hipGetLastError();
}

extern "C" int lbmi_k_grad(const lbmi_kparam_t * kp, int npt,
			   const double * phi, double * grad, double * delsq,
			   void * stream) {
  hipStream_t st = (hipStream_t) stream;
  int ne = kp->nhalo - 1;
  int i0 = (kp->nhalo - ne)*kp->strx;
  int i1 = (kp->nhalo + kp->nlocal[0] + ne)*kp->strx;
  int i0a = (i0/LBMI_ALIGN)*LBMI_ALIGN;
  unsigned nblk = (unsigned) ((i1 - i0a + BLOCK - 1)/BLOCK);
  dim3 grid(grid_for(nblk, (unsigned) kp->xcd_group)), block(BLOCK);
  if (npt == 27) {
    hipLaunchKernelGGL((k_grad<27>), grid, block, 0, st, *kp, phi, grad,
		       delsq, i0, i1, nblk);
  }
  else {
    hipLaunchKernelGGL((k_grad<7>), grid, block, 0, st, *kp, phi, grad, delsq,
		       i0, i1, nblk);
  }
  return (int) hipGetLastError();
}

extern "C" int lbmi_k_symm_force(const lbmi_kparam_t * kp, int npt, double a,
				 double b, double kappa, const double * phi,
				 const double * grad, const double * delsq,
				 double * force, void * stream) {
  hipStream_t st = (hipStream_t) stream;
  Range1D r = interior_range(*kp);
  dim3 grid(r.grid), block(BLOCK);
  Symm q = {a, b, kappa};
  if (grad && delsq) {
    hipLaunchKernelGGL((k_symm_force<true, 7>), grid, block, 0, st, *kp, q,
		       phi, grad, delsq, force, r.i0, r.i1, r.nblk);
  }
  else if (npt == 27) {
    hipLaunchKernelGGL((k_symm_force<false, 27>), grid, block, 0, st, *kp, q,
		       phi, grad, delsq, force, r.i0, r.i1, r.nblk);
  }
  else {
    hipLaunchKernelGGL((k_symm_force<false, 7>), grid, block, 0, st, *kp, q,
		       phi, grad, delsq, force, r.i0, r.i1, r.nblk);
  }
  return (int) hipGetLastError();
}

extern "C" int lbmi_k_cahn_hilliard(const lbmi_kparam_t * kp, int npt,
				    int order, double a, double b,
				    double kappa, double mobility,
				    const double * phi, const double * delsq,
				    const double * u, double * phi_out,
				    void * stream) {
  hipStream_t st = (hipStream_t) stream;
  Range1D r = interior_range(*kp);
  dim3 grid(r.grid), block(BLOCK);
  Symm q = {a, b, kappa};
  if (delsq) {
    hipLaunchKernelGGL((k_cahn_hilliard<true, 7>), grid, block, 0, st, *kp, q,
		       mobility, order, phi, delsq, u, phi_out, r.i0, r.i1,
		       r.nblk);
  }
  else if (npt == 27) {
    hipLaunchKernelGGL((k_cahn_hilliard<false, 27>), grid, block, 0, st, *kp,
		       q, mobility, order, phi, delsq, u, phi_out, r.i0, r.i1,
		       r.nblk);
  }
  else {
    hipLaunchKernelGGL((k_cahn_hilliard<false, 7>), grid, block, 0, st, *kp,
		       q, mobility, order, phi, delsq, u, phi_out, r.i0, r.i1,
		       r.nblk);
  }
  return (int) hipGetLastError();
}

/* npt = 7 or 27: everything from phi; npt = 0: from the arrays grad, delsq */

template <bool ACCUMULATE>
static void launch_fe_step(const lbmi_kparam_t * kp, int npt, int order,
			   Symm q, double mobility, const double * phi,
			   const double * grad, const double * delsq,
			   const double * u, double * force, double * phi_out,
			   int wrap, hipStream_t st) {
  if (npt != 0 && (kp->fe_tiled || wrap)) {
    /* phi through an LDS tile */
    const int ntx = (kp->nlocal[0] + FT_X - 1)/FT_X;
    const int nty = (kp->nlocal[1] + FT_Y - 1)/FT_Y;
    const int ntz = (kp->nlocal[2] + FT_Z - 1)/FT_Z;
    const unsigned nblk = (unsigned) ntx*(unsigned) nty*(unsigned) ntz;
    dim3 tgrid(grid_for(nblk, (unsigned) kp->xcd_group)), tblock(FT_Y*FT_Z);
    if (wrap) {
      if (npt == 27) {
	hipLaunchKernelGGL((k_symm_fe_step_tiled<ACCUMULATE, 27, true>), tgrid,
			   tblock, 0, st, *kp, q, mobility, order, phi, u,
			   force, phi_out, nty, ntz, nblk);
      }
      else {
	hipLaunchKernelGGL((k_symm_fe_step_tiled<ACCUMULATE, 7, true>), tgrid,
			   tblock, 0, st, *kp, q, mobility, order, phi, u,
			   force, phi_out, nty, ntz, nblk);
      }
    }
    else if (npt == 27) {
      hipLaunchKernelGGL((k_symm_fe_step_tiled<ACCUMULATE, 27, false>), tgrid,
			 tblock, 0, st, *kp, q, mobility, order, phi, u, force,
			 phi_out, nty, ntz, nblk);
    }
    else {
      hipLaunchKernelGGL((k_symm_fe_step_tiled<ACCUMULATE, 7, false>), tgrid,
			 tblock, 0, st, *kp, q, mobility, order, phi, u, force,
			 phi_out, nty, ntz, nblk);
    }
    return;
  }
  Range1D r = interior_range(*kp);
  dim3 grid(r.grid), block(BLOCK);
  if (npt == 0) {
    hipLaunchKernelGGL((k_symm_fe_step<ACCUMULATE, 0>), grid, block, 0, st,
		       *kp, q, mobility, order, phi, grad, delsq, u, force,
		       phi_out, r.i0, r.i1, r.nblk);
  }
  else if (npt == 27) {
    hipLaunchKernelGGL((k_symm_fe_step<ACCUMULATE, 27>), grid, block, 0, st,
		       *kp, q, mobility, order, phi, grad, delsq, u, force,
		       phi_out, r.i0, r.i1, r.nblk);
  }
  else {
    hipLaunchKernelGGL((k_symm_fe_step<ACCUMULATE, 7>), grid, block, 0, st,
		       *kp, q, mobility, order, phi, grad, delsq, u, force,
		       phi_out, r.i0, r.i1, r.nblk);
  }
}

/* k_symm_lb_step over the interior x planes; lay as lbmi_k_propagate_collide
 * (0 SoA -> SoA, 1 SoA -> blocked, 2 blocked -> blocked; kp->nt_store bit 0:
 * nontemporal stores of the blocked state) */

template <int NVEL, int LAY>
static int launch_symm_lb(const lbmi_kparam_t & kp, const double * f,
			  double * fp, const lbmi_hydro_dev_t & h,
			  const Symm & q, double mobility, int order,
			  const double * phi, const double * uprev,
			  double * phi_out, hipStream_t st) {
  constexpr int ALIGNV = ((LAY & 3) == 0) ? LBMI_ALIGN : LBW;
  const int i0 = kp.nhalo*kp.strx;
  const int i1 = (kp.nhalo + kp.nlocal[0])*kp.strx;
  const int i0a = (i0/ALIGNV)*ALIGNV;
  const unsigned nblk = (unsigned) ((i1 - i0a + BLOCK - 1)/BLOCK);
  /* stripes (kp.fe_stripes; measured slower, off by default): every XCD an
   * eighth of every plane */
  const int gstripe = kp.fe_stripes ? (kp.strx + 8*BLOCK - 1)/(8*BLOCK) : 0;
  dim3 grid(gstripe ? (unsigned) (kp.nlocal[0]*gstripe*8)
	    : grid_for(nblk, (unsigned) kp.fe_xcd_group)), block(BLOCK);
  /* no occupancy cap here (launch_pc_hio has one): with the free-energy
   * arithmetic in front of the collision the kernel is bound by latency and
   * issue as much as by HBM, and every resident wave helps
   * (profiles/r03_cfg4_sweep.txt: 0.164 -> 0.154 ms per step at 128^3) */
  const unsigned lds = 0u;
  switch (kp.scheme) {
  case LBMI_M10:
    hipLaunchKernelGGL((k_symm_lb_step<NVEL, LBMI_M10, LAY>), grid, block, lds,
		       st, kp, f, fp, h, q, mobility, order, phi, uprev,
		       phi_out, i0, i1, nblk, gstripe);
    break;
  case LBMI_BGK:
    hipLaunchKernelGGL((k_symm_lb_step<NVEL, LBMI_BGK, LAY>), grid, block, lds,
		       st, kp, f, fp, h, q, mobility, order, phi, uprev,
		       phi_out, i0, i1, nblk, gstripe);
    break;
  default:
    return (int) hipErrorInvalidValue;
  }
  return (int) hipGetLastError();
}

extern "C" int lbmi_k_symm_lb_step(const lbmi_kparam_t * kp, const double * f,
				   double * fprime, const lbmi_hydro_dev_t * h,
				   double a, double b, double kappa,
				   double mobility, int order,
				   const double * phi, const double * uprev,
				   double * phi_out, int lay, void * stream) {
  hipStream_t st = (hipStream_t) stream;
  const Symm q = {a, b, kappa};
  if (kp->nvel == 19) {
    if (lay == 0) return launch_symm_lb<19, 0>(*kp, f, fprime, *h, q, mobility, order, phi, uprev, phi_out, st);
#if LBMI_BLOCK*LBMI_SPT == 256
    if (kp->nt_store & 1) {
      if (lay == 1) return launch_symm_lb<19, 5>(*kp, f, fprime, *h, q, mobility, order, phi, uprev, phi_out, st);
      if (lay == 2) return launch_symm_lb<19, 6>(*kp, f, fprime, *h, q, mobility, order, phi, uprev, phi_out, st);
    }
    if (lay == 1) return launch_symm_lb<19, 1>(*kp, f, fprime, *h, q, mobility, order, phi, uprev, phi_out, st);
    if (lay == 2) return launch_symm_lb<19, 2>(*kp, f, fprime, *h, q, mobility, order, phi, uprev, phi_out, st);
#endif
  }
  if (kp->nvel == 27) {
    if (lay == 0) return launch_symm_lb<27, 0>(*kp, f, fprime, *h, q, mobility, order, phi, uprev, phi_out, st);
#if LBMI_BLOCK*LBMI_SPT == 256
    if (kp->nt_store & 1) {
      if (lay == 1) return launch_symm_lb<27, 5>(*kp, f, fprime, *h, q, mobility, order, phi, uprev, phi_out, st);
      if (lay == 2) return launch_symm_lb<27, 6>(*kp, f, fprime, *h, q, mobility, order, phi, uprev, phi_out, st);
    }
    if (lay == 1) return launch_symm_lb<27, 1>(*kp, f, fprime, *h, q, mobility, order, phi, uprev, phi_out, st);
    if (lay == 2) return launch_symm_lb<27, 2>(*kp, f, fprime, *h, q, mobility, order, phi, uprev, phi_out, st);
#endif
  }
  return (int) hipErrorInvalidValue;
}

extern "C" int lbmi_k_symm_fe_step(const lbmi_kparam_t * kp, int npt,
				   int order, double a, double b,
				   double kappa, double mobility,
				   const double * phi, const double * grad,
				   const double * delsq, const double * u,
				   double * force, double * phi_out,
				   int accumulate, int wrap, void * stream) {
  hipStream_t st = (hipStream_t) stream;
  Symm q = {a, b, kappa};
  if (accumulate) {
    launch_fe_step<true>(kp, npt, order, q, mobility, phi, grad, delsq, u,
			 force, phi_out, wrap, st);
  }
  else {
    launch_fe_step<false>(kp, npt, order, q, mobility, phi, grad, delsq, u,
			  force, phi_out, wrap, st);
  }
  return (int) hipGetLastError();
}

extern "C" int lbmi_k_phi_from_g(const lbmi_kparam_t * kp, const double * f2,
				 double * phi, int pull, int wrapmask,
				 int blocked, void * stream) {
  hipStream_t st = (hipStream_t) stream;
  Range1D r = interior_range(*kp);
  dim3 grid(r.grid), block(BLOCK);
  if (blocked && !pull) return (int) hipErrorInvalidValue;
  if (kp->nvel == 19 && !pull) {
    hipLaunchKernelGGL((k_phi_from_g<19, false>), grid, block, 0, st, *kp, f2, phi,
		       wrapmask, r.i0, r.i1, r.nblk);
  }
  else if (kp->nvel == 19 && !blocked) {
    hipLaunchKernelGGL((k_phi_from_g<19, true>), grid, block, 0, st, *kp, f2, phi,
		       wrapmask, r.i0, r.i1, r.nblk);
  }
  else if (kp->nvel == 19) {
    hipLaunchKernelGGL((k_phi_from_g<19, true, true>), grid, block, 0, st, *kp, f2, phi,
		       wrapmask, r.i0, r.i1, r.nblk);
  }
  else if (kp->nvel == 27 && !pull) {
    hipLaunchKernelGGL((k_phi_from_g<27, false>), grid, block, 0, st, *kp, f2, phi,
		       wrapmask, r.i0, r.i1, r.nblk);
  }
  else if (kp->nvel == 27 && !blocked) {
    hipLaunchKernelGGL((k_phi_from_g<27, true>), grid, block, 0, st, *kp, f2, phi,
		       wrapmask, r.i0, r.i1, r.nblk);
  }
  else if (kp->nvel == 27) {
    hipLaunchKernelGGL((k_phi_from_g<27, true, true>), grid, block, 0, st, *kp, f2, phi,
		       wrapmask, r.i0, r.i1, r.nblk);
  }
  else {
    return (int) hipErrorInvalidValue;
  }
  return (int) hipGetLastError();
}

/* the pull variants in the blocked order (no fluctuations there) */
template <int NVEL, int LAY>
static int launch_collide_binary_lay(const lbmi_kparam_t & kp, const double * src,
				     double * f2, const lbmi_hydro_dev_t & h,
				     Symm q, double rtau2, const double * phi,
				     const double * grad, const double * delsq,
				     int wrapmask, hipStream_t st) {
  constexpr int ALIGNV = ((LAY & 3) == 0) ? LBMI_ALIGN : LBW;
  const int i0 = kp.nhalo*kp.strx;
  const int i1 = (kp.nhalo + kp.nlocal[0])*kp.strx;
  const int i0a = (i0/ALIGNV)*ALIGNV;
  const unsigned nblk = (unsigned) ((i1 - i0a + BLOCK - 1)/BLOCK);
  dim3 grid(grid_for(nblk, (unsigned) kp.xcd_group)), block(BLOCK);
  switch (kp.scheme) {
  case LBMI_M10:
    hipLaunchKernelGGL((k_collide_binary<NVEL, LBMI_M10, true, false, LAY>), grid, block, 0, st,
		       kp, src, f2, h, q, rtau2, phi, grad, delsq, wrapmask, i0, i1, nblk);
    break;
  case LBMI_BGK:
    hipLaunchKernelGGL((k_collide_binary<NVEL, LBMI_BGK, true, false, LAY>), grid, block, 0, st,
		       kp, src, f2, h, q, rtau2, phi, grad, delsq, wrapmask, i0, i1, nblk);
    break;
  case LBMI_TRT:
    if constexpr (NVEL == 19) {
      hipLaunchKernelGGL((k_collide_binary<NVEL, LBMI_TRT, true, false, LAY>), grid, block, 0, st,
			 kp, src, f2, h, q, rtau2, phi, grad, delsq, wrapmask, i0, i1, nblk);
      break;
    }
    return (int) hipErrorInvalidValue;
  default:
    return (int) hipErrorInvalidValue;
  }
  return (int) hipGetLastError();
}

template <int NVEL, bool PULL>
static int launch_collide_binary(const lbmi_kparam_t & kp, const double * src,
				 double * f2,
				 const lbmi_hydro_dev_t & h, Symm q,
				 double rtau2, const double * phi,
				 const double * grad, const double * delsq,
				 int wrapmask, hipStream_t st) {
  Range1D r = interior_range(kp);
  dim3 grid(r.grid), block(BLOCK);
  if (h.noise != nullptr) {
    if constexpr (NVEL == 19) {
      switch (kp.scheme) {
      case LBMI_M10:
	hipLaunchKernelGGL((k_collide_binary<19, LBMI_M10, PULL, true>), grid, block, 0, st,
			   kp, src, f2, h, q, rtau2, phi, grad, delsq, wrapmask, r.i0, r.i1, r.nblk);
	break;
      case LBMI_BGK:
	hipLaunchKernelGGL((k_collide_binary<19, LBMI_BGK, PULL, true>), grid, block, 0, st,
			   kp, src, f2, h, q, rtau2, phi, grad, delsq, wrapmask, r.i0, r.i1, r.nblk);
	break;
      case LBMI_TRT:
	hipLaunchKernelGGL((k_collide_binary<19, LBMI_TRT, PULL, true>), grid, block, 0, st,
			   kp, src, f2, h, q, rtau2, phi, grad, delsq, wrapmask, r.i0, r.i1, r.nblk);
	break;
      default:
	return (int) hipErrorInvalidValue;
      }
      return (int) hipGetLastError();
    }
    return (int) hipErrorInvalidValue;
  }
  switch (kp.scheme) {
  case LBMI_M10:
    hipLaunchKernelGGL((k_collide_binary<NVEL, LBMI_M10, PULL>), grid, block, 0, st,
		       kp, src, f2, h, q, rtau2, phi, grad, delsq, wrapmask, r.i0, r.i1, r.nblk);
    break;
  case LBMI_BGK:
    hipLaunchKernelGGL((k_collide_binary<NVEL, LBMI_BGK, PULL>), grid, block, 0, st,
		       kp, src, f2, h, q, rtau2, phi, grad, delsq, wrapmask, r.i0, r.i1, r.nblk);
    break;
  case LBMI_TRT:
    if constexpr (NVEL == 19) {
      hipLaunchKernelGGL((k_collide_binary<NVEL, LBMI_TRT, PULL>), grid, block, 0,
			 st, kp, src, f2, h, q, rtau2, phi, grad, delsq, wrapmask,
			 r.i0, r.i1, r.nblk);
      break;
    }
    return (int) hipErrorInvalidValue;
  default:
    return (int) hipErrorInvalidValue;
  }
  return (int) hipGetLastError();
}

extern "C" int lbmi_k_collide_binary(const lbmi_kparam_t * kp,
				     const double * src, double * f2,
				     const lbmi_hydro_dev_t * h, double a,
				     double b, double kappa, double rtau2,
				     const double * phi, const double * grad,
				     const double * delsq, int wrapmask,
				     int lay, void * stream) {
  hipStream_t st = (hipStream_t) stream;
  Symm q = {a, b, kappa};
  /* src == f2: in place; otherwise pull from src (propagation fused in) */
  if (lay != 0) {
#if LBMI_BLOCK*LBMI_SPT == 256
    /* the deferred state in the blocked order: pull, no fluctuations */
    if (src == f2 || h->noise != nullptr) return (int) hipErrorInvalidValue;
    const int l = lay + ((kp->nt_store & 1) ? 4 : 0);
    if (kp->nvel == 19) {
      if (l == 1) return launch_collide_binary_lay<19, 1>(*kp, src, f2, *h, q, rtau2, phi, grad, delsq, wrapmask, st);
      if (l == 2) return launch_collide_binary_lay<19, 2>(*kp, src, f2, *h, q, rtau2, phi, grad, delsq, wrapmask, st);
      if (l == 5) return launch_collide_binary_lay<19, 5>(*kp, src, f2, *h, q, rtau2, phi, grad, delsq, wrapmask, st);
      if (l == 6) return launch_collide_binary_lay<19, 6>(*kp, src, f2, *h, q, rtau2, phi, grad, delsq, wrapmask, st);
    }
    if (kp->nvel == 27) {
      if (l == 1) return launch_collide_binary_lay<27, 1>(*kp, src, f2, *h, q, rtau2, phi, grad, delsq, wrapmask, st);
      if (l == 2) return launch_collide_binary_lay<27, 2>(*kp, src, f2, *h, q, rtau2, phi, grad, delsq, wrapmask, st);
      if (l == 5) return launch_collide_binary_lay<27, 5>(*kp, src, f2, *h, q, rtau2, phi, grad, delsq, wrapmask, st);
      if (l == 6) return launch_collide_binary_lay<27, 6>(*kp, src, f2, *h, q, rtau2, phi, grad, delsq, wrapmask, st);
    }
#endif
    return (int) hipErrorInvalidValue;
  }
  if (src == f2) {
    if (kp->nvel == 19) return launch_collide_binary<19, false>(*kp, f2, f2, *h, q, rtau2, phi, grad, delsq, 0, st);
    if (kp->nvel == 27) return launch_collide_binary<27, false>(*kp, f2, f2, *h, q, rtau2, phi, grad, delsq, 0, st);
  }
  else {
    if (kp->nvel == 19) return launch_collide_binary<19, true>(*kp, src, f2, *h, q, rtau2, phi, grad, delsq, wrapmask, st);
    if (kp->nvel == 27) return launch_collide_binary<27, true>(*kp, src, f2, *h, q, rtau2, phi, grad, delsq, wrapmask, st);
  }
  return (int) hipErrorInvalidValue;
}

extern "C" int lbmi_k_wall_nblk(int nlink) {
  int nblk = (nlink + WALL_BLOCK - 1)/WALL_BLOCK;
  if (nblk > WALL_NBLK_MAX) nblk = WALL_NBLK_MAX;
  if (nblk < 1) nblk = 1;
  return nblk;
}

extern "C" int lbmi_k_wall_bbl(const lbmi_kparam_t * kp,
			       const lbmi_wall_tab_t * tab, double * f,
			       int nlink, const int * linki,
			       const int * linkj, const int * linkp,
			       const int * linku, const char * status,
			       double * part, double * fnet, int * err,
			       void * stream) {
  hipStream_t st = (hipStream_t) stream;
  int nblk = lbmi_k_wall_nblk(nlink);
  if (nlink <= 0) return 0;
  hipLaunchKernelGGL(k_wall_bbl, dim3(nblk), dim3(WALL_BLOCK), 0, st, *kp,
		     *tab, f, nlink, linki, linkj, linkp, linku, status, part,
		     err);
  hipLaunchKernelGGL(k_wall_fnet, dim3(1), dim3(64), 0, st, nblk, part, fnet);
  return (int) hipGetLastError();
}

extern "C" int lbmi_k_wall_bbl_slip(const lbmi_kparam_t * kp,
				    const lbmi_wall_tab_t * tab, double * f,
				    int nlink, const int * linki,
				    const int * linkj, const int * linkp,
				    const int * linkk, const int8_t * linkq,
				    const int8_t * links, const char * status,
				    double * part, double * fnet, int * err,
				    void * stream) {
  hipStream_t st = (hipStream_t) stream;
  int nblk = lbmi_k_wall_nblk(nlink);
  if (nlink <= 0) return 0;
  hipLaunchKernelGGL(k_wall_bbl_slip, dim3(nblk), dim3(WALL_BLOCK), 0, st, *kp,
		     *tab, f, nlink, linki, linkj, linkp, linkk, linkq, links,
		     status, part, err);
  hipLaunchKernelGGL(k_wall_fnet, dim3(1), dim3(64), 0, st, nblk, part, fnet);
  return (int) hipGetLastError();
}

extern "C" int lbmi_k_interior_copy(const lbmi_kparam_t * kp, int ncomp,
				    const double * src, double * dst,
				    void * stream) {
  hipStream_t st = (hipStream_t) stream;
  const long long n = (long long) kp->nlocal[0]*kp->strx;
  dim3 grid((unsigned) ((n + BLOCK - 1)/BLOCK)), block(BLOCK);
  hipLaunchKernelGGL(k_interior_copy, grid, block, 0, st, *kp, ncomp, src, dst);
  return (int) hipGetLastError();
}

/* two pointers into two device slots: lb->target->f / fprime after a swap,
 * in stream order, without a blocking copy from the host */
__global__ void k_store_pointers(double ** slot_f, double ** slot_fprime,
				 double * f, double * fprime) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    *slot_f = f;
    *slot_fprime = fprime;
  }
}

extern "C" int lbmi_k_store_pointers(double ** slot_f, double ** slot_fprime,
				     double * f, double * fprime, void * stream) {
  hipLaunchKernelGGL(k_store_pointers, dim3(1), dim3(64), 0, (hipStream_t) stream,
		     slot_f, slot_fprime, f, fprime);
  return (int) hipGetLastError();
}

extern "C" int lbmi_k_density(const lbmi_kparam_t * kp, const double * f,
			      double * rho, void * stream) {
  hipStream_t st = (hipStream_t) stream;
  const long long n = (long long) kp->nlocal[0]*kp->nlocal[1]*kp->nlocal[2];
  dim3 grid((unsigned) ((n + BLOCK - 1)/BLOCK)), block(BLOCK);
  if (kp->nvel == 19) hipLaunchKernelGGL(k_density<19>, grid, block, 0, st, *kp, f, rho, n);
  else if (kp->nvel == 27) hipLaunchKernelGGL(k_density<27>, grid, block, 0, st, *kp, f, rho, n);
  else return (int) hipErrorInvalidValue;
  return (int) hipGetLastError();
}

extern "C" int lbmi_k_field_stats(const lbmi_kparam_t * kp, const double * field,
				  const char * status, double * work,
				  double * out_dev, void * stream) {
  hipStream_t st = (hipStream_t) stream;
  int n = kp->nlocal[0]*kp->strx;
  int nblk = (n + BLOCK - 1)/BLOCK;
  if (nblk > MOM_NBLK) nblk = MOM_NBLK;
  hipLaunchKernelGGL(k_field_stats_partial, dim3(nblk), dim3(BLOCK), 0, st, *kp,
		     field, status, work);
  hipLaunchKernelGGL(k_moments_final, dim3(1), dim3(BLOCK), 0, st, nblk, work,
		     out_dev);
  return (int) hipGetLastError();
}

extern "C" int lbmi_k_moments_nblk(void) { return MOM_NBLK; }

extern "C" int lbmi_k_moments(const lbmi_kparam_t * kp, const double * f,
			      const char * status, double * work,
			      double * out_dev, void * stream) {
  hipStream_t st = (hipStream_t) stream;
  int n = kp->nlocal[0]*kp->strx;
  int nblk = (n + BLOCK - 1)/BLOCK;
  if (nblk > MOM_NBLK) nblk = MOM_NBLK;
  if (kp->nvel == 19) {
    hipLaunchKernelGGL((k_moments_partial<19>), dim3(nblk), dim3(BLOCK), 0, st,
		       *kp, f, status, work);
  }
  else if (kp->nvel == 27) {
    hipLaunchKernelGGL((k_moments_partial<27>), dim3(nblk), dim3(BLOCK), 0, st,
		       *kp, f, status, work);
  }
  else {
    return (int) hipErrorInvalidValue;
  }
  hipLaunchKernelGGL(k_moments_final, dim3(1), dim3(BLOCK), 0, st, nblk, work,
		     out_dev);
  return (int) hipGetLastError();
}

extern "C" int lbmi_k_model(int nvel, int8_t * cv, double * wv, double * na,
			    double * ma) {
  if (nvel == 19) return model_tables<19>(cv, wv, na, ma);
  if (nvel == 27) return model_tables<27>(cv, wv, na, ma);
  return -1;
}
