/*
 * lb_oracle.c -- TEST INFRASTRUCTURE ONLY (see lb_oracle.h).
 *
 * A CPU restatement of the reference algorithm for the lattice-Boltzmann
 * hot path: collision, halo swap, propagation and the conserved-moment
 * statistics. Written from the reference's behaviour; each function cites
 * the reference file:line it follows. Summation orders follow the reference
 * (modes summed over p = 0..nvel-1, populations over m = 0..nvel-1) and the
 * file is compiled with -ffp-contract=off so that no FMA contraction
 * changes the rounding relative to the reference's x86-64 gcc -O build.
 *
 * Parity status: PINNED by tests/golden (compiled-reference outputs) and by
 * the reference regression logs; see tests/test_oracle_golden.py.
 */

#include <float.h>
#include <math.h>
#include <stddef.h>
#include <string.h>

#include "lb_oracle.h"

enum {X = 0, Y = 1, Z = 2};
enum {NHYDRO = 10};                    /* 1 + 3 + 6, lb_data.h:140 */
enum {MAP_FLUID = 0};                  /* map.h:23 */

static const double cs2 = (1.0/3.0);   /* LB_CS2_DOUBLE, lb_data.h:149 */

/* Velocity sets and weights: values of lb_d3q19.h:26-38, lb_d3q27.h:28-43 */

static const int8_t cv19[19][3] = {
  { 0, 0, 0},
  { 1, 1, 0}, { 1, 0, 1}, { 1, 0, 0}, { 1, 0,-1}, { 1,-1, 0},
  { 0, 1, 1}, { 0, 1, 0}, { 0, 1,-1}, { 0, 0, 1}, { 0, 0,-1},
  { 0,-1, 1}, { 0,-1, 0}, { 0,-1,-1},
  {-1, 1, 0}, {-1, 0, 1}, {-1, 0, 0}, {-1, 0,-1}, {-1,-1, 0}};

static int weight19(const int8_t c[3]) {
  int c2 = c[X]*c[X] + c[Y]*c[Y] + c[Z]*c[Z];
  return (c2 == 0) ? 12 : ((c2 == 1) ? 2 : 1);          /* over 36 */
}

static int weight27(const int8_t c[3]) {
  int c2 = c[X]*c[X] + c[Y]*c[Y] + c[Z]*c[Z];
  return (c2 == 0) ? 64 : ((c2 == 1) ? 16 : ((c2 == 2) ? 4 : 1)); /* /216 */
}

/*
 * lbo_model_create
 *
 * cv, wv: lb_d3q19.h / lb_d3q27.h. ma: lb_d3q19.c:107-153,
 * lb_d3q27.c:150-195. na[m] = 1/sum_p wv[p] ma[m][p]^2 (lb_d3q19.c:69-77).
 * mi[p][m] = wv[p] na[m] ma[m][p] (model.c:381-387).
 */

int lbo_model_create(int nvel, lbo_model_t * model) {

  if (model == NULL) return -1;
  if (nvel != 19 && nvel != 27) return -1;

  memset(model, 0, sizeof(lbo_model_t));
  model->nvel = nvel;

  if (nvel == 19) {
    for (int p = 0; p < 19; p++) {
      for (int ia = 0; ia < 3; ia++) model->cv[p][ia] = cv19[p][ia];
      model->wv[p] = weight19(cv19[p])/36.0;
    }
  }
  else {
    /* D3Q27: p = 0 is rest; then x slowest, z fastest from (-1,-1,-1),
     * skipping the rest vector (lb_d3q27.h:28-34) */
    int p = 1;
    for (int ix = -1; ix <= 1; ix++) {
      for (int iy = -1; iy <= 1; iy++) {
	for (int iz = -1; iz <= 1; iz++) {
	  if (ix == 0 && iy == 0 && iz == 0) continue;
	  model->cv[p][X] = ix; model->cv[p][Y] = iy; model->cv[p][Z] = iz;
	  p += 1;
	}
      }
    }
    for (p = 0; p < 27; p++) model->wv[p] = weight27(model->cv[p])/216.0;
  }

  for (int p = 0; p < nvel; p++) {
    double cx = 1.0*model->cv[p][X];
    double cy = 1.0*model->cv[p][Y];
    double cz = 1.0*model->cv[p][Z];
    double (* ma)[LBO_NVEL_MAX] = model->ma;

    ma[0][p] = 1.0;
    ma[1][p] = cx;
    ma[2][p] = cy;
    ma[3][p] = cz;
    ma[4][p] = cx*cx - cs2;
    ma[5][p] = cx*cy;
    ma[6][p] = cx*cz;
    ma[7][p] = cy*cy - cs2;
    ma[8][p] = cy*cz;
    ma[9][p] = cz*cz - cs2;

    if (nvel == 19) {
      double c2   = cx*cx + cy*cy + cz*cz;
      double chi1 = (2.0*c2 - 3.0)*(3.0*cz*cz - c2);
      double chi2 = (2.0*c2 - 3.0)*(cy*cy - cx*cx);
      double chi3 = 3.0*c2*c2 - 6.0*c2 + 1;
      ma[10][p] = chi1;
      ma[11][p] = chi1*cx;
      ma[12][p] = chi1*cy;
      ma[13][p] = chi1*cz;
      ma[14][p] = chi2;
      ma[15][p] = chi2*cx;
      ma[16][p] = chi2*cy;
      ma[17][p] = chi2*cz;
      ma[18][p] = chi3;
    }
    else {
      double hx = cx*cx - cs2;
      double hy = cy*cy - cs2;
      double hz = cz*cz - cs2;
      ma[10][p] = 3.0*hx*cy;
      ma[11][p] = 3.0*hx*cz;
      ma[12][p] = 3.0*hy*cz;
      ma[13][p] = 3.0*hy*cx;
      ma[14][p] = 3.0*hz*cx;
      ma[15][p] = 3.0*hz*cy;
      ma[16][p] = cx*cy*cz;
      ma[17][p] = 9.0*hx*hy;
      ma[18][p] = 9.0*hy*hz;
      ma[19][p] = 9.0*hz*hx;
      ma[20][p] = 9.0*hx*cy*cz;
      ma[21][p] = 9.0*hy*cz*cx;
      ma[22][p] = 9.0*hz*cx*cy;
      ma[23][p] = 9.0*hx*hy*cz;
      ma[24][p] = 9.0*hy*hz*cx;
      ma[25][p] = 9.0*hz*hx*cy;
      ma[26][p] = 27.0*hx*hy*hz;
    }
  }

  for (int m = 0; m < nvel; m++) {
    double wip = 0.0;
    for (int p = 0; p < nvel; p++) {
      wip += model->wv[p]*model->ma[m][p]*model->ma[m][p];
    }
    model->na[m] = 1.0/wip;
  }

  for (int p = 0; p < nvel; p++) {
    for (int m = 0; m < nvel; m++) {
      model->mi[p][m] = model->wv[p]*model->na[m]*model->ma[m][p];
    }
  }

  return 0;
}

int lbo_nsite(const lbo_param_t * p) {
  int nh2 = 2*p->nhalo;
  return (p->nlocal[X] + nh2)*(p->nlocal[Y] + nh2)*(p->nlocal[Z] + nh2);
}

static void strides(const lbo_param_t * p, int nall[3], ptrdiff_t str[3]) {
  for (int ia = 0; ia < 3; ia++) nall[ia] = p->nlocal[ia] + 2*p->nhalo;
  str[Z] = 1;
  str[Y] = nall[Z];
  str[X] = (ptrdiff_t) nall[Y]*nall[Z];       /* coords.c:211-215 */
}

/*
 * Relaxation rates for the single-fluid kernel: collision.c:1287-1300
 * (shear), :1339-1373 (bulk), :1443-1538 (ghosts; scheme-driven).
 */

static int relaxation_rates_eta(const lbo_param_t * p, double eta,
				double eta_bulk, double * rtau_shear,
				double * rtau_bulk,
				double rtau_ghost[LBO_NVEL_MAX]);

static int relaxation_rates(const lbo_param_t * p, double * rtau_shear,
			    double * rtau_bulk, double rtau_ghost[LBO_NVEL_MAX]) {
  return relaxation_rates_eta(p, p->eta_shear, p->eta_bulk, rtau_shear,
			      rtau_bulk, rtau_ghost);
}

/* lb_relaxation_time_shear_v, _bulk_v, _ghosts_v (collision.c:1287-1538) for
 * a given (possibly local) shear viscosity eta and bulk viscosity eta_bulk */

static int relaxation_rates_eta(const lbo_param_t * p, double eta,
				double eta_bulk, double * rtau_shear,
				double * rtau_bulk,
				double rtau_ghost[LBO_NVEL_MAX]) {

  double rtau = 1.0/(0.5 + eta/(p->rho0*cs2));

  *rtau_shear = rtau;

  for (int m = 0; m < LBO_NVEL_MAX; m++) rtau_ghost[m] = 0.0;

  switch (p->scheme) {
  case LBO_M10:
    *rtau_bulk = 1.0/(0.5 + eta_bulk/(p->rho0*cs2));
    for (int m = NHYDRO; m < p->nvel; m++) rtau_ghost[m] = 1.0;
    break;
  case LBO_BGK:
    *rtau_bulk = rtau;
    for (int m = NHYDRO; m < p->nvel; m++) rtau_ghost[m] = rtau;
    break;
  case LBO_TRT:
    {
      /* Only defined for nvel = 19 here (the reference leaves the d3q27
       * ghost rates uninitialised: collision.c:1487-1534). */
      double tau = eta/(p->rho0*cs2);
      double rtau_odd = 0.5 + 2.0*tau/(tau + 3.0/8.0);
      if (rtau_odd > 2.0) rtau_odd = 2.0;
      if (p->nvel != 19) return -1;
      *rtau_bulk = 1.0/(0.5 + eta_bulk/(p->rho0*cs2));
      rtau_ghost[10] = rtau; rtau_ghost[14] = rtau; rtau_ghost[18] = rtau;
      rtau_ghost[11] = rtau_odd; rtau_ghost[12] = rtau_odd;
      rtau_ghost[13] = rtau_odd; rtau_ghost[15] = rtau_odd;
      rtau_ghost[16] = rtau_odd; rtau_ghost[17] = rtau_odd;
    }
    break;
  default:
    return -1;
  }

  return 0;
}

/*
 * lbo_collide
 *
 * Single-fluid collision for all interior sites: lb_collision_mrt1_site,
 * collision.c:259-599 (fluctuations off, no free-energy stress, constant
 * viscosity). Non-fluid sites are left untouched (collision.c:299-304,
 * 581-595). force may be NULL (zero field); status may be NULL (all fluid).
 * rho, u (SoA, addr_rank1(nsite,3,index,ia) = nsite*ia + index) may be NULL.
 *
 * The reference also "collides" the y/z halo sites in the x-interior range
 * (kernel.c:194-209); those results are overwritten by lb_halo and are not
 * reproduced here.
 */

static void symm_stress(double a, double b, double kappa, double phi,
			const double g[3], double delsq, double s[3][3]);

/* fe != NULL: fe->use_stress_relaxation (collision.c:413-429) with the
 * symmetric free energy: its stress (fe_symm_str_v, symmetric.c:371-420) is
 * added to the equilibrium stress. fe = {a, b, kappa}; phi, grad, delsq as
 * field_grad_compute left them. */

typedef struct {
  uint32_t * state;        /* noise->state: [ia*nsite + index], ia = 0 .. 3 */
  double kt;
  int ghosts_on;           /* lb->param->isghost == LB_GHOST_ON */
} lbo_noise_t;

static int collide_impl(const lbo_param_t * p, double * f,
			const double * force, const char * status,
			const double * fe, const double * phi,
			const double * grad, const double * delsq,
			const double * eta_site, const lbo_noise_t * noise,
			double * rho_out, double * u_out);

int lbo_collide(const lbo_param_t * p, double * f, const double * force,
		const char * status, double * rho_out, double * u_out) {
  return collide_impl(p, f, force, status, NULL, NULL, NULL, NULL, NULL, NULL,
		      rho_out, u_out);
}

/* With isothermal fluctuations (noise->on[NOISE_RHO], collision.c:476-518):
 * state is the reference's per-site generator state, advanced here as
 * lb_collide advances it. eta may be NULL (no viscosity model). */

int lbo_collide_noise(const lbo_param_t * p, double * f, const double * force,
		      const char * status, const double * eta,
		      uint32_t * state, double kt, int ghosts_on,
		      double * rho_out, double * u_out) {
  lbo_noise_t noise = {state, kt, ghosts_on};
  if (p->nvel != 19) return -1;      /* NNOISE_MAX = 10 < 17 ghosts, noise.h:18 */
  return collide_impl(p, f, force, status, NULL, NULL, NULL, NULL, eta, &noise,
		      rho_out, u_out);
}

/* noise_uniform (noise.c:467-487): Marsaglia's combination of a congruential
 * generator, a 3-shift register and two multiply-with-carry generators. */

static uint32_t noise_uniform(uint32_t state[4]) {
  uint32_t b;
  state[0] = 69069u*state[0] + 1234567u;
  b = state[1] ^ (state[1] << 17);
  b ^= (b >> 13);
  state[1] = b ^ (b << 5);
  state[2] = 36969u*(state[2] & 0xffffu) + (state[2] >> 16);
  state[3] = 18000u*(state[3] & 0xffffu) + (state[3] >> 16);
  b = (state[2] << 16) + state[3];
  return state[1] + (state[0] ^ b);
}

/* noise_reap_n (noise.c:397-424): one draw, its two leading bits dropped,
 * three bits per number into the table of noise.c:72-79 */

static void noise_reap_n(uint32_t * state, ptrdiff_t nsite, ptrdiff_t index,
			 int nmax, double * reap) {
  const double a = sqrt(2.0 + sqrt(2.0));
  const double b = sqrt(2.0 - sqrt(2.0));
  const double rtable[8] = {-a, -b, 0.0, 0.0, 0.0, 0.0, +b, +a};
  uint32_t s[4], iuniform;
  for (int ia = 0; ia < 4; ia++) s[ia] = state[nsite*ia + index];
  iuniform = noise_uniform(s);
  for (int ia = 0; ia < 4; ia++) state[nsite*ia + index] = s[ia];
  iuniform >>= 2;
  for (int ia = 0; ia < nmax; ia++) {
    reap[ia] = rtable[iuniform & 7];
    iuniform >>= 3;
  }
}

/* The random stress and ghost-mode parts of one site. mrt2 = 0:
 * lb_fluctuations_var_eta, _var_bulk, _stress, _var_ghost, _ghosts as
 * lb_collision_mrt1_site calls them (collision.c:491-516, 1753-1918);
 * mrt2 = 1: lb_collision_fluctuations of the two-distribution collision
 * (:1663-1745), the same numbers with the ghost variance written as
 * sqrt(rna*rcs2*kt). rna[p] = 1/na[p] (model.c:377-379). */

static void site_fluctuations(const lbo_noise_t * noise, const lbo_model_t * model,
			      int nvel, ptrdiff_t nsite, ptrdiff_t index,
			      double rtau, double rtau_bulk,
			      const double * rtau_ghost, int mrt2,
			      double shat[3][3], double * ghat) {
  const double rcs2 = 3.0;
  double random[10];
  double kt = noise->kt*rcs2;
  double tr;
  double tau = 1.0/rtau, tau_b = 1.0/rtau_bulk;
  double var = sqrt(kt)*sqrt(1.0/9.0)*sqrt((tau + tau - 1.0)/(tau*tau));
  double var_bulk = sqrt(kt)*sqrt(2.0/9.0)
    *sqrt((tau_b + tau_b - 1.0)/(tau_b*tau_b));

  noise_reap_n(noise->state, nsite, index, 6, random);
  shat[X][X] = random[0]; shat[X][Y] = random[1]; shat[X][Z] = random[2];
  shat[Y][X] = shat[X][Y]; shat[Y][Y] = random[3]; shat[Y][Z] = random[4];
  shat[Z][X] = shat[X][Z]; shat[Z][Y] = shat[Y][Z]; shat[Z][Z] = random[5];
  tr = (1.0/3)*(shat[X][X] + shat[Y][Y] + (3 - 2.0)*shat[Z][Z]);
  shat[X][X] -= tr; shat[Y][Y] -= tr; shat[Z][Z] -= tr;
  shat[X][X] *= var*sqrt(2.0); shat[X][Y] *= var; shat[X][Z] *= var;
  shat[Y][X] *= var; shat[Y][Y] *= var*sqrt(2.0); shat[Y][Z] *= var;
  shat[Z][X] *= var; shat[Z][Y] *= var; shat[Z][Z] *= var*sqrt(2.0);
  tr *= var_bulk;
  shat[X][X] += tr; shat[Y][Y] += tr; shat[Z][Z] += tr;

  if (noise->ghosts_on) {
    noise_reap_n(noise->state, nsite, index, nvel - NHYDRO, random);
    for (int m = NHYDRO; m < nvel; m++) {
      double tau_g = 1.0/rtau_ghost[m];
      double rna = 1.0/model->na[m];
      double varg = mrt2 ? sqrt(rna*rcs2*noise->kt) : sqrt(kt*rna);
      varg = varg*sqrt((tau_g + tau_g - 1.0)/(tau_g*tau_g));
      ghat[m] = varg*random[m - NHYDRO];
    }
  }
}

/* With a viscosity model (visc != NULL in lb_collide): the local shear
 * viscosity comes from hydro->eta and the bulk viscosity keeps the
 * Newtonian ratio, (eta_bulk/eta_shear) eta (collision.c:386-404). */

int lbo_collide_visc(const lbo_param_t * p, double * f, const double * force,
		     const char * status, const double * eta,
		     double * rho_out, double * u_out) {
  return collide_impl(p, f, force, status, NULL, NULL, NULL, NULL, eta, NULL,
		      rho_out, u_out);
}

int lbo_collide_fe(const lbo_param_t * p, double * f, const double * force,
		   const char * status, double a, double b, double kappa,
		   const double * phi, const double * grad,
		   const double * delsq, double * rho_out, double * u_out) {
  const double fe[3] = {a, b, kappa};
  return collide_impl(p, f, force, status, fe, phi, grad, delsq, NULL, NULL,
		      rho_out, u_out);
}

static int collide_impl(const lbo_param_t * p, double * f,
			const double * force, const char * status,
			const double * fe, const double * phi,
			const double * grad, const double * delsq,
			const double * eta_site, const lbo_noise_t * noise,
			double * rho_out, double * u_out) {

  int nall[3];
  ptrdiff_t str[3];
  ptrdiff_t nsite;
  int nvel = p->nvel;
  double rtau, rtau_bulk, rtau_ghost[LBO_NVEL_MAX];
  lbo_model_t model;
  const double rdim = 1.0/3.0;

  if (lbo_model_create(nvel, &model) != 0) return -1;
  if (relaxation_rates(p, &rtau, &rtau_bulk, rtau_ghost) != 0) return -1;

  /* REFERENCE QUIRK, reproduced on purpose. For D3Q19 the reference does
   * not use lb->param->ma in the collision but the unrolled literal
   * d3q19_f2mode_chunk, and that routine multiplies f[4] by 0 instead of
   * ma[13][4] = chi1*cz = -1 when forming mode 13 (collision.c:2300,
   * "mode[13] += fchunk[4]*c0"). All other 721 literal coefficients of
   * f2mode/mode2f equal ma and mi = wv*na*ma. The effect is invisible with
   * M10 (ghost modes are discarded) but changes BGK and TRT results, which
   * keep (1 - rtau_ghost) of mode 13. Parity means matching the reference,
   * so the collision transform (only) carries the same coefficient. */
  if (nvel == 19) model.ma[13][4] = 0.0;

  strides(p, nall, str);
  nsite = (ptrdiff_t) nall[X]*nall[Y]*nall[Z];

  #pragma omp parallel for collapse(2) schedule(static)
  for (int ic = 1; ic <= p->nlocal[X]; ic++) {
    for (int jc = 1; jc <= p->nlocal[Y]; jc++) {
      for (int kc = 1; kc <= p->nlocal[Z]; kc++) {

	ptrdiff_t index = str[X]*(p->nhalo + ic - 1)
	  + str[Y]*(p->nhalo + jc - 1) + (p->nhalo + kc - 1);
	double mode[LBO_NVEL_MAX];
	double fl[LBO_NVEL_MAX];
	double frc[3], u[3];
	double s[3][3], seq[3][3];
	double rho, rrho, tr_s, tr_seq;
	double srtau = rtau, srtau_bulk = rtau_bulk;   /* this site's rates */
	double srtau_ghost[LBO_NVEL_MAX];
	int m;

	for (m = 0; m < LBO_NVEL_MAX; m++) srtau_ghost[m] = rtau_ghost[m];

	if (status && status[index] != MAP_FLUID) continue;

	for (int q = 0; q < nvel; q++) fl[q] = f[nsite*q + index];

	for (int ia = 0; ia < 3; ia++) {
	  frc[ia] = p->fbody[ia];
	  if (force) frc[ia] += force[nsite*ia + index];
	}

	/* f -> modes (collision.c:338-349) */
	for (m = 0; m < nvel; m++) {
	  double sum = 0.0;
	  for (int q = 0; q < nvel; q++) sum += fl[q]*model.ma[m][q];
	  mode[m] = sum;
	}

	rho = mode[0];
	m = 0;
	for (int ia = 0; ia < 3; ia++) {
	  for (int ib = ia; ib < 3; ib++) {
	    s[ia][ib] = mode[4 + m];
	    s[ib][ia] = mode[4 + m];
	    m++;
	  }
	}

	/* velocity with half-force (collision.c:376-382) */
	rrho = 1.0/rho;
	for (int ia = 0; ia < 3; ia++) u[ia] = rrho*(mode[1+ia] + 0.5*frc[ia]);

	if (eta_site) {
	  /* local relaxation times (collision.c:386-404) */
	  double eta = eta_site[index];
	  relaxation_rates_eta(p, eta, (p->eta_bulk/p->eta_shear)*eta,
			       &srtau, &srtau_bulk, srtau_ghost);
	}

	/* stress relaxation (collision.c:408-474) */
	tr_s = 0.0; tr_seq = 0.0;
	for (int ia = 0; ia < 3; ia++) {
	  for (int ib = 0; ib < 3; ib++) seq[ia][ib] = rho*u[ia]*u[ib];
	}
	if (fe) {
	  double sth[3][3], gr[3];
	  for (int ia = 0; ia < 3; ia++) gr[ia] = grad[nsite*ia + index];
	  symm_stress(fe[0], fe[1], fe[2], phi[index], gr, delsq[index], sth);
	  for (int ia = 0; ia < 3; ia++) {
	    for (int ib = 0; ib < 3; ib++) seq[ia][ib] += sth[ia][ib];
	  }
	}
	for (int ia = 0; ia < 3; ia++) {
	  tr_s   += s[ia][ia];
	  tr_seq += seq[ia][ia];
	}
	for (int ia = 0; ia < 3; ia++) {
	  s[ia][ia]   -= rdim*tr_s;
	  seq[ia][ia] -= rdim*tr_seq;
	}
	tr_s = tr_s - srtau_bulk*(tr_s - tr_seq);

	for (int ia = 0; ia < 3; ia++) {
	  for (int ib = 0; ib < 3; ib++) {
	    double dab = (ia == ib);
	    s[ia][ib] -= srtau*(s[ia][ib] - seq[ia][ib]);
	    s[ia][ib] += dab*rdim*tr_s;
	    s[ia][ib] += (2.0 - srtau)*(u[ia]*frc[ib] + frc[ia]*u[ib]);
	  }
	}

	/* fluctuations (collision.c:476-518) */
	double shat[3][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
	double ghat[LBO_NVEL_MAX];
	for (m = 0; m < LBO_NVEL_MAX; m++) ghat[m] = 0.0;
	if (noise) {
	  site_fluctuations(noise, &model, nvel, nsite, index, srtau, srtau_bulk,
			    srtau_ghost, 0, shat, ghat);
	}

	/* post-collision modes (collision.c:523-544) */
	for (int ia = 0; ia < 3; ia++) mode[1+ia] += frc[ia];
	m = 0;
	for (int ia = 0; ia < 3; ia++) {
	  for (int ib = ia; ib < 3; ib++) {
	    mode[4 + m] = s[ia][ib] + shat[ia][ib];
	    m++;
	  }
	}
	for (m = NHYDRO; m < nvel; m++) {
	  mode[m] = mode[m] - srtau_ghost[m]*(mode[m] - 0.0) + ghat[m];
	}

	/* modes -> f (collision.c:548-559) */
	for (int q = 0; q < nvel; q++) {
	  double sum = 0.0;
	  for (m = 0; m < nvel; m++) sum += model.mi[q][m]*mode[m];
	  f[nsite*q + index] = sum;
	}

	if (rho_out) rho_out[index] = rho;
	if (u_out) {
	  for (int ia = 0; ia < 3; ia++) u_out[nsite*ia + index] = u[ia];
	}
      }
    }
  }

  return 0;
}

/*
 * lbo_phi_from_g, lbo_collide_binary
 *
 * The two-distribution ("symmetric_lb") step. f2 holds both distributions,
 * f2[(n*nvel + p)*nsite + index], n = 0 (LB_RHO) and n = 1 (LB_PHI).
 *
 * lbo_phi_from_g: phi_lb_to_field (phi_lb_coupler.c:39-112): phi = sum_p g_p
 * in p order at the interior sites.
 *
 * lbo_collide_binary: lb_collision_mrt2_site (collision.c:720-1027) without
 * noise. The density distribution relaxes as in lbo_collide with the
 * thermodynamic stress added to the equilibrium stress,
 *   seq_ab = rho u_a u_b + P_ab,  P_ab of fe_symm_str_v (symmetric.c:371-420,
 *   see symm_stress below), at EVERY interior site (no status test), and
 *   writes hydro->u only. The order-parameter distribution is re-projected
 *   (:955-1024): jphi_a = sum_{p>=1} c_pa g_p relaxed towards phi u_a at rate
 *   rtau2 = 2/(1 + 2 M) (:1968), sphi_ab = phi u_a u_b + mu delta_ab,
 *   g_p = w_p (3 jphi.c_p + 4.5 sphi:(c_p c_p - delta/3)) + phi delta_p0,
 *   mu = a phi + b phi^3 - kappa delsq (symmetric.c:303-316).
 * For D3Q19 the f -> mode transform carries the reference's literal
 * coefficient (see lbo_collide).
 */

int lbo_phi_from_g(const lbo_param_t * p, const double * f2, double * phi) {

  int nall[3];
  ptrdiff_t str[3];
  ptrdiff_t nsite;
  const double * g;

  strides(p, nall, str);
  nsite = (ptrdiff_t) nall[X]*nall[Y]*nall[Z];
  g = f2 + (ptrdiff_t) p->nvel*nsite;

  for (int ic = 1; ic <= p->nlocal[X]; ic++) {
    for (int jc = 1; jc <= p->nlocal[Y]; jc++) {
      for (int kc = 1; kc <= p->nlocal[Z]; kc++) {
	ptrdiff_t index = str[X]*(p->nhalo + ic - 1)
	  + str[Y]*(p->nhalo + jc - 1) + (p->nhalo + kc - 1);
	double sum = 0.0;
	for (int q = 0; q < p->nvel; q++) sum += g[nsite*q + index];
	phi[index] = sum;
      }
    }
  }

  return 0;
}

static void symm_stress(double a, double b, double kappa, double phi,
			const double g[3], double delsq, double s[3][3]);

static int collide_binary_impl(const lbo_param_t * p, double * f2,
			       const double * force, double a, double b, double kappa,
			       double mobility, const double * phi,
			       const double * grad, const double * delsq,
			       const lbo_noise_t * noise, double * u_out);

int lbo_collide_binary(const lbo_param_t * p, double * f2,
		       const double * force, double a, double b, double kappa,
		       double mobility, const double * phi,
		       const double * grad, const double * delsq,
		       double * u_out) {
  return collide_binary_impl(p, f2, force, a, b, kappa, mobility, phi, grad,
			     delsq, NULL, u_out);
}

/* ... with isothermal fluctuations: lb_collision_fluctuations at every site
 * (collision.c:884-900; no status test in lb_collision_mrt2_site) */

int lbo_collide_binary_noise(const lbo_param_t * p, double * f2,
			     const double * force, double a, double b,
			     double kappa, double mobility, const double * phi,
			     const double * grad, const double * delsq,
			     uint32_t * state, double kt, int ghosts_on,
			     double * u_out) {
  lbo_noise_t noise = {state, kt, ghosts_on};
  if (p->nvel != 19) return -1;
  return collide_binary_impl(p, f2, force, a, b, kappa, mobility, phi, grad,
			     delsq, &noise, u_out);
}

static int collide_binary_impl(const lbo_param_t * p, double * f2,
			       const double * force, double a, double b, double kappa,
			       double mobility, const double * phi,
			       const double * grad, const double * delsq,
			       const lbo_noise_t * noise, double * u_out) {

  int nall[3];
  ptrdiff_t str[3];
  ptrdiff_t nsite;
  int nvel = p->nvel;
  double rtau, rtau_bulk, rtau_ghost[LBO_NVEL_MAX];
  double rtau2 = 2.0/(1.0 + 2.0*mobility);
  lbo_model_t model, mphi;
  const double rdim = 1.0/3.0;
  double * g;

  if (lbo_model_create(nvel, &model) != 0) return -1;
  if (lbo_model_create(nvel, &mphi) != 0) return -1;
  if (relaxation_rates(p, &rtau, &rtau_bulk, rtau_ghost) != 0) return -1;
  if (nvel == 19) model.ma[13][4] = 0.0;      /* see lbo_collide */

  strides(p, nall, str);
  nsite = (ptrdiff_t) nall[X]*nall[Y]*nall[Z];
  g = f2 + (ptrdiff_t) nvel*nsite;

  for (int ic = 1; ic <= p->nlocal[X]; ic++) {
    for (int jc = 1; jc <= p->nlocal[Y]; jc++) {
      for (int kc = 1; kc <= p->nlocal[Z]; kc++) {

	ptrdiff_t index = str[X]*(p->nhalo + ic - 1)
	  + str[Y]*(p->nhalo + jc - 1) + (p->nhalo + kc - 1);
	double mode[LBO_NVEL_MAX];
	double fl[LBO_NVEL_MAX];
	double frc[3], u[3], gr[3];
	double s[3][3], seq[3][3], sth[3][3], sphi[3][3];
	double jphi[3] = {0.0, 0.0, 0.0};
	double rho, rrho, tr_s, tr_seq, ph, mu;
	int m;

	for (int q = 0; q < nvel; q++) fl[q] = f2[nsite*q + index];
	for (int ia = 0; ia < 3; ia++) {
	  frc[ia] = p->fbody[ia];
	  if (force) frc[ia] += force[nsite*ia + index];
	}
	for (m = 0; m < nvel; m++) {
	  double sum = 0.0;
	  for (int q = 0; q < nvel; q++) sum += fl[q]*model.ma[m][q];
	  mode[m] = sum;
	}
	rho = mode[0];
	m = 0;
	for (int ia = 0; ia < 3; ia++) {
	  for (int ib = ia; ib < 3; ib++) {
	    s[ia][ib] = mode[4 + m];
	    s[ib][ia] = mode[4 + m];
	    m++;
	  }
	}
	rrho = 1.0/rho;
	for (int ia = 0; ia < 3; ia++) u[ia] = rrho*(mode[1+ia] + 0.5*frc[ia]);

	ph = phi[index];
	for (int ia = 0; ia < 3; ia++) gr[ia] = grad[nsite*ia + index];
	symm_stress(a, b, kappa, ph, gr, delsq[index], sth);

	tr_s = 0.0; tr_seq = 0.0;
	for (int ia = 0; ia < 3; ia++) {
	  for (int ib = 0; ib < 3; ib++) {
	    seq[ia][ib] = rho*u[ia]*u[ib] + sth[ia][ib];
	  }
	  tr_s   += s[ia][ia];
	  tr_seq += seq[ia][ia];
	}
	for (int ia = 0; ia < 3; ia++) {
	  s[ia][ia]   -= rdim*tr_s;
	  seq[ia][ia] -= rdim*tr_seq;
	}
	tr_s = tr_s - rtau_bulk*(tr_s - tr_seq);
	for (int ia = 0; ia < 3; ia++) {
	  for (int ib = 0; ib < 3; ib++) {
	    double dab = (ia == ib);
	    s[ia][ib] -= rtau*(s[ia][ib] - seq[ia][ib]);
	    s[ia][ib] += dab*rdim*tr_s;
	    s[ia][ib] += (2.0 - rtau)*(u[ia]*frc[ib] + frc[ia]*u[ib]);
	  }
	}
	double shat[3][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
	double ghat[LBO_NVEL_MAX];
	for (m = 0; m < LBO_NVEL_MAX; m++) ghat[m] = 0.0;
	if (noise) {
	  site_fluctuations(noise, &model, nvel, nsite, index, rtau, rtau_bulk,
			    rtau_ghost, 1, shat, ghat);
	}
	for (int ia = 0; ia < 3; ia++) mode[1+ia] += frc[ia];
	m = 0;
	for (int ia = 0; ia < 3; ia++) {
	  for (int ib = ia; ib < 3; ib++) {
	    mode[4 + m] = s[ia][ib] + shat[ia][ib];
	    m++;
	  }
	}
	for (m = NHYDRO; m < nvel; m++) {
	  mode[m] = mode[m] - rtau_ghost[m]*(mode[m] - 0.0) + ghat[m];
	}
	for (int q = 0; q < nvel; q++) {
	  double sum = 0.0;
	  for (m = 0; m < nvel; m++) sum += model.mi[q][m]*mode[m];
	  f2[nsite*q + index] = sum;
	}
	if (u_out) {
	  for (int ia = 0; ia < 3; ia++) u_out[nsite*ia + index] = u[ia];
	}

	/* order-parameter distribution (collision.c:955-1024) */
	mu = a*ph + b*ph*ph*ph - kappa*delsq[index];
	for (int q = 1; q < nvel; q++) {
	  for (int ia = 0; ia < 3; ia++) {
	    jphi[ia] += mphi.cv[q][ia]*g[nsite*q + index];
	  }
	}
	for (int ia = 0; ia < 3; ia++) {
	  for (int ib = 0; ib < 3; ib++) {
	    sphi[ia][ib] = ph*u[ia]*u[ib] + mu*(ia == ib);
	  }
	  jphi[ia] = jphi[ia] - rtau2*(jphi[ia] - ph*u[ia]);
	}
	for (int q = 0; q < nvel; q++) {
	  double jdotc = 0.0;
	  double sphidotq = 0.0;
	  for (int ia = 0; ia < 3; ia++) {
	    jdotc += jphi[ia]*mphi.cv[q][ia];
	    for (int ib = 0; ib < 3; ib++) {
	      sphidotq += sphi[ia][ib]*(mphi.cv[q][ia]*mphi.cv[q][ib]
					- cs2*(ia == ib));
	    }
	  }
	  g[nsite*q + index] = mphi.wv[q]*(jdotc*3.0 + sphidotq*4.5)
	    + ph*(q == 0);
	}
      }
    }
  }

  return 0;
}

/*
 * lbo_wall_map, lbo_wall_links, lbo_wall_bbl
 *
 * Flat walls and bounce-back on links (row f4).
 *
 * lbo_wall_map: wall_init_map (wall.c:1219-1268) on one rank: every site
 * (halo included) whose coordinate in a wall direction is 0 or ntotal+1
 * becomes MAP_BOUNDARY (status: nsite chars, other entries untouched).
 *
 * lbo_wall_links: wall_init_boundaries (wall.c:381-470) + wall_init_uw
 * (:864-890): for every interior MAP_FLUID site i, in (ic, jc, kc) order, and
 * p = 1..nvel-1, a link if i + c_p is MAP_BOUNDARY; linku = 0 (WALL_UZERO),
 * or with walls in exactly ONE direction iw: 2 (WALL_UWBOT) if c_p[iw] = -1,
 * 1 (WALL_UWTOP) if c_p[iw] = +1 (wall.c:32-35: WALL_UZERO, WALL_UWTOP, WALL_UWBOT).
 * Returns the number of links; fills at most maxlink entries.
 *
 * lbo_wall_bbl: wall_bbl_kernel (wall.c:996-1107), no colloids, one
 * distribution: f[j, nvel - p] = f[i, p] - 2 rcs2 w_p rho0 c_p.u_w, and the
 * momentum transfer (2 f[i,p] - 2 rcs2 w_p rho0 c_p.u_w - 2 w_p) c_p is
 * ADDED to fnet.
 */

#define MAP_BOUNDARY_ 1          /* map.h: MAP_FLUID = 0, MAP_BOUNDARY = 1 */

int lbo_wall_map(const lbo_param_t * p, const int isboundary[3],
		 char * status) {
  int nall[3];
  ptrdiff_t str[3];
  int h = p->nhalo;

  strides(p, nall, str);
  for (int ic = 1 - h; ic <= p->nlocal[X] + h; ic++) {
    for (int jc = 1 - h; jc <= p->nlocal[Y] + h; jc++) {
      for (int kc = 1 - h; kc <= p->nlocal[Z] + h; kc++) {
	ptrdiff_t index = str[X]*(h + ic - 1) + str[Y]*(h + jc - 1) + (h + kc - 1);
	if (isboundary[Z] && (kc == 0 || kc == p->nlocal[Z] + 1)) status[index] = MAP_BOUNDARY_;
	if (isboundary[Y] && (jc == 0 || jc == p->nlocal[Y] + 1)) status[index] = MAP_BOUNDARY_;
	if (isboundary[X] && (ic == 0 || ic == p->nlocal[X] + 1)) status[index] = MAP_BOUNDARY_;
      }
    }
  }
  return 0;
}

int lbo_wall_links(const lbo_param_t * p, const char * status,
		   const int isboundary[3], int maxlink, int * linki,
		   int * linkj, int * linkp, int * linku) {
  int nall[3];
  ptrdiff_t str[3];
  int h = p->nhalo;
  int nlink = 0;
  int nwall = isboundary[X] + isboundary[Y] + isboundary[Z];
  int iw = -1;
  lbo_model_t model;

  if (lbo_model_create(p->nvel, &model) != 0) return -1;
  strides(p, nall, str);
  if (nwall == 1) {
    if (isboundary[X]) iw = X;
    if (isboundary[Y]) iw = Y;
    if (isboundary[Z]) iw = Z;
  }

  for (int ic = 1; ic <= p->nlocal[X]; ic++) {
    for (int jc = 1; jc <= p->nlocal[Y]; jc++) {
      for (int kc = 1; kc <= p->nlocal[Z]; kc++) {
	ptrdiff_t i = str[X]*(h + ic - 1) + str[Y]*(h + jc - 1) + (h + kc - 1);
	if (status[i] != MAP_FLUID) continue;
	for (int q = 1; q < p->nvel; q++) {
	  ptrdiff_t j = i + str[X]*model.cv[q][X] + str[Y]*model.cv[q][Y]
	    + model.cv[q][Z];
	  if (status[j] != MAP_BOUNDARY_) continue;
	  if (nlink < maxlink) {
	    linki[nlink] = (int) i;
	    linkj[nlink] = (int) j;
	    linkp[nlink] = q;
	    linku[nlink] = 0;
	    if (iw >= 0 && model.cv[q][iw] == -1) linku[nlink] = 2;
	    if (iw >= 0 && model.cv[q][iw] == +1) linku[nlink] = 1;
	  }
	  nlink += 1;
	}
      }
    }
  }
  return nlink;
}

#define MAP_COLLOID_ 2           /* map.h:23 */

/* status (may be NULL): a fluid-side site that a colloid covers (MAP_COLLOID)
 * is not bounced; it only enters the accounting (wall.c:1048-1061, 1148-1161) */

static int wall_colloid_link(const lbo_model_t * model, int nvel,
			     ptrdiff_t nsite, const double * f,
			     const char * status, int i, int j, int ij,
			     double fnet[3]) {
  if (status == NULL || status[i] != MAP_COLLOID_) return 0;
  {
    double fp = f[nsite*ij + i] + f[nsite*(nvel - ij) + j];
    fnet[X] += (fp - 2.0*model->wv[ij])*model->cv[ij][X];
    fnet[Y] += (fp - 2.0*model->wv[ij])*model->cv[ij][Y];
    fnet[Z] += (fp - 2.0*model->wv[ij])*model->cv[ij][Z];
  }
  return 1;
}

int lbo_wall_bbl(const lbo_param_t * p, double * f, int nlink,
		 const int * linki, const int * linkj, const int * linkp,
		 const int * linku, const double ubot[3],
		 const double utop[3], double fnet[3], const char * status) {
  int nall[3];
  ptrdiff_t str[3];
  ptrdiff_t nsite;
  lbo_model_t model;
  const double rcs2 = 3.0;
  double uw[3][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};

  if (lbo_model_create(p->nvel, &model) != 0) return -1;
  strides(p, nall, str);
  nsite = (ptrdiff_t) nall[X]*nall[Y]*nall[Z];
  for (int ia = 0; ia < 3; ia++) {
    uw[1][ia] = utop[ia];
    uw[2][ia] = ubot[ia];
  }

  for (int n = 0; n < nlink; n++) {
    int ij = linkp[n];
    int ji = p->nvel - ij;
    int ia = linku[n];
    double cdotu = model.cv[ij][X]*uw[ia][X] + model.cv[ij][Y]*uw[ia][Y]
      + model.cv[ij][Z]*uw[ia][Z];
    double fp = f[nsite*ij + linki[n]];
    double force = 2.0*fp - 2.0*rcs2*model.wv[ij]*p->rho0*cdotu;
    if (wall_colloid_link(&model, p->nvel, nsite, f, status, linki[n],
			  linkj[n], ij, fnet)) continue;
    fnet[X] += (force - 2.0*model.wv[ij])*model.cv[ij][X];
    fnet[Y] += (force - 2.0*model.wv[ij])*model.cv[ij][Y];
    fnet[Z] += (force - 2.0*model.wv[ij])*model.cv[ij][Z];
    fp = fp - 2.0*rcs2*model.wv[ij]*p->rho0*cdotu;
    f[nsite*ji + linkj[n]] = fp;
  }
  return 0;
}

/*
 * lbo_wall_slip_table, lbo_wall_slip_links, lbo_wall_bbl_slip
 *
 * Flat walls with partial slip (row f4).
 *
 * lbo_wall_slip_table: wall_slip (wall.c:285-316): the 19-entry table of slip
 * fractions, index = wall_slip_enum_t (wall.h:25-41): 0 no slip, 1..6 the
 * faces XBOT XTOP YBOT YTOP ZBOT ZTOP, 7..18 the edges (mean of the two faces).
 * Returns "active" (any fraction non-zero).
 *
 * lbo_wall_slip_links: wall_init_boundaries_slip (wall.c:489-593) with
 * wall_link_normal (:606-642), wall_link_slip_direction (:658-693) and
 * wall_link_slip (:707-757): for every link (i, p) the wall normal wn at the
 * crossing (from the status of the three axis neighbours of i along c_p), the
 * tangent wt = c_p - (c_p.wn/wn.wn) wn, the partner fluid site k = i + wt, the
 * partner direction q with c_q = -2 wn - c_p, and the table index of the slip
 * fraction (faces and edges; corners and purely normal links: no slip, and
 * then k = i, q = p).
 *
 * lbo_wall_bbl_slip: wall_bbl_slip_kernel (wall.c:1118-1205), no colloids:
 * f[j, nvel-p] = (1-s) f[i,p] + s f[k,q]; the momentum keeps the reference's
 * integer arithmetic for the normal factor w = -(c_p + c_q)/2. LB_RHO only.
 */

int lbo_wall_slip_table(const double sbot[3], const double stop[3],
			double s[19]) {
  const double face[3][2] = {{sbot[X], stop[X]}, {sbot[Y], stop[Y]},
			     {sbot[Z], stop[Z]}};
  s[0] = 0.0;
  for (int ia = 0; ia < 3; ia++) {
    s[1 + 2*ia] = face[ia][0];
    s[2 + 2*ia] = face[ia][1];
  }
  /* edges: XB_YB XB_YT XB_ZB XB_ZT XT_YB XT_YT XT_ZB XT_ZT YB_ZB YB_ZT YT_ZB
   * YT_ZT (B = 0, T = 1) */
  for (int ta = 0; ta < 2; ta++) {
    for (int tb = 0; tb < 2; tb++) {
      s[7 + 4*ta + tb]  = 0.5*(face[X][ta] + face[Y][tb]);
      s[9 + 4*ta + tb]  = 0.5*(face[X][ta] + face[Z][tb]);
      s[15 + 2*ta + tb] = 0.5*(face[Y][ta] + face[Z][tb]);
    }
  }
  return (sbot[X] != 0.0 || sbot[Y] != 0.0 || sbot[Z] != 0.0 ||
	  stop[X] != 0.0 || stop[Y] != 0.0 || stop[Z] != 0.0);
}

static void slip_normal(const lbo_model_t * model, const char * status,
			const ptrdiff_t str[3], int i, int p, int wn[3]) {
  for (int ia = 0; ia < 3; ia++) {
    ptrdiff_t j = i + str[ia]*model->cv[p][ia];
    wn[ia] = (status[j] != MAP_FLUID) ? -model->cv[p][ia] : 0;
  }
}

int lbo_wall_slip_links(const lbo_param_t * p, const char * status, int nlink,
			const int * linki, const int * linkp, int * linkk,
			int * linkq, int * links) {
  int nall[3];
  ptrdiff_t str[3];
  lbo_model_t model;

  if (lbo_model_create(p->nvel, &model) != 0) return -1;
  strides(p, nall, str);

  for (int n = 0; n < nlink; n++) {
    int wn[3], wt[3];
    int pn = linkp[n];
    int cvdotwn, modwn, modwt;

    slip_normal(&model, status, str, linki[n], pn, wn);
    cvdotwn = model.cv[pn][X]*wn[X] + model.cv[pn][Y]*wn[Y] + model.cv[pn][Z]*wn[Z];
    modwn = wn[X]*wn[X] + wn[Y]*wn[Y] + wn[Z]*wn[Z];
    if (modwn == 0 || modwn != -cvdotwn) return -2;     /* wall.c:557-558 */
    for (int ia = 0; ia < 3; ia++) {
      wt[ia] = model.cv[pn][ia] - cvdotwn*wn[ia]/modwn;
    }
    modwt = wt[X]*wt[X] + wt[Y]*wt[Y] + wt[Z]*wt[Z];

    if (modwt == 0) {
      linkk[n] = linki[n];
      linkq[n] = pn;
      links[n] = 0;
    }
    else {
      int cq[3];
      int q = -1;
      int s = 0;
      linkk[n] = (int) (linki[n] + str[X]*wt[X] + str[Y]*wt[Y] + str[Z]*wt[Z]);
      for (int ia = 0; ia < 3; ia++) cq[ia] = -2*wn[ia] - model.cv[pn][ia];
      for (int m = 0; m < p->nvel; m++) {
	if (cq[X] == model.cv[m][X] && cq[Y] == model.cv[m][Y] &&
	    cq[Z] == model.cv[m][Z]) q = m;
      }
      if (q <= 0) return -3;
      linkq[n] = q;
      if (modwn == 1) {
	if (wn[X] == +1) s = 1;
	if (wn[X] == -1) s = 2;
	if (wn[Y] == +1) s = 3;
	if (wn[Y] == -1) s = 4;
	if (wn[Z] == +1) s = 5;
	if (wn[Z] == -1) s = 6;
      }
      if (modwn == 2) {
	/* XB_YB XB_YT XB_ZB XB_ZT XT_YB XT_YT XT_ZB XT_ZT YB_ZB YB_ZT YT_ZB YT_ZT */
	if (wn[X] != 0 && wn[Y] != 0) s = 7 + 4*(wn[X] == -1) + (wn[Y] == -1);
	if (wn[X] != 0 && wn[Z] != 0) s = 9 + 4*(wn[X] == -1) + (wn[Z] == -1);
	if (wn[Y] != 0 && wn[Z] != 0) s = 15 + 2*(wn[Y] == -1) + (wn[Z] == -1);
      }
      links[n] = s;
    }
  }
  return 0;
}

int lbo_wall_bbl_slip(const lbo_param_t * p, double * f, int nlink,
		      const int * linki, const int * linkj, const int * linkp,
		      const int * linkk, const int * linkq, const int * links,
		      const double stab[19], double fnet[3],
		      const char * status) {
  int nall[3];
  ptrdiff_t str[3];
  ptrdiff_t nsite;
  lbo_model_t model;

  if (lbo_model_create(p->nvel, &model) != 0) return -1;
  strides(p, nall, str);
  nsite = (ptrdiff_t) nall[X]*nall[Y]*nall[Z];

  for (int n = 0; n < nlink; n++) {
    int ij = linkp[n];
    int ji = p->nvel - ij;
    int q = linkq[n];
    double s = stab[links[n]];
    double fi = f[nsite*ij + linki[n]];
    double fk = f[nsite*q + linkk[n]];
    double fp = (1.0 - s)*fi + s*fk;

    if (wall_colloid_link(&model, p->nvel, nsite, f, status, linki[n],
			  linkj[n], ij, fnet)) continue;
    f[nsite*ji + linkj[n]] = fp;

    for (int ia = 0; ia < 3; ia++) {
      int iw = -(model.cv[ij][ia] + model.cv[q][ia])/2;      /* int, as there */
      double w = iw;
      fnet[ia] += 2.0*(1.0 - s)*(fi - model.wv[ij])*model.cv[ij][ia];
      fnet[ia] += 2.0*w*w*s*(fk - model.wv[q])*model.cv[q][ia];
    }
  }
  return 0;
}

/*
 * lbo_halo
 *
 * Net effect on one rank with periodic boundaries of lb_halo() with the
 * LB_HALO_TARGET scheme, i.e. halo_swap_packed (halo_swap.c:709-1063) with
 * nswap = 1: three sequential passes X, Y, Z; each copies the first/last
 * interior plane (full extent, *including* halo, in the other two
 * directions) to the opposite width-1 halo layer adjacent to the interior,
 * so that edges and corners are completed by the later passes (that is what
 * the host corner fill of halo_swap.c:893-915,968-1012 achieves). With
 * nhalo > 1 only the layer next to the interior is filled.
 *
 * data has nel components in SoA order: data[nsite*n + index].
 */

int lbo_halo(const lbo_param_t * p, int nel, double * data) {
  return lbo_halo_width(p, nel, data, 7, 1);
}

int lbo_halo_dirs(const lbo_param_t * p, int nel, double * data, int dirmask) {
  return lbo_halo_width(p, nel, data, dirmask, 1);
}

/* dirmask: bit d set = run the pass for direction d (X = 1, Y = 2, Z = 4);
 * a slab-decomposed test does X through its own exchange and Y, Z here. */

/* nswap: number of halo layers exchanged (halo_swap_create's nhcomm:
 * 1 for the distributions, model.c:431-440; the field's own halo width for
 * field_halo, e.g. 2 for phi with the symmetric free energy). */

int lbo_halo_width(const lbo_param_t * p, int nel, double * data, int dirmask,
		   int nswap) {

  int nall[3];
  ptrdiff_t str[3];
  ptrdiff_t nsite;
  int nh = p->nhalo;

  strides(p, nall, str);
  nsite = (ptrdiff_t) nall[X]*nall[Y]*nall[Z];

  for (int id = 0; id < 3; id++) {
    if (!(dirmask & (1 << id))) continue;
    int d1 = (id + 1) % 3;
    int d2 = (id + 2) % 3;
    /* plane coordinates (0-based, in nall): source lo/hi, destination */
    for (int w = 0; w < nswap; w++) {
    int src_lo = nh + w;                      /* w-th interior plane */
    int src_hi = nh + p->nlocal[id] - 1 - w;  /* w-th from the top */
    int dst_lo = nh - 1 - w;                  /* halo plane below */
    int dst_hi = nh + p->nlocal[id] + w;      /* halo plane above */
    {
    #pragma omp parallel for schedule(static)
    for (int n = 0; n < nel; n++) {
      double * d = data + nsite*n;
      for (int i1 = 0; i1 < nall[d1]; i1++) {
	for (int i2 = 0; i2 < nall[d2]; i2++) {
	  ptrdiff_t off = str[d1]*i1 + str[d2]*i2;
	  d[off + str[id]*dst_lo] = d[off + str[id]*src_hi];
	  d[off + str[id]*dst_hi] = d[off + str[id]*src_lo];
	}
      }
    }
    }
    }
  }

  return 0;
}

/*
 * lbo_propagate
 *
 * Pull streaming, lb_propagation_kernel (propagation.c:162-212):
 * fprime[index, p] = f[index - cv[p].str, p] for interior sites; the 1-d
 * iteration covers whole x-planes imin..imax (kernel.c:172-188: kindex0 is
 * the first site of plane imin, extent nlocal[X]*nall[Y]*nall[Z]), so y/z
 * halo sites in those planes copy in place (mask = 0, kernel.c:374-400).
 * The x halo planes of fprime are not written. Pointer swap (propagation.c:223-252) is the
 * caller's business.
 */

int lbo_propagate(const lbo_param_t * p, const double * f, double * fprime) {

  int nall[3];
  ptrdiff_t str[3];
  ptrdiff_t nsite;
  int nh = p->nhalo;
  lbo_model_t model;

  if (lbo_model_create(p->nvel, &model) != 0) return -1;

  strides(p, nall, str);
  nsite = (ptrdiff_t) nall[X]*nall[Y]*nall[Z];

  #pragma omp parallel for collapse(2) schedule(static)
  for (int q = 0; q < p->nvel; q++) {
    for (int i = nh; i < nh + p->nlocal[X]; i++) {
      ptrdiff_t dp = model.cv[q][X]*str[X] + model.cv[q][Y]*str[Y]
	+ model.cv[q][Z]*str[Z];
      for (int j = 0; j < nall[Y]; j++) {
	for (int k = 0; k < nall[Z]; k++) {
	  ptrdiff_t index = str[X]*i + str[Y]*j + k;
	  int interior = (j >= nh && j < nh + p->nlocal[Y] &&
			  k >= nh && k < nh + p->nlocal[Z]);
	  fprime[nsite*q + index] = f[nsite*q + index - (interior ? dp : 0)];
	}
      }
    }
  }

  return 0;
}

/*
 * lbo_moments
 *
 * out[0..4]: volume, sum rho, sum rho^2, min rho, max rho over fluid sites
 *            (stats_distribution_print, stats_distribution.c:55-117; plain
 *            sums in x,y,z order; rho summed over p = 0..nvel-1 as
 *            lb_0th_moment, model.c:819-833).
 * out[5..7]: total momentum sum_p f_p c_p over fluid sites with Kahan
 *            compensation over p = 1..nvel-1 (distribution_gm_kernel,
 *            stats_distribution.c:281-350; kahan_add_double util_sum.c:30).
 *            The reference's accumulation order across sites depends on its
 *            thread decomposition; here sites are visited in x,y,z order by
 *            one accumulator (the serial, one-thread order).
 * out[8]   : unused (0).
 */

typedef struct {double sum; double cs;} kahan_acc_t;

static void kahan_add_val(kahan_acc_t * k, double val) {
  volatile double y = val + k->cs;
  volatile double t = k->sum + y;
  k->cs  = y - (t - k->sum);
  k->sum = t;
}

int lbo_moments(const lbo_param_t * p, const double * f, const char * status,
		double out[9]) {

  int nall[3];
  ptrdiff_t str[3];
  ptrdiff_t nsite;
  lbo_model_t model;
  kahan_acc_t g[3] = {{0.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}};

  if (lbo_model_create(p->nvel, &model) != 0) return -1;

  strides(p, nall, str);
  nsite = (ptrdiff_t) nall[X]*nall[Y]*nall[Z];

  out[0] = 0.0; out[1] = 0.0; out[2] = 0.0;
  out[3] = +DBL_MAX; out[4] = -DBL_MAX;
  out[8] = 0.0;

  for (int ic = 1; ic <= p->nlocal[X]; ic++) {
    for (int jc = 1; jc <= p->nlocal[Y]; jc++) {
      for (int kc = 1; kc <= p->nlocal[Z]; kc++) {
	ptrdiff_t index = str[X]*(p->nhalo + ic - 1)
	  + str[Y]*(p->nhalo + jc - 1) + (p->nhalo + kc - 1);
	double rho = 0.0;
	if (status && status[index] != MAP_FLUID) continue;
	for (int q = 0; q < p->nvel; q++) rho += f[nsite*q + index];
	out[0] += 1.0;
	out[1] += rho;
	out[2] += rho*rho;
	if (rho < out[3]) out[3] = rho;
	if (rho > out[4]) out[4] = rho;
	for (int q = 1; q < p->nvel; q++) {
	  double fq = f[nsite*q + index];
	  kahan_add_val(&g[X], fq*model.cv[q][X]);
	  kahan_add_val(&g[Y], fq*model.cv[q][Y]);
	  kahan_add_val(&g[Z], fq*model.cv[q][Z]);
	}
      }
    }
  }

  out[5] = g[X].sum + g[X].cs;
  out[6] = g[Y].sum + g[Y].cs;
  out[7] = g[Z].sum + g[Z].cs;

  return 0;
}

/*
 * lbo_equilibrium
 *
 * Second-order equilibrium, lb_1st_moment_equilib_set (model.c:915-941):
 * f_p = rho w_p (1 + 3 u.c + 4.5 (c c - delta/3):uu) with the reference's
 * loop order for the double contraction.
 */

int lbo_equilibrium(const lbo_model_t * m, double rho, const double u[3],
		    double * feq) {

  for (int p = 0; p < m->nvel; p++) {
    double rcs2 = 1.0/cs2;
    double udotc = 0.0;
    double sdotq = 0.0;
    for (int ia = 0; ia < 3; ia++) {
      udotc += u[ia]*m->cv[p][ia];
      for (int ib = 0; ib < 3; ib++) {
	double dab = (ia == ib);
	sdotq += (m->cv[p][ia]*m->cv[p][ib] - cs2*dab)*u[ia]*u[ib];
      }
    }
    feq[p] = rho*m->wv[p]*(1.0 + rcs2*udotc + 0.5*rcs2*rcs2*sdotq);
  }

  return 0;
}

/*
 * lbo_init_synthetic
 *
 * The seeded synthetic state of SURVEY.md section 8(d) (the same generator
 * as oracle/ref_driver.c:init_f, which produced the golden vectors):
 * equilibrium of rho = 1 + 0.01 cos(2 pi (x/Lx + y/Ly + z/Lz)),
 * u = 0.01 (sin 2 pi y/Ly, sin 2 pi z/Lz, sin 2 pi x/Lx), each population
 * multiplied by (1 + 1e-3 (r - 1/2)), r from the 32-bit LCG
 * s = 1664525 s + 1013904223 (seed 12345) advanced in global (x,y,z,p)
 * order. ntotal is the global box, noffset the offset of this sub-domain
 * (zero for a single domain). Halo sites are set to zero.
 */

static void lcg_jump(uint32_t n, uint32_t * amul, uint32_t * cadd) {
  /* (a, c) of the n-fold composition of s -> 1664525 s + 1013904223 */
  uint32_t a = 1u, c = 0u;
  uint32_t ab = 1664525u, cb = 1013904223u;
  while (n) {
    if (n & 1u) { a = ab*a; c = ab*c + cb; }
    cb = (ab + 1u)*cb;
    ab = ab*ab;
    n >>= 1;
  }
  *amul = a; *cadd = c;
}

int lbo_init_synthetic(const lbo_param_t * p, const int ntotal[3],
		       const int noffset[3], double * f) {

  int nall[3];
  ptrdiff_t str[3];
  ptrdiff_t nsite;
  lbo_model_t model;
  const double pi = 3.14159265358979323846;

  if (lbo_model_create(p->nvel, &model) != 0) return -1;

  strides(p, nall, str);
  nsite = (ptrdiff_t) nall[X]*nall[Y]*nall[Z];
  memset(f, 0, sizeof(double)*nsite*p->nvel);

  #pragma omp parallel for collapse(2) schedule(static)
  for (int ic = 1; ic <= p->nlocal[X]; ic++) {
    for (int jc = 1; jc <= p->nlocal[Y]; jc++) {
      /* global position of the first site of this z-row in LCG order */
      int ig = noffset[X] + ic - 1;
      int jg = noffset[Y] + jc - 1;
      uint64_t n0 = (((uint64_t) ig*ntotal[Y] + jg)*ntotal[Z] + noffset[Z])
	*(uint64_t) p->nvel;
      uint32_t a, c, s;
      lcg_jump((uint32_t) (n0 & 0xffffffffu), &a, &c);
      s = a*12345u + c;
      for (int kc = 1; kc <= p->nlocal[Z]; kc++) {
	ptrdiff_t index = str[X]*(p->nhalo + ic - 1)
	  + str[Y]*(p->nhalo + jc - 1) + (p->nhalo + kc - 1);
	double x = (ig*1.0)/ntotal[X];
	double y = (jg*1.0)/ntotal[Y];
	double z = (noffset[Z] + kc - 1.0)/ntotal[Z];
	double rho = 1.0 + 0.01*cos(2.0*pi*(x + y + z));
	double u[3];
	double feq[LBO_NVEL_MAX];
	u[X] = 0.01*sin(2.0*pi*y);
	u[Y] = 0.01*sin(2.0*pi*z);
	u[Z] = 0.01*sin(2.0*pi*x);
	lbo_equilibrium(&model, rho, u, feq);
	for (int q = 0; q < p->nvel; q++) {
	  double r;
	  s = 1664525u*s + 1013904223u;
	  r = s/4294967296.0;
	  f[nsite*q + index] = feq[q]*(1.0 + 1.0e-3*(r - 0.5));
	}
      }
    }
  }

  return 0;
}

/*
 * lbo_records_pack / lbo_records_unpack
 *
 * The binary record stream of the distribution files: lb_io_aggr_pack with
 * lb_write_buf (model.c:1385-1402, 1479-1510) -- one record of nvel doubles
 * in p order per INTERIOR site, sites in (ic, jc, kc) order, independent of
 * the memory order -- and its inverse lb_io_aggr_unpack / lb_read_buf
 * (model.c:1412-1430, 1520-1550). With ndist distributions the record is [n][p],
 * i.e. the component order of f: call with p->nvel = ndist*nvel (only the count
 * is used here).
 */

int lbo_records_pack(const lbo_param_t * p, const double * f, double * rec) {
  int nall[3];
  ptrdiff_t str[3];
  ptrdiff_t nsite;
  strides(p, nall, str);
  nsite = (ptrdiff_t) nall[X]*nall[Y]*nall[Z];
  ptrdiff_t ib = 0;
  for (int ic = 1; ic <= p->nlocal[X]; ic++) {
    for (int jc = 1; jc <= p->nlocal[Y]; jc++) {
      for (int kc = 1; kc <= p->nlocal[Z]; kc++) {
	ptrdiff_t index = str[X]*(p->nhalo + ic - 1)
	  + str[Y]*(p->nhalo + jc - 1) + (p->nhalo + kc - 1);
	for (int q = 0; q < p->nvel; q++) rec[ib*p->nvel + q] = f[nsite*q + index];
	ib += 1;
      }
    }
  }
  return 0;
}

int lbo_records_unpack(const lbo_param_t * p, double * f, const double * rec) {
  int nall[3];
  ptrdiff_t str[3];
  ptrdiff_t nsite;
  strides(p, nall, str);
  nsite = (ptrdiff_t) nall[X]*nall[Y]*nall[Z];
  ptrdiff_t ib = 0;
  for (int ic = 1; ic <= p->nlocal[X]; ic++) {
    for (int jc = 1; jc <= p->nlocal[Y]; jc++) {
      for (int kc = 1; kc <= p->nlocal[Z]; kc++) {
	ptrdiff_t index = str[X]*(p->nhalo + ic - 1)
	  + str[Y]*(p->nhalo + jc - 1) + (p->nhalo + kc - 1);
	for (int q = 0; q < p->nvel; q++) f[nsite*q + index] = rec[ib*p->nvel + q];
	ib += 1;
      }
    }
  }
  return 0;
}

/*
 * lbo_grad_7pt
 *
 * grad_3d_7pt_fluid_d2 (gradient_3d_7pt_fluid.c:232-320, no Lees-Edwards
 * planes): central differences grad_a = (phi(+e_a) - phi(-e_a))/2 and the
 * 7-point Laplacian, for the sites 1-nextra .. nlocal+nextra in every
 * direction (nextra = nhalo - 1). grad: 3*nsite (SoA, component slowest),
 * delsq: nsite. Summation order of the Laplacian as the reference:
 * +x -x +y -y +z -z then -6 phi.
 */

int lbo_grad_7pt(const lbo_param_t * p, const double * phi, double * grad,
		 double * delsq) {

  int nall[3];
  ptrdiff_t str[3];
  ptrdiff_t nsite;
  int nextra = p->nhalo - 1;

  strides(p, nall, str);
  nsite = (ptrdiff_t) nall[X]*nall[Y]*nall[Z];

  for (int ic = 1 - nextra; ic <= p->nlocal[X] + nextra; ic++) {
    for (int jc = 1 - nextra; jc <= p->nlocal[Y] + nextra; jc++) {
      for (int kc = 1 - nextra; kc <= p->nlocal[Z] + nextra; kc++) {
	ptrdiff_t i = str[X]*(p->nhalo + ic - 1)
	  + str[Y]*(p->nhalo + jc - 1) + (p->nhalo + kc - 1);
	grad[0*nsite + i] = 0.5*(phi[i + str[X]] - phi[i - str[X]]);
	grad[1*nsite + i] = 0.5*(phi[i + str[Y]] - phi[i - str[Y]]);
	grad[2*nsite + i] = 0.5*(phi[i + 1] - phi[i - 1]);
	delsq[i] = phi[i + str[X]] + phi[i - str[X]]
	  + phi[i + str[Y]] + phi[i - str[Y]]
	  + phi[i + 1] + phi[i - 1]
	  - 6.0*phi[i];
      }
    }
  }

  return 0;
}

/*
 * lbo_grad_27pt
 *
 * grad_3d_27pt_kernel, GRAD_DEL2 (gradient_3d_27pt_fluid.c:216-364, no
 * Lees-Edwards planes): grad_a = (1/18) sum over the nine pairs of the 3x3
 * plane normal to a of (phi(+e_a) - phi(-e_a)); delsq = (1/9) (sum of the 26
 * neighbours - 26 phi). Same extent as lbo_grad_7pt. Summation orders are
 * the reference's: for grad_x the pairs run over (dy, dz) with dz fastest;
 * for grad_y over (dx, dz); for grad_z over (dx, dy); the Laplacian runs
 * over (dx, dy, dz), dz fastest, and subtracts the centre last.
 */

int lbo_grad_27pt(const lbo_param_t * p, const double * phi, double * grad,
		  double * delsq) {

  int nall[3];
  ptrdiff_t str[3];
  ptrdiff_t nsite;
  int nextra = p->nhalo - 1;
  const double r9 = (1.0/9.0);

  strides(p, nall, str);
  nsite = (ptrdiff_t) nall[X]*nall[Y]*nall[Z];

  for (int ic = 1 - nextra; ic <= p->nlocal[X] + nextra; ic++) {
    for (int jc = 1 - nextra; jc <= p->nlocal[Y] + nextra; jc++) {
      for (int kc = 1 - nextra; kc <= p->nlocal[Z] + nextra; kc++) {
	ptrdiff_t i = str[X]*(p->nhalo + ic - 1)
	  + str[Y]*(p->nhalo + jc - 1) + (p->nhalo + kc - 1);
	double v[3][3][3];
	double sx = 0.0, sy = 0.0, sz = 0.0, d2 = 0.0;

	for (int a = 0; a < 3; a++) {
	  for (int b = 0; b < 3; b++) {
	    for (int c = 0; c < 3; c++) {
	      v[a][b][c] = phi[i + (a - 1)*str[X] + (b - 1)*str[Y] + (c - 1)];
	    }
	  }
	}
	for (int b = 0; b < 3; b++) {
	  for (int c = 0; c < 3; c++) {
	    sx += v[2][b][c]; sx -= v[0][b][c];
	    sy += v[b][2][c]; sy -= v[b][0][c];
	    sz += v[b][c][2]; sz -= v[b][c][0];
	  }
	}
	for (int a = 0; a < 3; a++) {
	  for (int b = 0; b < 3; b++) {
	    for (int c = 0; c < 3; c++) {
	      if (a == 1 && b == 1 && c == 1) continue;
	      d2 += v[a][b][c];
	    }
	  }
	}
	d2 -= 26.0*v[1][1][1];
	grad[0*nsite + i] = 0.5*r9*sx;
	grad[1*nsite + i] = 0.5*r9*sy;
	grad[2*nsite + i] = 0.5*r9*sz;
	delsq[i] = r9*d2;
      }
    }
  }

  return 0;
}

/*
 * lbo_symm_force
 *
 * Thermodynamic force of the symmetric free energy by stress divergence:
 * pth_stress_compute with fe_symm_str_v (phi_force_stress.c:171-300,
 * symmetric.c:371-420) for the sites 0 .. nlocal+1, then
 * pth_force_fluid_kernel_v (phi_force_colloid.c:324-480):
 *   P_ab = p0 delta_ab + kappa d_a phi d_b phi,
 *   p0 = a phi^2/2 + 3 b phi^4/4 - kappa phi delsq - kappa |grad phi|^2/2,
 *   F_a = - sum_b [ (P_ab(+e_b) + P_ab)/2 - (P_ab(-e_b) + P_ab)/2 ],
 * accumulated in the reference's order (+x, -x, +y, -y, +z, -z), and ADDED
 * to force (3*nsite) at the interior sites.
 */

static void symm_stress(double a, double b, double kappa, double phi,
			const double g[3], double delsq, double s[3][3]) {
  double p0 = 0.5*a*phi*phi + 0.75*b*phi*phi*phi*phi - kappa*phi*delsq
    - 0.5*kappa*(g[X]*g[X] + g[Y]*g[Y] + g[Z]*g[Z]);
  for (int ia = 0; ia < 3; ia++) {
    for (int ib = 0; ib < 3; ib++) {
      s[ia][ib] = p0*(ia == ib) + kappa*g[ia]*g[ib];
    }
  }
}

int lbo_symm_force(const lbo_param_t * p, double a, double b, double kappa,
		   const double * phi, const double * grad,
		   const double * delsq, double * force) {

  int nall[3];
  ptrdiff_t str[3];
  ptrdiff_t nsite;

  strides(p, nall, str);
  nsite = (ptrdiff_t) nall[X]*nall[Y]*nall[Z];

  for (int ic = 1; ic <= p->nlocal[X]; ic++) {
    for (int jc = 1; jc <= p->nlocal[Y]; jc++) {
      for (int kc = 1; kc <= p->nlocal[Z]; kc++) {
	ptrdiff_t i = str[X]*(p->nhalo + ic - 1)
	  + str[Y]*(p->nhalo + jc - 1) + (p->nhalo + kc - 1);
	double pth0[3][3], pth1[3][3];
	double f[3];
	double g[3];

	g[X] = grad[i]; g[Y] = grad[nsite + i]; g[Z] = grad[2*nsite + i];
	symm_stress(a, b, kappa, phi[i], g, delsq[i], pth0);

	for (int id = 0; id < 3; id++) {
	  for (int sgn = +1; sgn >= -1; sgn -= 2) {
	    ptrdiff_t j = i + sgn*str[id];
	    g[X] = grad[j]; g[Y] = grad[nsite + j]; g[Z] = grad[2*nsite + j];
	    symm_stress(a, b, kappa, phi[j], g, delsq[j], pth1);
	    for (int ia = 0; ia < 3; ia++) {
	      double face = 0.5*(pth1[ia][id] + pth0[ia][id]);
	      if (id == 0 && sgn == +1) f[ia] = -face;
	      else if (sgn == +1) f[ia] -= face;
	      else f[ia] += face;
	    }
	  }
	}
	for (int ia = 0; ia < 3; ia++) force[nsite*ia + i] += f[ia];
      }
    }
  }

  return 0;
}

/*
 * lbo_cahn_hilliard
 *
 * One Cahn-Hilliard step of the symmetric binary fluid as phi_cahn_hilliard
 * runs it without noise, walls or Lees-Edwards planes
 * (phi_cahn_hilliard.c:195-284):
 *   advective fluxes of order 1..4 (advection_x, advection.c:433-482): at
 *   the face between i and its neighbour the velocity is the mean of the
 *   two site velocities; phi at the face is upwind (1), the mean (2), a
 *   three-point upwind-biased (3) or a four-point centred (4) interpolation;
 *   diffusive fluxes -M (mu_1 - mu_0) with mu = a phi + b phi^3 - kappa
 *   delsq (phi_ch_flux_mu1_kernel :349-402, fe_symm_mu symmetric.c:303-316);
 *   forward step phi -= fe - fw + fy - fy(-y) + fz - fz(-z)
 *   (phi_ch_ufs_kernel :1026-1060).
 * u must carry a valid width-1 halo (hydro_u_halo), delsq the width-1 layer
 * around the interior and phi its width-2 halo (orders 3 and 4 reach two
 * sites). phi is updated in place at the
 * interior sites; work: 4*nsite doubles (the flux arrays fw, fe, fy, fz).
 */

/* Advective flux through the face between the sites l and l + s (s = the
 * stride of the face normal), face velocity uf = (u_l + u_{l+s})/2. west = 1
 * selects the "west" form of the first-order kernel (it differs from the
 * other only in which side a zero velocity takes). */

static double advective_flux(int order, int west, double uf,
			     const double * phi, ptrdiff_t l, ptrdiff_t s) {
  const ptrdiff_t r = l + s;
  switch (order) {
  case 1:
    /* advection_le_1st_kernel (advection.c:542-640) */
    if (west) return uf*phi[(uf > 0.0) ? l : r];
    return uf*phi[(uf < 0.0) ? r : l];
  case 2:
    /* advection_2nd_kernel_v (advection.c:790-916) */
    return uf*0.5*(phi[l] + phi[r]);
  case 3:
    /* advection_le_3rd_kernel_v (advection.c:977-1176) */
    {
      const double a1 = -0.213933;
      const double a2 =  0.927865;
      const double a3 =  0.286067;
      int down = west ? !(uf > 0.0) : (uf < 0.0);
      if (down) return uf*(a1*phi[r + s] + a2*phi[r] + a3*phi[l]);
      return uf*(a1*phi[l - s] + a2*phi[l] + a3*phi[r]);
    }
  default:
    /* advection_le_4th (advection.c:1188-1296) */
    {
      const double a1 = (1.0/16.0);
      const double a2 = (9.0/16.0);
      return uf*(- a1*phi[l - s] + a2*phi[l] + a2*phi[r] - a1*phi[r + s]);
    }
  }
}

int lbo_cahn_hilliard(const lbo_param_t * p, double a, double b, double kappa,
		      double mobility, int order, double * phi,
		      const double * delsq, const double * u, double * work) {

  int nall[3];
  ptrdiff_t str[3];
  ptrdiff_t nsite;
  double * fw, * fe, * fy, * fz;

  strides(p, nall, str);
  nsite = (ptrdiff_t) nall[X]*nall[Y]*nall[Z];
  fw = work; fe = work + nsite; fy = work + 2*nsite; fz = work + 3*nsite;

#define MU(j) (a*phi[j] + b*phi[j]*phi[j]*phi[j] - kappa*delsq[j])

  /* fluxes for ic = 1..nlocal, jc, kc = 0..nlocal (advection.c:508-510) */
  for (int ic = 1; ic <= p->nlocal[X]; ic++) {
    for (int jc = 0; jc <= p->nlocal[Y]; jc++) {
      for (int kc = 0; kc <= p->nlocal[Z]; kc++) {
	ptrdiff_t i = str[X]*(p->nhalo + ic - 1)
	  + str[Y]*(p->nhalo + jc - 1) + (p->nhalo + kc - 1);
	ptrdiff_t j;
	double uf, mu0 = MU(i);

	j = i - str[X];                              /* west face */
	uf = 0.5*(u[i] + u[j]);
	fw[i] = advective_flux(order, 1, uf, phi, j, str[X]);
	fw[i] -= mobility*(mu0 - MU(j));

	j = i + str[X];                              /* east face */
	uf = 0.5*(u[i] + u[j]);
	fe[i] = advective_flux(order, 0, uf, phi, i, str[X]);
	fe[i] -= mobility*(MU(j) - mu0);

	j = i + str[Y];
	uf = 0.5*(u[nsite + i] + u[nsite + j]);
	fy[i] = advective_flux(order, 0, uf, phi, i, str[Y]);
	fy[i] -= mobility*(MU(j) - mu0);

	j = i + 1;
	uf = 0.5*(u[2*nsite + i] + u[2*nsite + j]);
	fz[i] = advective_flux(order, 0, uf, phi, i, 1);
	fz[i] -= mobility*(MU(j) - mu0);
      }
    }
  }
#undef MU

  for (int ic = 1; ic <= p->nlocal[X]; ic++) {
    for (int jc = 1; jc <= p->nlocal[Y]; jc++) {
      for (int kc = 1; kc <= p->nlocal[Z]; kc++) {
	ptrdiff_t i = str[X]*(p->nhalo + ic - 1)
	  + str[Y]*(p->nhalo + jc - 1) + (p->nhalo + kc - 1);
	double wz = (p->nlocal[Z] == 1) ? 0.0 : 1.0;
	phi[i] -= (+ fe[i] - fw[i] + fy[i] - fy[i - str[Y]]
		   + wz*fz[i] - wz*fz[i - 1]);
      }
    }
  }

  return 0;
}
