"""Host-side mirror of the reference's LB interface on top of liblbmi.so.

Names follow the reference (zazu29/ludwig): lb_collide (collision.c:143),
lb_halo / lb_halo_swap (model.c:553-595), lb_propagation (propagation.c:43),
lb_memcpy (model.c:228), hydro_t fields rho/u/force (hydro.h:31-47),
map_t status (map.h:22-40). PyTorch is used only as plumbing: device memory
(torch tensors) and host<->device copies. All arithmetic happens in the HIP
kernels behind the C-ABI; nothing here computes on the CPU.

Array convention (the reference's SoA order, see include/lbmi.h):
  f:      (nvel, nall_x, nall_y, nall_z) float64
  force:  (3, nall_x, nall_y, nall_z),  u likewise;  rho: (nall_x, ...)
  status: (nall_x, nall_y, nall_z) int8, 0 = MAP_FLUID
"""

import ctypes
import weakref

import numpy as np

from . import lib as _l

M10, BGK, TRT = 0, 1, 2                    # lb_relaxation_enum_t
EAGER, FUSED, INPLACE, FUSED_HALO = 0, 1, 2, 3   # lbmi_mode_t
# Python-side shorthand: FUSED with the deferred state kept in the reference's
# SoA order (lbmi_tune "blocked" = 0) instead of the default blocked order
FUSED_SOA = 101
HALO_FULL, HALO_REDUCED = 0, 2             # lbmi_halo_t
_SCHEMES = {"m10": M10, "bgk": BGK, "trt": TRT}


def _torch():
    import torch
    if not torch.cuda.is_available():
        raise _l.LbmiError("no GPU visible to torch: ludwig_amd has no CPU path")
    return torch


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


class Hydro:
    """Device arrays of hydro_t (+ map_t status) that lb_collide borrows."""

    def __init__(self, nall, device, force=None, status=None, with_rho_u=True,
                 eta=None):
        torch = _torch()
        self.nall = tuple(nall)
        self.device = device
        self.force = None
        self.status = None
        self.rho = None
        self.u = None
        self.eta = None
        if eta is not None:
            # hydro->eta of a viscosity model: local shear viscosity
            self.eta = torch.from_numpy(np.ascontiguousarray(
                eta, dtype=np.float64)).to(device)
            assert tuple(self.eta.shape) == self.nall
        if force is not None:
            self.force = torch.from_numpy(np.ascontiguousarray(
                force, dtype=np.float64)).to(device)
            assert tuple(self.force.shape) == (3,) + self.nall
        if status is not None:
            self.status = torch.from_numpy(np.ascontiguousarray(
                status, dtype=np.int8)).to(device)
            assert tuple(self.status.shape) == self.nall
        if with_rho_u:
            self.rho = torch.zeros(self.nall, dtype=torch.float64, device=device)
            self.u = torch.zeros((3,) + self.nall, dtype=torch.float64,
                                 device=device)
        # the kernels run on the library's own stream: make sure torch's
        # fills/copies have landed
        torch.cuda.synchronize(device)

    def ptrs(self):
        h = _l.HydroPtrs()
        h.force = _ptr(self.force)
        h.status = _ptr(self.status)
        h.rho = _ptr(self.rho)
        h.u = _ptr(self.u)
        h.eta = _ptr(self.eta)
        # lbmi_hydro_t::nsite: the distance between the components of force
        # and u when it is not the lattice's nsite (attribute `stride`, set by
        # a caller who has allocated them that way; 0 = the lattice's)
        h.nsite = int(getattr(self, "stride", 0) or 0)
        return h


class LB:
    """lb_t: the distributions of one rank and the operators of a time step."""

    def __init__(self, nvel=19, nlocal=(64, 64, 64), nhalo=1, mode=EAGER,
                 halo_scheme=HALO_FULL, device=0, cartsz=1, cartrank=0,
                 own_stream=False, ndist=1, cartdim=0, cartgrid=None,
                 cartcoords=None):
        """own_stream=False (default): the library works on torch's current
        stream of `device`, so its kernels are ordered with torch operations
        on the same tensors. own_stream=True keeps the handle's private
        non-blocking stream: then the caller must synchronise explicitly
        (lb.synchronize() / torch.cuda.synchronize()) between torch
        operations and library calls that touch the same memory."""
        torch = _torch()
        self._lib = _l.library()
        self._h = ctypes.c_void_p()
        self._zeroed = {}          # data_ptr -> (weakref, version): _zeros_still_hold
        opts = _l.Options()
        _l.check(self._lib.lbmi_options_default(ctypes.byref(opts)))
        opts.nvel = nvel
        opts.ndist = ndist
        opts.nlocal[:] = list(nlocal)
        opts.nhalo = nhalo
        opts.device = device
        opts.mode = FUSED if mode == FUSED_SOA else mode
        opts.halo_scheme = halo_scheme
        opts.cartsz = cartsz
        opts.cartrank = cartrank
        opts.cartdim = cartdim
        if cartgrid is not None:
            # LBMI_CART_GENERAL: a grid of ranks, this one at cartcoords
            _cart_options(opts, cartgrid, cartcoords)
        _l.check(self._lib.lbmi_create(ctypes.byref(opts), ctypes.byref(self._h)))
        self.nvel = nvel
        self.ndist = ndist
        self.nlocal = tuple(nlocal)
        self.nhalo = nhalo
        self.mode = mode
        self.nall = tuple(n + 2 * nhalo for n in nlocal)
        self.nsite = self.nall[0] * self.nall[1] * self.nall[2]
        self.device = torch.device("cuda", device)
        # lb_data_create zero-initialises f and fprime (model.c:106-147)
        self._a = torch.zeros((ndist * nvel,) + self.nall, dtype=torch.float64,
                              device=self.device)
        self._b = torch.zeros_like(self._a)
        torch.cuda.synchronize(self.device)
        if not own_stream:
            st = torch.cuda.current_stream(self.device).cuda_stream
            _l.check(self._lib.lbmi_set_stream(self._h, ctypes.c_void_p(st)))
        _l.check(self._lib.lbmi_lb_bind(self._h, _ptr(self._a), _ptr(self._b)))
        if mode == FUSED_SOA:
            self.tune("blocked", 0)

    # -- life cycle ---------------------------------------------------------

    def free(self):
        if self._h:
            self._lib.lbmi_free(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    # -- parameters ---------------------------------------------------------

    def relaxation_set(self, scheme, eta=0.1, zeta=0.1, rho0=1.0):
        """lb_collision_relaxation_set + relaxation_times_set."""
        if isinstance(scheme, str):
            scheme = _SCHEMES[scheme]
        _l.check(self._lib.lbmi_set_relaxation(self._h, scheme, rho0, eta, zeta))

    def body_force_set(self, fbody):
        a = (ctypes.c_double * 3)(*[float(x) for x in fbody])
        _l.check(self._lib.lbmi_set_body_force(self._h, a))

    def relaxation_rates(self):
        a = (ctypes.c_double * 4)()
        _l.check(self._lib.lbmi_relaxation_rates(self._h, a))
        return tuple(a)

    # -- distributions ------------------------------------------------------

    @property
    def f(self):
        """Device tensor currently playing the role of lb->target->f."""
        pf = ctypes.c_void_p()
        _l.check(self._lib.lbmi_lb_pointers(self._h, ctypes.byref(pf), None))
        return self._a if pf.value == self._a.data_ptr() else self._b

    @property
    def fprime(self):
        return self._b if self.f is self._a else self._a

    def lb_memcpy_h2d(self, f_host):
        f_host = np.ascontiguousarray(f_host, dtype=np.float64)
        assert f_host.shape == (self.ndist * self.nvel,) + self.nall
        _l.check(self._lib.lbmi_lb_memcpy_h2d(
            self._h, f_host.ctypes.data_as(ctypes.c_void_p)))

    def lb_dirty(self):
        """lbmi_lb_dirty: the caller has overwritten the current f on the
        device itself (the canonical state, e.g. through lb.f after a flush)."""
        _l.check(self._lib.lbmi_lb_dirty(self._h))

    def lb_memcpy_d2h(self):
        out = np.empty((self.ndist * self.nvel,) + self.nall, dtype=np.float64)
        _l.check(self._lib.lbmi_lb_memcpy_d2h(
            self._h, out.ctypes.data_as(ctypes.c_void_p)))
        return out

    # -- the time step (ludwig.c:802-860) ----------------------------------

    def _zeros_still_hold(self):
        """The library remembers arrays zeroed through hydro_field_set by
        ADDRESS and does not read a force field it knows to hold zeros
        (include/lbmi.h: any other writer reports its write). torch writes
        without telling, and its caching allocator hands the address of a dead
        tensor to a new one: every such array is checked here before a
        collision -- the tensor must still be the one that was zeroed and no
        in-place operation of torch may have touched it since (its version
        counter); otherwise the library is told (lbmi_hydro_field_dirty)."""
        for ptr, (ref, version) in list(self._zeroed.items()):
            t = ref()
            if t is None or t.data_ptr() != ptr or t._version != version:
                _l.check(self._lib.lbmi_hydro_field_dirty(
                    self._h, ctypes.c_void_p(ptr)))
                del self._zeroed[ptr]

    def lb_collide(self, hydro=None):
        self._zeros_still_hold()
        h = hydro.ptrs() if hydro is not None else None
        _l.check(self._lib.lbmi_lb_collide(
            self._h, ctypes.byref(h) if h is not None else None))

    def lb_halo(self):
        _l.check(self._lib.lbmi_lb_halo(self._h))

    def lb_propagation(self):
        _l.check(self._lib.lbmi_lb_propagation(self._h))

    def lb_flush(self):
        _l.check(self._lib.lbmi_lb_flush(self._h))

    def step(self, hydro=None):
        self.lb_collide(hydro)
        self.lb_halo()
        self.lb_propagation()

    def run(self, hydro=None, nsteps=1):
        """nsteps x (lb_collide, lb_halo, lb_propagation) in one foreign call."""
        self._zeros_still_hold()
        h = hydro.ptrs() if hydro is not None else None
        _l.check(self._lib.lbmi_lb_run(
            self._h, ctypes.byref(h) if h is not None else None, int(nsteps)))

    def moments(self, status=None):
        """Volume, sum rho, sum rho^2, min, max, g_x, g_y, g_z, 0 (interior,
        fluid sites; this rank)."""
        out = (ctypes.c_double * 9)()
        _l.check(self._lib.lbmi_lb_moments(self._h, _ptr(status), out))
        return np.array(out[:])

    # -- stateless kernels on caller-owned tensors ---------------------------

    def collide(self, f, hydro=None):
        h = hydro.ptrs() if hydro is not None else None
        _l.check(self._lib.lbmi_collide(
            self._h, _ptr(f), ctypes.byref(h) if h is not None else None))

    def halo(self, f, scheme=HALO_FULL):
        _l.check(self._lib.lbmi_halo(self._h, _ptr(f), scheme))

    def halo_x_count(self, scheme=HALO_FULL):
        a = ctypes.c_size_t()
        b = ctypes.c_size_t()
        _l.check(self._lib.lbmi_halo_x_count(self._h, scheme, ctypes.byref(a),
                                             ctypes.byref(b)))
        return a.value, b.value

    def halo_x_pack(self, f, sendlo, sendhi, scheme=HALO_FULL):
        _l.check(self._lib.lbmi_halo_x_pack(self._h, _ptr(f), scheme,
                                            _ptr(sendlo), _ptr(sendhi)))

    def halo_x_unpack(self, f, recvlo, recvhi, scheme=HALO_FULL):
        _l.check(self._lib.lbmi_halo_x_unpack(self._h, _ptr(f), scheme,
                                              _ptr(recvlo), _ptr(recvhi)))

    def halo_yz(self, f, scheme=HALO_FULL):
        _l.check(self._lib.lbmi_halo_yz(self._h, _ptr(f), scheme))

    def propagate(self, f, fprime):
        _l.check(self._lib.lbmi_propagate(self._h, _ptr(f), _ptr(fprime)))

    def propagate_collide(self, f, fprime, hydro=None, wrap=True):
        h = hydro.ptrs() if hydro is not None else None
        _l.check(self._lib.lbmi_propagate_collide(
            self._h, _ptr(f), _ptr(fprime),
            ctypes.byref(h) if h is not None else None, 1 if wrap else 0))

    def field_halo(self, data):
        nel = 1 if data.dim() == 3 else data.shape[0]
        _l.check(self._lib.lbmi_field_halo(self._h, nel, _ptr(data)))

    def moments_of(self, f, status=None):
        out = (ctypes.c_double * 9)()
        _l.check(self._lib.lbmi_moments(self._h, _ptr(f), _ptr(status), out))
        return np.array(out[:])

    # -- rows "next": hydro housekeeping, record stream ----------------------

    def hydro_field_set(self, field, values):
        """hydro_u_zero / hydro_f_zero (hydro.c:279-330) on a device field."""
        ncomp = 1 if field.dim() == 3 else field.shape[0]
        a = (ctypes.c_double * 3)(*([float(v) for v in values] + [0.0] * 3)[:3])
        _l.check(self._lib.lbmi_hydro_field_set(self._h, _ptr(field), ncomp, a))
        if all(float(v) == 0.0 for v in values[:ncomp]):
            self._zeroed[field.data_ptr()] = (weakref.ref(field), field._version)
        else:
            self._zeroed.pop(field.data_ptr(), None)

    def density(self):
        """lbmi_lb_density: rho of the interior sites, (nlocal) array."""
        out = np.zeros(self.nlocal, dtype=np.float64)
        _l.check(self._lib.lbmi_lb_density(
            self._h, out.ctypes.data_as(ctypes.c_void_p)))
        return out

    def field_stats(self, field, status=None):
        """lbmi_field_stats: (volume, Kahan sum, sum of squares, min, max)
        of a scalar device field over interior fluid sites."""
        out = (ctypes.c_double * 5)()
        _l.check(self._lib.lbmi_field_stats(
            self._h, _ptr(field), None if status is None else _ptr(status), out))
        return np.array(out[:])

    def hydro_sync(self):
        """lbmi_lb_hydro_sync: rho, u of the last collision, if a lazy
        collision (tune hydro_lazy) still owes them."""
        _l.check(self._lib.lbmi_lb_hydro_sync(self._h))

    def noise_set(self, state, kt, ghosts_on=True):
        """lbmi_noise_set: isothermal fluctuations in lb_collide. state: an
        int32/uint32 device tensor of shape (4, nsite) -- the reference's
        noise->state (SoA) -- advanced in place by every collision; None:
        off. The tensor must stay alive while it is set."""
        if state is None:
            _l.check(self._lib.lbmi_noise_set(self._h, None, 0, 0.0, 0))
            self._noise = None
            return
        assert state.dim() == 2 and state.shape[0] == 4 and state.is_contiguous()
        assert state.element_size() == 4
        self._noise = state
        _l.check(self._lib.lbmi_noise_set(self._h, _ptr(state), int(state.shape[1]),
                                          float(kt), 1 if ghosts_on else 0))

    def hydro_field_dirty(self, field):
        """Somebody outside the library has written to this device field."""
        self._zeroed.pop(field.data_ptr(), None)
        _l.check(self._lib.lbmi_hydro_field_dirty(self._h, _ptr(field)))

    def field_halo_n(self, data, nswap):
        """field_halo with a swap of nswap layers (phi: 2)."""
        nel = 1 if data.dim() == 3 else data.shape[0]
        _l.check(self._lib.lbmi_field_halo_n(self._h, nel, nswap, _ptr(data)))

    def fe_scheme_set(self, grad_npt=7, advection_order=1):
        """fd_gradient_calculation 3d_7pt_fluid | 3d_27pt_fluid and
        fd_advection_scheme_order 1..4 (advection_order_set)."""
        _l.check(self._lib.lbmi_fe_scheme_set(self._h, int(grad_npt),
                                              int(advection_order)))

    def field_grad_7pt(self, phi, grad, delsq):
        """field_grad_compute (grad_3d_7pt_fluid)."""
        _l.check(self._lib.lbmi_field_grad_7pt(self._h, _ptr(phi), _ptr(grad),
                                               _ptr(delsq)))

    def field_grad_27pt(self, phi, grad, delsq):
        """field_grad_compute (grad_3d_27pt_fluid)."""
        _l.check(self._lib.lbmi_field_grad_27pt(self._h, _ptr(phi), _ptr(grad),
                                                _ptr(delsq)))

    def field_grad(self, phi, grad, delsq):
        """field_grad_compute with the stencil of fe_scheme_set."""
        _l.check(self._lib.lbmi_field_grad(self._h, _ptr(phi), _ptr(grad),
                                           _ptr(delsq)))

    def symmetric_force(self, a, b, kappa, phi, force, grad=None, delsq=None):
        """phi_force_calculation (symmetric, stress divergence): force += F."""
        _l.check(self._lib.lbmi_symmetric_force(
            self._h, a, b, kappa, _ptr(phi), _ptr(grad), _ptr(delsq),
            _ptr(force)))

    def cahn_hilliard(self, a, b, kappa, mobility, phi, u, phi_out, delsq=None):
        """phi_cahn_hilliard (symmetric): phi_out <- step(phi)."""
        _l.check(self._lib.lbmi_cahn_hilliard(
            self._h, a, b, kappa, mobility, _ptr(phi), _ptr(delsq), _ptr(u),
            _ptr(phi_out)))

    def symmetric_step(self, a, b, kappa, mobility, phi, u, force, phi_out,
                       accumulate=True):
        """symmetric_force + cahn_hilliard (both from phi) in one kernel;
        accumulate=False overwrites the force (absorbs hydro_f_zero)."""
        _l.check(self._lib.lbmi_symmetric_step(
            self._h, a, b, kappa, mobility, _ptr(phi), _ptr(u), _ptr(force),
            _ptr(phi_out), 1 if accumulate else 0))

    def symmetric_step_periodic(self, a, b, kappa, mobility, phi, u, force,
                                phi_out, accumulate=True):
        """symmetric_step without halo swaps of phi and u: the kernel wraps
        the periodic box by index (one rank)."""
        _l.check(self._lib.lbmi_symmetric_step_periodic(
            self._h, a, b, kappa, mobility, _ptr(phi), _ptr(u), _ptr(force),
            _ptr(phi_out), 1 if accumulate else 0))

    def symmetric_lb_step(self, hydro, u_prev, a, b, kappa, mobility, phi, phi_out):
        """lbmi_symmetric_lb_step: one whole step of the binary fluid
        (force, Cahn-Hilliard, lb_collide, lb_halo, lb_propagation); in the
        steady state of FUSED one kernel. u_prev: u of the previous
        collision; hydro.u (another tensor) receives the new one."""
        self._zeros_still_hold()
        h = hydro.ptrs()
        _l.check(self._lib.lbmi_symmetric_lb_step(
            self._h, ctypes.byref(h), _ptr(u_prev), a, b, kappa, mobility,
            _ptr(phi), _ptr(phi_out)))

    def symmetric_lb_collide(self, hydro, u_prev, a, b, kappa, mobility, phi, phi_out):
        """lbmi_symmetric_lb_collide: the same up to and including lb_collide;
        lb_halo and lb_propagation are the caller's."""
        self._zeros_still_hold()
        h = hydro.ptrs()
        _l.check(self._lib.lbmi_symmetric_lb_collide(
            self._h, ctypes.byref(h), _ptr(u_prev), a, b, kappa, mobility,
            _ptr(phi), _ptr(phi_out)))

    def symmetric_step_grad(self, a, b, kappa, mobility, phi, grad, delsq, u,
                            force, phi_out, accumulate=True):
        """symmetric_step with the gradients read from the arrays of
        field_grad instead of re-evaluated from phi."""
        _l.check(self._lib.lbmi_symmetric_step_grad(
            self._h, a, b, kappa, mobility, _ptr(phi), _ptr(grad), _ptr(delsq),
            _ptr(u), _ptr(force), _ptr(phi_out), 1 if accumulate else 0))

    def lb_io_aggr_pack(self):
        """lb_io_aggr_pack (model.c:1479): the binary record stream as a
        host array (nx, ny, nz, ndist*nvel); the record of a site is [n][p]."""
        torch = _torch()
        rec = torch.empty(self.nlocal + (self.ndist * self.nvel,), dtype=torch.float64,
                          device=self.device)
        torch.cuda.synchronize(self.device)
        _l.check(self._lib.lbmi_lb_records_pack(self._h, _ptr(rec)))
        self.synchronize()
        return rec.cpu().numpy()

    def lb_io_aggr_unpack(self, records):
        """lb_io_aggr_unpack (model.c:1520): restore the interior of f from
        a record stream."""
        torch = _torch()
        rec = torch.from_numpy(np.ascontiguousarray(records, dtype=np.float64))
        assert tuple(rec.shape) == self.nlocal + (self.ndist * self.nvel,)
        rec = rec.to(self.device)
        torch.cuda.synchronize(self.device)
        _l.check(self._lib.lbmi_lb_records_unpack(self._h, _ptr(rec)))
        self.synchronize()

    # -- streams, timing, communicator -------------------------------------

    def synchronize(self):
        _l.check(self._lib.lbmi_synchronize(self._h))

    # -- walls (wall.c) -----------------------------------------------------

    def wall_map(self, isboundary, status):
        """wall_init_map: MAP_BOUNDARY into the device map (int8, nall)."""
        b = (ctypes.c_int * 3)(*[int(x) for x in isboundary])
        _l.check(self._lib.lbmi_wall_map(self._h, b, _ptr(status)))

    def wall_links_build(self, status, isboundary):
        """wall_init_boundaries + wall_init_uw; returns the number of links."""
        b = (ctypes.c_int * 3)(*[int(x) for x in isboundary])
        n = ctypes.c_int(0)
        _l.check(self._lib.lbmi_wall_links_build(self._h, _ptr(status), b,
                                                 ctypes.byref(n)))
        self.nlink = n.value
        return n.value

    def wall_links(self):
        """(linki, linkj, linkp, linku) as the reference holds them."""
        arr = [np.zeros(max(self.nlink, 1), dtype=np.int32) for _ in range(4)]
        _l.check(self._lib.lbmi_wall_links(
            self._h, *[a.ctypes.data_as(ctypes.c_void_p) for a in arr]))
        return tuple(a[:self.nlink] for a in arr)

    def wall_links_set(self, linki, linkj, linkp, linku):
        """Links made by the caller (host arrays, wall.c:399-451): checked
        record by record on the host, copied, owned by the handle."""
        arr = [np.ascontiguousarray(a, dtype=np.int32)
               for a in (linki, linkj, linkp, linku)]
        n = len(arr[0])
        assert all(len(a) == n for a in arr)
        _l.check(self._lib.lbmi_wall_links_set(
            self._h, n, *[a.ctypes.data_as(ctypes.c_void_p) for a in arr]))
        self.nlink = n

    def wall_slip_links_set(self, linkk, linkq, links, stab):
        """Slip records made by the caller, in the reference's types (int,
        int8, int8) and its table of 19 fractions."""
        k = np.ascontiguousarray(linkk, dtype=np.int32)
        q = np.ascontiguousarray(linkq, dtype=np.int8)
        s = np.ascontiguousarray(links, dtype=np.int8)
        assert len(k) == len(q) == len(s) == self.nlink
        t = (ctypes.c_double * 19)(*[float(x) for x in stab])
        _l.check(self._lib.lbmi_wall_slip_links_set(
            self._h, k.ctypes.data_as(ctypes.c_void_p),
            q.ctypes.data_as(ctypes.c_void_p),
            s.ctypes.data_as(ctypes.c_void_p), t))

    def wall_fnet_bind(self, fnet):
        """Momentum of lbmi_wall_bbl into 3 doubles on the device owned by
        the caller (None: the handle's accumulator)."""
        self._wall_fnet = fnet
        _l.check(self._lib.lbmi_wall_fnet_bind(
            self._h, None if fnet is None else _ptr(fnet)))

    def wall_bbl_arrays(self, linki, linkj, linkp, linku, ubot, utop, fnet):
        """wall_bbl on DEVICE link arrays the caller owns (int32 tensors)."""
        ub = (ctypes.c_double * 3)(*[float(x) for x in ubot])
        ut = (ctypes.c_double * 3)(*[float(x) for x in utop])
        _l.check(self._lib.lbmi_wall_bbl_arrays(
            self._h, int(linki.numel()), _ptr(linki), _ptr(linkj), _ptr(linkp),
            _ptr(linku), ub, ut, _ptr(fnet)))

    def mode_set(self, mode):
        """lbmi_lb_mode_set: flush, then continue in another execution mode."""
        _l.check(self._lib.lbmi_lb_mode_set(self._h, int(mode)))
        self.mode = int(mode)

    def wall_velocity_set(self, ubot, utop):
        ub = (ctypes.c_double * 3)(*[float(x) for x in ubot])
        ut = (ctypes.c_double * 3)(*[float(x) for x in utop])
        _l.check(self._lib.lbmi_wall_velocity_set(self._h, ub, ut))

    def wall_bbl(self):
        """wall_bbl: between lb_halo and lb_propagation (EAGER)."""
        _l.check(self._lib.lbmi_wall_bbl(self._h))

    def wall_status_set(self, status):
        """The device map wall_bbl tests for MAP_COLLOID (None: no test);
        the tensor is kept alive by the handle."""
        self._wall_status = status
        _l.check(self._lib.lbmi_wall_status_set(
            self._h, None if status is None else _ptr(status)))

    def wall_slip_set(self, status, sbot, stop):
        """wall_slip + wall_init_boundaries_slip for the links built: slip
        fractions of the bottom / top wall of each direction (all zero: off).
        wall_bbl then runs the reference's slip kernel."""
        sb = (ctypes.c_double * 3)(*[float(x) for x in sbot])
        st = (ctypes.c_double * 3)(*[float(x) for x in stop])
        _l.check(self._lib.lbmi_wall_slip_set(self._h, _ptr(status), sb, st))

    def wall_slip_links(self):
        """(linkk, linkq, links): partner site, partner direction, index of s."""
        arr = [np.zeros(max(self.nlink, 1), dtype=np.int32) for _ in range(3)]
        _l.check(self._lib.lbmi_wall_slip_links(
            self._h, *[a.ctypes.data_as(ctypes.c_void_p) for a in arr]))
        return tuple(a[:self.nlink] for a in arr)

    def wall_momentum(self):
        out = (ctypes.c_double * 3)()
        _l.check(self._lib.lbmi_wall_momentum(self._h, out))
        return np.array(out[:])

    def phi_to_field(self, phi):
        """phi_lb_to_field (ndist = 2): phi = sum_p g_p."""
        _l.check(self._lib.lbmi_lb_phi_to_field(self._h, _ptr(phi)))

    def lb_collide_binary(self, hydro, a, b, kappa, mobility, phi, grad, delsq,
                          grad_stride=0):
        """lb_collide with ndist = 2 (lb_collision_binary). grad_stride:
        lbmi_fe_symm_t::nsite, the distance between the components of grad
        when it is not the lattice's nsite."""
        fe = _l.FeSymm()
        fe.a, fe.b, fe.kappa, fe.mobility = a, b, kappa, mobility
        fe.phi, fe.grad, fe.delsq = _ptr(phi), _ptr(grad), _ptr(delsq)
        fe.nsite = int(grad_stride)
        if hydro is None:
            _l.check(self._lib.lbmi_lb_collide_binary(self._h, None,
                                                      ctypes.byref(fe)))
        else:
            h = hydro.ptrs()
            _l.check(self._lib.lbmi_lb_collide_binary(self._h, ctypes.byref(h),
                                                      ctypes.byref(fe)))

    def lb_collide_fe(self, hydro, a, b, kappa, phi, grad, delsq, grad_stride=0):
        """lb_collide with fe->use_stress_relaxation (symmetric free energy)."""
        fe = _l.FeSymm()
        fe.a, fe.b, fe.kappa, fe.mobility = a, b, kappa, 0.0
        fe.phi, fe.grad, fe.delsq = _ptr(phi), _ptr(grad), _ptr(delsq)
        fe.nsite = int(grad_stride)
        h = hydro.ptrs()
        _l.check(self._lib.lbmi_lb_collide_fe(self._h, ctypes.byref(h),
                                              ctypes.byref(fe)))

    def io_format_set(self, ascii=False, single=False):
        """lbmi_io_format_set: binary (default) or text records
        (distribution_io_format ascii) for lb_io_write / lb_io_read;
        single: the old-style files of a run that names no i/o mode
        (io_harness.c: dist-%8.8d.001-001 and dist.001-001.meta)."""
        _l.check(self._lib.lbmi_io_format_set(
            self._h, (IO_ASCII if ascii else 0) | (IO_SINGLE if single else 0)))

    def io_file_set(self, nfile=1, index=0, file_nx=0, file_x0=0, periodic=(1, 1, 1)):
        """lbmi_io_file_set: this rank's file of the i/o grid {nfile, 1, 1}
        (io_subfile_create, io_subfile.c:49-91) and the periodicity the
        metadata prints; nfile=None: back to one periodic file."""
        if nfile is None:
            _l.check(self._lib.lbmi_io_file_set(self._h, None))
            return
        _l.check(self._lib.lbmi_io_file_set(self._h, ctypes.byref(
            _io_file(nfile, index, file_nx, file_x0, periodic))))

    def lb_io_write(self, directory, timestep, ntotal_x=None, offset_x=0):
        """lb_io_write (model.c:1568): dist-metadata.001-001 and
        dist-<timestep>.001-001 in `directory` (MPI-IO mode, one file)."""
        nx = self.nlocal[0] if ntotal_x is None else ntotal_x
        _l.check(self._lib.lbmi_lb_io_write(self._h, str(directory).encode(),
                                            int(timestep), nx, offset_x))

    def lb_io_read(self, directory, timestep, ntotal_x=None, offset_x=0):
        """lb_io_read (model.c:1622): replace the state by the records of
        dist-<timestep>.001-001."""
        nx = self.nlocal[0] if ntotal_x is None else ntotal_x
        _l.check(self._lib.lbmi_lb_io_read(self._h, str(directory).encode(),
                                           int(timestep), nx, offset_x))

    def state(self):
        """(halo pending, propagation pending, order of f) -- lbmi_lb_state."""
        st = (ctypes.c_int * 3)()
        _l.check(self._lib.lbmi_lb_state(self._h, st))
        return tuple(st)

    def tune(self, key, value):
        _l.check(self._lib.lbmi_tune(self._h, key.encode(), int(value)))

    def timing(self, on=True):
        """on: False/0 off, True/1 every launch, k every k-th launch."""
        _l.check(self._lib.lbmi_timing(self._h, int(on)))

    def timing_read(self):
        ms = ctypes.c_double()
        n = ctypes.c_int()
        _l.check(self._lib.lbmi_timing_read(self._h, ctypes.byref(ms),
                                            ctypes.byref(n)))
        return ms.value, n.value

    @staticmethod
    def comm_unique_id():
        buf = ctypes.create_string_buffer(_l.UNIQUE_ID_BYTES)
        _l.check(_l.library().lbmi_comm_unique_id(buf))
        return buf.raw

    def comm_init_ring(self, ring):
        """Join an in-process ring (lbmi_comm_init_ring) instead of RCCL."""
        _l.check(self._lib.lbmi_comm_init_ring(self._h, ring._r))

    def comm_info(self):
        """(ranks of the ring, own rank, transport: 0 none, 1 RCCL, 2 in-process)"""
        n, r, t = ctypes.c_int(0), ctypes.c_int(-1), ctypes.c_int(0)
        _l.check(self._lib.lbmi_comm_info(self._h, ctypes.byref(n),
                                          ctypes.byref(r), ctypes.byref(t)))
        return n.value, r.value, t.value

    def timing_read_detail(self):
        """Average ms of (interior launch, exchange, boundary launch) over the
        sampled slab steps, and the number of samples."""
        ms = (ctypes.c_double * 3)()
        n = ctypes.c_int(0)
        _l.check(self._lib.lbmi_timing_read_detail(self._h, ms, ctypes.byref(n)))
        return [ms[0], ms[1], ms[2]], n.value

    def comm_init(self, unique_id):
        buf = ctypes.create_string_buffer(bytes(unique_id), _l.UNIQUE_ID_BYTES)
        _l.check(self._lib.lbmi_comm_init(self._h, buf))


def io_metadata_write(directory, stub, nel, ntotal):
    """io_metadata_write (io_metadata.c): <stub>-metadata.001-001. Host only."""
    n = (ctypes.c_int * 3)(*[int(x) for x in ntotal])
    _l.check(_l.library().lbmi_io_metadata_write(str(directory).encode(),
                                                 stub.encode(), int(nel), n))


IO_ASCII, IO_SINGLE = 1, 2           # lbmi.h: LBMI_IO_ASCII, LBMI_IO_SINGLE


def io_metadata_write_fmt(directory, stub, nvel, ndist, ntotal, ascii=False,
                          single=False):
    """lbmi_io_metadata_write_fmt: the metadata of records of ndist*nvel values
    in binary or text (distribution_io_format ascii) form, of the MPI-IO or
    the single mode. Host only."""
    lib = _l.library()
    nt = (ctypes.c_int * 3)(*[int(v) for v in ntotal])
    _l.check(lib.lbmi_io_metadata_write_fmt(str(directory).encode(), stub.encode(),
                                            int(nvel), int(ndist), nt,
                                            (IO_ASCII if ascii else 0)
                                            | (IO_SINGLE if single else 0)))


def _io_file(nfile, index, file_nx, file_x0, periodic):
    f = _l.IoFile()
    f.nfile, f.index, f.file_nx, f.file_x0 = int(nfile), int(index), int(file_nx), int(file_x0)
    for ia in range(3):
        f.periodic[ia] = int(periodic[ia])
    return f


def io_metadata_write_file(directory, stub, nvel, ndist, ntotal, nfile=1, index=0,
                           file_nx=None, file_x0=0, periodic=(1, 1, 1), ascii=False,
                           single=False):
    """lbmi_io_metadata_write_file: <stub>-metadata.<1+index>-<nfile> of one
    file of the i/o grid {nfile, 1, 1}. Host only."""
    nt = (ctypes.c_int * 3)(*[int(v) for v in ntotal])
    f = _io_file(nfile, index, ntotal[0] if file_nx is None else file_nx, file_x0, periodic)
    _l.check(_l.library().lbmi_io_metadata_write_file(
        str(directory).encode(), stub.encode(), int(nvel), int(ndist), nt,
        (IO_ASCII if ascii else 0) | (IO_SINGLE if single else 0), ctypes.byref(f)))


def io_single_metadata_write(directory, stub, nvel, ndist, ntotal, cartdim=0,
                             nslab=None):
    """lbmi_io_single_metadata_write: <stub>.001-001.meta of the old-style
    i/o (io_write_metadata_file, io_harness.c:369-466). Host only."""
    nt = (ctypes.c_int * 3)(*[int(v) for v in ntotal])
    sz = 1 if nslab is None else len(nslab)
    ns = None if nslab is None else (ctypes.c_int * sz)(*[int(v) for v in nslab])
    _l.check(_l.library().lbmi_io_single_metadata_write(
        str(directory).encode(), stub.encode(), int(nvel), int(ndist), nt,
        int(cartdim), sz, ns))


def io_filename(directory, stub, timestep, single=False):
    """io_subfile_name: <stub>-%9.9d.001-001 (single: the old style's
    <stub>-%8.8d.001-001). Host only."""
    buf = ctypes.create_string_buffer(1024)
    _l.check(_l.library().lbmi_io_filename_fmt(str(directory).encode(),
                                               stub.encode(), int(timestep),
                                               IO_SINGLE if single else 0, buf, 1024))
    return buf.value.decode()


class Ring:
    """lbmi_ring_t: cartsz handles of ONE process on one device as the ranks
    of the X ring (one thread each; LB(..., own_stream=True))."""

    def __init__(self, nranks):
        self._lib = _l.library()
        self._r = ctypes.c_void_p()
        _l.check(self._lib.lbmi_ring_create(int(nranks), ctypes.byref(self._r)))
        self.nranks = int(nranks)

    def abort(self):
        """A rank's driver has failed: release the others from their waits."""
        if self._r:
            self._lib.lbmi_ring_abort(self._r)

    def free(self):
        if self._r:
            _l.check(self._lib.lbmi_ring_free(self._r))
            self._r = ctypes.c_void_p()


def x_schedule(nvel, nlocal, nhalo, cartsz, cartrank, scheme=HALO_REDUCED,
               packed=True, cartdim=0, cartgrid=None, cartcoords=None, dim=0):
    """lbmi_x_schedule: the point-to-point operations of one X exchange of a
    rank, in issue order, as dicts (kind 'send'|'recv', peer, buffer 'sendlo'|
    'sendhi'|'recvlo'|'recvhi'|'data', offset, count). Pure host: no GPU."""
    lib = _l.library()
    opts = _l.Options()
    _l.check(lib.lbmi_options_default(ctypes.byref(opts)))
    opts.nvel = nvel
    opts.nlocal[:] = list(nlocal)
    opts.nhalo = nhalo
    opts.cartsz = cartsz
    opts.cartrank = cartrank
    opts.cartdim = cartdim
    ops = (_l.XOp * 128)()
    n = ctypes.c_int(0)
    if cartgrid is not None:
        _cart_options(opts, cartgrid, cartcoords)
        _l.check(lib.lbmi_x_schedule_dim(ctypes.byref(opts), int(dim), int(scheme),
                                         int(bool(packed)),
                                         ctypes.cast(ops, ctypes.c_void_p), 128,
                                         ctypes.byref(n)))
    else:
        _l.check(lib.lbmi_x_schedule(ctypes.byref(opts), int(scheme), int(bool(packed)),
                                     ctypes.cast(ops, ctypes.c_void_p), 128,
                                     ctypes.byref(n)))
    names = ("sendlo", "sendhi", "recvlo", "recvhi", "data")
    return [{"kind": "send" if ops[k].kind == 0 else "recv", "peer": ops[k].peer,
             "buffer": names[ops[k].buffer], "offset": int(ops[k].offset),
             "count": int(ops[k].count)} for k in range(n.value)]


CART_GENERAL = 3                      # lbmi.h: LBMI_CART_GENERAL


def _cart_options(opts, cartgrid, cartcoords):
    g = [int(v) for v in cartgrid]
    c = [int(v) for v in cartcoords]
    opts.cartdim = CART_GENERAL
    opts.cartgrid[:] = g
    opts.cartcoords[:] = c
    opts.cartsz = g[0] * g[1] * g[2]
    opts.cartrank = (c[0] * g[1] + c[1]) * g[2] + c[2]


def model(nvel):
    """cv, wv, na, ma as the reference builds them (lb_d3q19.c, lb_d3q27.c)."""
    lib = _l.library()
    cv = np.zeros((nvel, 3), dtype=np.int8)
    wv = np.zeros(nvel)
    na = np.zeros(nvel)
    ma = np.zeros((nvel, nvel))
    _l.check(lib.lbmi_model(nvel, cv.ctypes.data_as(ctypes.c_void_p),
                            wv.ctypes.data_as(ctypes.c_void_p),
                            na.ctypes.data_as(ctypes.c_void_p),
                            ma.ctypes.data_as(ctypes.c_void_p)))
    return {"nvel": nvel, "cv": cv, "wv": wv, "na": na, "ma": ma}
