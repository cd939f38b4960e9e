"""1-d slab decomposition: the reference's Cartesian decomposition
(coords.c:146-215, `grid N_1_1`, `1_N_1` or `1_1_N`, coords_rt.c:46-47)
restricted to one axis. X (the default) is the slowest index of the
reference's memory order, so a slab's boundary planes are contiguous per
population; slabs along Y or Z have planes that are gathered and scattered.
"""


class SlabDecomposition:

    def __init__(self, ntotal, cartsz, cartrank, nhalo=1, dim=0):
        if cartsz < 1 or not (0 <= cartrank < cartsz):
            raise ValueError("cartsz/cartrank")
        if dim not in (0, 1, 2):
            raise ValueError("dim")
        if ntotal[dim] % cartsz != 0:
            # the reference requires an exact division (coords.c:327-338)
            raise ValueError("ntotal[%d] = %d not divisible by %d ranks"
                             % (dim, ntotal[dim], cartsz))
        self.ntotal = tuple(ntotal)
        self.cartsz = cartsz
        self.cartrank = cartrank
        self.nhalo = nhalo
        self.dim = dim
        nlocal = list(ntotal)
        nlocal[dim] = ntotal[dim] // cartsz
        self.nlocal = tuple(nlocal)
        noffset = [0, 0, 0]
        noffset[dim] = nlocal[dim] * cartrank
        self.noffset = tuple(noffset)

    @property
    def nall(self):
        return tuple(n + 2 * self.nhalo for n in self.nlocal)

    @property
    def prev(self):
        """cs_cart_neighb(cs, BACKWARD, X) on a periodic ring."""
        return (self.cartrank - 1) % self.cartsz

    @property
    def next(self):
        """cs_cart_neighb(cs, FORWARD, X)."""
        return (self.cartrank + 1) % self.cartsz

    def plane_doubles(self, ncomp):
        """Message length of one X face with ncomp components
        (hsz[X]*nfel, halo_swap.c:763)."""
        nall = self.nall
        return nall[0] * nall[1] * nall[2] // nall[self.dim] * ncomp

    def local_slice(self):
        """Slice of the global interior range (along dim) owned by this rank."""
        d = self.dim
        return slice(self.noffset[d], self.noffset[d] + self.nlocal[d])

    @staticmethod
    def reduced_populations(cv, axis=0):
        """Populations needed in the (low, high) halo plane of `axis`:
        c_axis = +1 are pulled across the low face, -1 across the high face
        (model.c:1192-1219: cv.m == |m|^2)."""
        lo = [p for p in range(len(cv)) if cv[p][axis] == 1]
        hi = [p for p in range(len(cv)) if cv[p][axis] == -1]
        return lo, hi
