"""1-d slab decomposition along X: the reference's Cartesian decomposition
(coords.c:146-215, `grid N_1_1`) restricted to one axis. X is the slowest
index of the reference's memory order, so a slab's boundary planes are
contiguous per population.
"""


class SlabDecomposition:

    def __init__(self, ntotal, cartsz, cartrank, nhalo=1):
        if cartsz < 1 or not (0 <= cartrank < cartsz):
            raise ValueError("cartsz/cartrank")
        if ntotal[0] % cartsz != 0:
            # the reference requires an exact division (coords.c:327-338)
            raise ValueError("ntotal[X] = %d not divisible by %d ranks"
                             % (ntotal[0], cartsz))
        self.ntotal = tuple(ntotal)
        self.cartsz = cartsz
        self.cartrank = cartrank
        self.nhalo = nhalo
        self.nlocal = (ntotal[0] // cartsz, ntotal[1], ntotal[2])
        self.noffset = (self.nlocal[0] * cartrank, 0, 0)

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
        return nall[1] * nall[2] * ncomp

    def local_slice(self):
        """Slice of the global interior x-range owned by this rank."""
        return slice(self.noffset[0], self.noffset[0] + self.nlocal[0])

    @staticmethod
    def reduced_populations(cv, axis=0):
        """Populations needed in the (low, high) halo plane of `axis`:
        c_axis = +1 are pulled across the low face, -1 across the high face
        (model.c:1192-1219: cv.m == |m|^2)."""
        lo = [p for p in range(len(cv)) if cv[p][axis] == 1]
        hi = [p for p in range(len(cv)) if cv[p][axis] == -1]
        return lo, hi
