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


class CartDecomposition:
    """A Cartesian decomposition of any shape (the reference's default for N
    ranks is MPI_Dims_create's, e.g. 2_2_2 for eight; coords.c:520-560), even
    parts (coords.c:327-338). Ranks as MPI_Cart_create numbers them without
    reordering: rank = (cx*gy + cy)*gz + cz."""

    def __init__(self, ntotal, grid, rank, nhalo=1):
        grid = tuple(int(g) for g in grid)
        size = grid[0] * grid[1] * grid[2]
        if not (0 <= rank < size):
            raise ValueError("rank")
        for d in range(3):
            if ntotal[d] % grid[d] != 0:
                raise ValueError("ntotal[%d] = %d not divisible by %d ranks"
                                 % (d, ntotal[d], grid[d]))
        self.ntotal = tuple(ntotal)
        self.grid = grid
        self.rank = rank
        self.size = size
        self.nhalo = nhalo
        self.coords = (rank // (grid[1] * grid[2]), (rank // grid[2]) % grid[1], rank % grid[2])
        self.nlocal = tuple(ntotal[d] // grid[d] for d in range(3))
        self.noffset = tuple(self.nlocal[d] * self.coords[d] for d in range(3))

    @property
    def nall(self):
        return tuple(n + 2 * self.nhalo for n in self.nlocal)

    def rank_of(self, coords):
        g = self.grid
        c = [coords[d] % g[d] for d in range(3)]
        return (c[0] * g[1] + c[1]) * g[2] + c[2]

    def neighbours(self, dim):
        """(rank below, rank above) along dim, periodic."""
        lo = list(self.coords)
        hi = list(self.coords)
        lo[dim] -= 1
        hi[dim] += 1
        return self.rank_of(lo), self.rank_of(hi)

    def local_block(self, h=None):
        """Slices of a global array WITH halo (nhalo = h) that hold this
        rank's block with its halo."""
        h = self.nhalo if h is None else h
        return tuple(slice(self.noffset[d], self.noffset[d] + self.nlocal[d] + 2 * h)
                     for d in range(3))
