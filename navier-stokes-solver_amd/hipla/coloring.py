"""Multicolour ordering of block-Jacobi blocks for the parallel block Gauss-Seidel sweep
(SURVEY.md section 8f row N1).

Two blocks are *coupled* when the matrix has an entry between their dofs; blocks of one colour
must be pairwise uncoupled so that a whole colour can be updated in one kernel launch.  The
colouring is a sequence of maximal independent sets (Luby's algorithm with fixed pseudo-random
priorities), fully vectorised with scipy/numpy so that millions of blocks colour in seconds.
Host-side set-up code, deterministic for a given seed."""

import numpy as np
import scipy.sparse as sp


def block_graph(csr, idx):
    """Adjacency (nblocks x nblocks, CSR, no diagonal) of the blocks `idx` (bs, nblocks; -1 pad)
    induced by the sparsity of `csr`."""
    n = csr.shape[0]
    bs, nb = idx.shape
    live = idx >= 0
    dofs = idx[live]
    blocks = np.broadcast_to(np.arange(nb, dtype=np.int64), idx.shape)[live]
    member = sp.csr_matrix((np.ones(dofs.size, dtype=np.float64), (dofs, blocks)), shape=(n, nb))
    pattern = sp.csr_matrix((np.ones(csr.nnz), csr.indices, csr.indptr), shape=csr.shape)
    g = (member.T @ pattern @ member).tocsr()
    g = g + g.T
    g.setdiag(0)
    g.eliminate_zeros()
    g.sort_indices()
    return g


def _neighbour_max(g, values, fill):
    """max over graph neighbours of `values` (fill for isolated nodes)."""
    out = np.full(g.shape[0], fill, dtype=values.dtype)
    deg = np.diff(g.indptr)
    has = deg > 0
    if g.nnz:
        red = np.maximum.reduceat(values[g.indices], g.indptr[:-1][has])
        out[has] = red
    return out


def color_blocks(g, seed=0):
    """Colours (int32 per block), each colour a maximal independent set of the remaining graph."""
    nb = g.shape[0]
    rng = np.random.default_rng(seed)
    priority = rng.permutation(nb).astype(np.int64) + 1          # distinct, > 0
    colors = -np.ones(nb, dtype=np.int32)
    c = 0
    while np.any(colors < 0):
        cand = colors < 0
        chosen = np.zeros(nb, dtype=bool)
        while cand.any():
            pri = np.where(cand, priority, 0)
            winners = cand & (pri > _neighbour_max(g, pri, 0))
            chosen |= winners
            hit = _neighbour_max(g, winners.astype(np.int64), 0) > 0
            cand &= ~winners & ~hit
        colors[chosen] = c
        c += 1
    return colors


def color_blocks_greedy(g):
    """First-fit colours in block order: block i takes the smallest colour none of its already coloured
    neighbours (j < i) has -- on grid-like block graphs the parity colouring (2 - 4 balanced colours).  The host
    twin of `nss_graph_color_greedy` (same colours); a plain loop: set-up of small systems and tests only."""
    nb = g.shape[0]
    indptr, indices = g.indptr, g.indices
    colors = -np.ones(nb, dtype=np.int32)
    for i in range(nb):
        nbr = indices[indptr[i]:indptr[i + 1]]
        taken = set(int(c) for c in colors[nbr[nbr < i]])
        c = 0
        while c in taken:
            c += 1
        colors[i] = c
    return colors


def colour_major_order(colors):
    """Permutation that sorts blocks by colour (stable) and the colour offsets."""
    order = np.argsort(colors, kind="stable")
    ptr = np.concatenate([[0], np.cumsum(np.bincount(colors, minlength=int(colors.max()) + 1))]).astype(np.int32)
    return order, ptr


def check_coloring(g, colors):
    coo = g.tocoo()
    return not np.any(colors[coo.row] == colors[coo.col])


def colour_permutation(csr, idx, seed=0):
    """Colour-major re-ordering of the dofs for a block Gauss-Seidel sweep.

    Returns ``(perm, new_idx, new_colors)``: ``perm[k]`` = old dof at new position k (blocks
    colour by colour, the dofs of a block adjacent, dofs in no block last), the block table in
    the new numbering (contiguous ranges) and the colour of each block (ascending).  Apply with
    ``A' = A[perm][:, perm]``, ``B' = B[:, perm]``, ``f' = f[perm]``; ``u = u'[inverse]``."""
    colors = color_blocks(block_graph(csr, idx), seed)
    order, _ = colour_major_order(colors)
    table = idx[:, order]
    live = table.T >= 0
    in_blocks = table.T[live].astype(np.int64)
    rest = np.setdiff1d(np.arange(csr.shape[0], dtype=np.int64), in_blocks, assume_unique=False)
    perm = np.concatenate([in_blocks, rest])
    new_idx = -np.ones(table.T.shape, dtype=np.int32)
    new_idx[live] = np.arange(in_blocks.size, dtype=np.int32)
    return perm, np.ascontiguousarray(new_idx.T), colors[order]
