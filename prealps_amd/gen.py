"""Synthetic SPD test matrices and geometric partitions, generated from their
definition (no files, no downloads): the 7-point Poisson operator of SURVEY
Appendix A.2 and sub-box partitions of the grid (the stand-in for METIS k-way
on structured problems)."""
import numpy as np


def poisson3d_csr(n):
    """A = T(x)I(x)I + I(x)T(x)I + I(x)I(x)T, T = tridiag(-1, 2, -1); row (i*n+j)*n+k.
    Returns int32 rowptr, int32 colind, float64 val with sorted columns."""
    N = n ** 3
    idx = np.arange(N, dtype=np.int64)
    i, j, k = idx // (n * n), (idx // n) % n, idx % n
    cols = [idx - n * n, idx - n, idx - 1, idx, idx + 1, idx + n, idx + n * n]
    ok = [i > 0, j > 0, k > 0, np.ones(N, bool), k < n - 1, j < n - 1, i < n - 1]
    vals = [-1.0, -1.0, -1.0, 6.0, -1.0, -1.0, -1.0]
    C = np.stack(cols, axis=1)
    M = np.stack(ok, axis=1)
    V = np.broadcast_to(np.array(vals), (N, 7))
    counts = M.sum(axis=1)
    rowptr = np.zeros(N + 1, dtype=np.int64)
    rowptr[1:] = np.cumsum(counts)
    return rowptr.astype(np.int32), C[M].astype(np.int32), np.ascontiguousarray(V[M], dtype=np.float64)


def box_partition(n, box):
    """Part id of every node of an n^3 grid cut into boxes of `box` = (bi, bj, bk)
    nodes (the last box of a direction may be smaller).  Parts are numbered
    box-lexicographically, so consecutive parts are neighbours."""
    bi, bj, bk = box
    idx = np.arange(n ** 3, dtype=np.int64)
    i, j, k = idx // (n * n), (idx // n) % n, idx % n
    ni, nj, nk = -(-n // bi), -(-n // bj), -(-n // bk)
    part = ((i // bi) * nj + (j // bj)) * nk + (k // bk)
    return part.astype(np.int32), ni * nj * nk


def q1_elasticity_element(nu=0.25, E=1.0):
    """24 x 24 stiffness of the trilinear (Q1) unit-cube element of isotropic linear
    elasticity, 2x2x2 Gauss quadrature.  Node order (1,0,0)-first counter-clockwise on the
    bottom face, then the top face, dof = 3*node + component: the ordering the reference's
    assembly uses (examples/test_ecg_petsc_ela.c:294-301).  The reference hard-codes its own
    24 x 24 table (reference_q1_element below, the default of elasticity3d_csr); this one is
    derived from the definition, so values differ while the sparsity, block structure and
    SPD-ness are the same."""
    lam = E * nu / ((1 + nu) * (1 - 2 * nu))
    mu = E / (2 * (1 + nu))
    D = np.zeros((6, 6))
    D[:3, :3] = lam
    D[np.arange(3), np.arange(3)] += 2 * mu
    D[np.arange(3, 6), np.arange(3, 6)] = mu
    corners = np.array([(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)], float)
    g = 0.5 + np.array([-1.0, 1.0]) / (2 * np.sqrt(3.0))
    K = np.zeros((24, 24))
    for x in g:
        for y in g:
            for z in g:
                p = np.array([x, y, z])
                f = np.where(corners == 1, p, 1 - p)           # 1-D factors per node and axis
                df = np.where(corners == 1, 1.0, -1.0)
                dN = np.stack([df[:, 0] * f[:, 1] * f[:, 2], f[:, 0] * df[:, 1] * f[:, 2],
                               f[:, 0] * f[:, 1] * df[:, 2]], axis=1)   # (8, 3)
                B = np.zeros((6, 24))
                for a in range(8):
                    dx, dy, dz = dN[a]
                    B[0, 3 * a] = dx
                    B[1, 3 * a + 1] = dy
                    B[2, 3 * a + 2] = dz
                    B[3, 3 * a], B[3, 3 * a + 1] = dy, dx
                    B[4, 3 * a + 1], B[4, 3 * a + 2] = dz, dy
                    B[5, 3 * a], B[5, 3 * a + 2] = dz, dx
                K += B.T @ D @ B / 8.0
    return 0.5 * (K + K.T)


def reference_q1_element():
    """The reference's own 24 x 24 element matrix `elem_3d_elast_v_25`
    (examples/test_ecg_petsc_ela.c:65-211), exact: integer numerators over 1080."""
    from .q1_table import Q1_DENOMINATOR, Q1_NUMERATORS
    return np.array(Q1_NUMERATORS, dtype=np.float64) / float(Q1_DENOMINATOR)


def _dims(nn):
    if np.isscalar(nn):
        return int(nn), int(nn), int(nn)
    nx, ny, nz = (int(v) for v in nn)
    return nx, ny, nz


def elasticity3d_csr(nn, nu=0.25, element="reference"):
    """Q1 3-D elasticity on nn^3 nodes -- or (nx, ny, nz) nodes -- assembled the way
    examples/test_ecg_petsc_ela.c:217-347 does for one process: node id = i + nx*j + nx*ny*k,
    dof = 3*node + c, N = 3 nx ny nz; element coefficient 1 except 1e-5 / 1e5 where the element
    centre lies within 0.05 of one of the eight points (.25|.75)^3 of the unit cube (lines
    275-323, tested in the reference's order); in the k = 0 element layer the bottom-face nodes
    are decoupled and their diagonal scaled by 0.1 (DD2, lines 234-246), which keeps the matrix
    SPD without eliminating rows.  element = "reference": the reference's own element matrix
    (q1_table.py); "derived": the textbook Q1 stiffness for Poisson ratio nu by 2x2x2 Gauss
    quadrature (same sparsity and block structure, different values).  The reference only
    generates cubes; on a box the unit cube is cut into nx-1 x ny-1 x nz-1 elements and the
    (size-independent) element matrix is kept.  Returns int32 rowptr, int32 colind, float64 val."""
    nx, ny, nz = _dims(nn)
    ex, ey, ez = nx - 1, ny - 1, nz - 1
    if element == "reference":
        DD1 = reference_q1_element()
    elif element == "derived":
        DD1 = q1_elasticity_element(nu)
    else:
        raise ValueError("element must be 'reference' or 'derived'")
    DD2 = DD1.copy()
    for a in range(24):
        for b in range(24):
            if a < 12 or b < 12:
                DD2[a, b] = 0.1 * DD1[a, a] if a == b else 0.0
    # element coefficient, indexed [i, j, k]
    cx, cy, cz = ((np.arange(e) + 0.5) / e for e in (ex, ey, ez))
    X, Y, Z = np.meshgrid(cx, cy, cz, indexing="ij")      # element centres
    alpha = np.ones((ex, ey, ez))
    r = 0.05
    order = [((.25, .25, .25), (.25, .25, .75), 1e-5), ((.75, .25, .25), (.75, .25, .75), 1e5),
             ((.25, .75, .25), (.25, .75, .75), 1e-5), ((.75, .75, .25), (.75, .75, .75), 1e5)]
    for c1, c2, val in order:
        for px, py, pz in (c1, c2):
            alpha[np.sqrt((X - px) ** 2 + (Y - py) ** 2 + (Z - pz) ** 2) < r] = val
    corners = [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)]
    # node-block accumulation: blocks[i, j, k, offset(di,dj,dk), 3, 3]
    blocks = np.zeros((nx, ny, nz, 27, 3, 3))
    for a, ca in enumerate(corners):
        for b, cb in enumerate(corners):
            off = ((cb[2] - ca[2] + 1) * 3 + (cb[1] - ca[1] + 1)) * 3 + (cb[0] - ca[0] + 1)
            blk1 = DD1[3 * a:3 * a + 3, 3 * b:3 * b + 3]
            blk2 = DD2[3 * a:3 * a + 3, 3 * b:3 * b + 3]
            # element (i,j,k) contributes to node (i+ca0, j+ca1, k+ca2)
            tgt = blocks[ca[0]:ca[0] + ex, ca[1]:ca[1] + ey, ca[2]:ca[2] + ez, off]
            tgt[:, :, 1:] += alpha[:, :, 1:, None, None] * blk1
            tgt[:, :, :1] += alpha[:, :, :1, None, None] * blk2
    # CSR: node rows in id order (k slowest), neighbours in increasing node id
    I, J, K_ = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    node = (I + nx * J + nx * ny * K_)
    offs = [(di, dj, dk) for dk in (-1, 0, 1) for dj in (-1, 0, 1) for di in (-1, 0, 1)]
    exist = np.zeros((nx, ny, nz, 27), bool)
    nbr = np.zeros((nx, ny, nz, 27), np.int64)
    for o, (di, dj, dk) in enumerate(offs):
        ok = ((I + di >= 0) & (I + di < nx) & (J + dj >= 0) & (J + dj < ny) & (K_ + dk >= 0) & (K_ + dk < nz))
        exist[..., o] = ok
        nbr[..., o] = node + di + nx * dj + nx * ny * dk
    perm = np.argsort(node.ravel())                     # row order = node id
    exist = exist.reshape(-1, 27)[perm]
    nbr = nbr.reshape(-1, 27)[perm]
    blocks = blocks.reshape(-1, 27, 3, 3)[perm]
    cnt = exist.sum(axis=1)                             # neighbours per node
    nnodes = nx * ny * nz
    rowptr = np.zeros(3 * nnodes + 1, dtype=np.int64)
    rowptr[1:] = np.cumsum(np.repeat(3 * cnt, 3))
    nb_flat = nbr[exist]                                # (sum cnt,) in node-major, offset-minor order
    bl_flat = blocks[exist]                             # (sum cnt, 3, 3)
    node_of = np.repeat(np.arange(nnodes), cnt)
    first = np.zeros(nnodes + 1, dtype=np.int64)
    first[1:] = np.cumsum(cnt)
    pos_in_row = np.arange(len(nb_flat)) - first[node_of]
    colind = np.empty(rowptr[-1], dtype=np.int32)
    val = np.empty(rowptr[-1], dtype=np.float64)
    for ci in range(3):                                 # row component
        base = rowptr[3 * node_of + ci] + 3 * pos_in_row
        for cj in range(3):
            colind[base + cj] = 3 * nb_flat + cj
            val[base + cj] = bl_flat[:, ci, cj]
    return rowptr.astype(np.int32), colind, val


def box_partition_nodes(nn, box, dofs=3):
    """box_partition for a vector problem with `dofs` unknowns per node (dof = dofs*node + c,
    node id = i + nx*j + nx*ny*k) on nn^3 or (nx, ny, nz) nodes."""
    nx, ny, nz = _dims(nn)
    bi, bj, bk = box
    nid = np.arange(nx * ny * nz, dtype=np.int64)
    i, j, k = nid % nx, (nid // nx) % ny, nid // (nx * ny)
    ni, nj, nk = -(-nx // bi), -(-ny // bj), -(-nz // bk)
    part = ((k // bk) * nj + (j // bj)) * ni + (i // bi)
    return np.repeat(part, dofs).astype(np.int32), ni * nj * nk
