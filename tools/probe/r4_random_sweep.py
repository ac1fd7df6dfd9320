#!/usr/bin/env python3
"""Randomised parity sweep (a one-off search for cases the fixed tests do not reach, not part of the suite):
random matrix class (Poisson boxes / slabs of random shape, Q1 elasticity with its coefficient jumps, an
unstructured SPD graph Laplacian through the library's partitioner), random number of subdomains, random panel
width 1 .. 16, random variant (Orthodir / Orthomin / fused Orthodir, with and without reduction of the search
directions), several solves on the same problem object in a random order.  Each solve against the oracle:
iteration count (within 2 when the recurrence is chaotic, exact otherwise), the first residuals to 1e-7, the
block-size sequence for the first iterations.  Prints one line per case and a summary; exits non-zero on a
mismatch.  usage: r4_random_sweep.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import scipy.sparse as sp
import prealps_amd as pa
from prealps_amd import gen
from prealps_amd.solver import partition_kway
from oracle import oracle as O

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
ALGS = {"odir": (pa.ORTHODIR, O.ORTHODIR), "omin": (pa.ORTHOMIN, O.ORTHOMIN), "fused": (pa.ORTHODIR_FUSED, O.ORTHODIR_FUSED)}


def make_problem():
    kind = rng.choice(["poisson_boxes", "poisson_slabs", "elasticity", "unstructured"])
    if kind == "poisson_boxes":
        n = int(rng.integers(8, 21))
        divs = [d for d in (2, 3, 4, 5) if n % d == 0] or [1]
        box = tuple(n // int(rng.choice(divs)) for _ in range(3))
        rp, ci, v = gen.poisson3d_csr(n)
        part, P = gen.box_partition(n, box)
        A = sp.csr_matrix((v, ci, rp), shape=(n ** 3, n ** 3))
        desc = "poisson %d^3 boxes %s" % (n, box)
    elif kind == "poisson_slabs":
        n = int(rng.integers(8, 17))
        A = O.poisson3d(n)
        P = int(rng.choice([4, 8, 16]))
        part = O.contiguous_partition(A.shape[0], P)
        desc = "poisson %d^3 in %d slabs" % (n, P)
    elif kind == "elasticity":
        nn = int(rng.integers(6, 12))
        divs = [d for d in (2, 3) if nn % d == 0] or [1]
        box = tuple(max(2, nn // int(rng.choice(divs))) for _ in range(3))
        rp, ci, v = gen.elasticity3d_csr(nn)
        part, P = gen.box_partition_nodes(nn, box)
        N = len(rp) - 1
        A = sp.csr_matrix((v, ci, rp), shape=(N, N))
        desc = "elasticity %d^3 nodes boxes %s" % (nn, box)
    else:
        N = int(rng.integers(600, 4000))
        deg = int(rng.integers(3, 9))
        rows = np.repeat(np.arange(N), deg)
        # neighbours close in index (a banded random graph: connected, partitionable)
        cols = np.clip(rows + rng.integers(-40, 41, size=rows.size), 0, N - 1)
        keep = rows != cols
        W = sp.coo_matrix((rng.uniform(0.5, 2.0, keep.sum()), (rows[keep], cols[keep])), shape=(N, N)).tocsr()
        W = W + W.T
        chain = sp.diags([np.ones(N - 1), np.ones(N - 1)], [1, -1])       # keeps the graph connected
        W = (W + chain).tocsr()
        A = (sp.diags(np.asarray(W.sum(axis=1)).ravel() + rng.uniform(0.01, 0.5, N)) - W).tocsr()
        A.sort_indices()
        P = int(rng.choice([4, 8, 16, 32]))
        rp, ci, v = O.as_csr(A)
        part = partition_kway(rp, ci, P)
        desc = "unstructured N=%d deg~%d, %d parts (k-way)" % (N, 2 * deg, P)
    return A, part, int(P), desc


bad = 0
done = 0
while done < cases:
    A, part, P, desc = make_problem()
    if P < 2 or A.shape[0] // P < 4:
        continue
    rp, ci, v = O.as_csr(A)
    prob = pa.EcgProblem(rp, ci, v, P, part, scale=True, device=0)
    B, perm, rowpos = O.permute_by_part(O.symrac_scale(A), part, P)
    rhs = prob.reference_rhs()
    for _ in range(int(rng.integers(2, 5))):
        t = int(rng.choice([1, 2, 3, 4, 4, 4, 5, 6, 8, 8, 12, 16]))
        if t > P:
            continue
        name = str(rng.choice(["odir", "odir", "omin", "fused"]))
        red = bool(rng.integers(0, 2)) and not (name == "fused" and t > 4)
        iters = int(rng.choice([15, 40, 120]))
        a = ALGS[name]
        done += 1
        tag = "%-46s t=%2d %-5s red=%d maxit=%3d" % (desc, t, name, red, iters)
        try:
            ref = O.ECG(B, rowpos, t, a[1], O.ADAPT_BS if red else O.NO_BS_RED, 1e-5, iters).solve(rhs)
        except RuntimeError as e:
            try:
                prob.solve(rhs, t, ortho_alg=a[0], bs_red=pa.ADAPT_BS if red else pa.NO_BS_RED, max_iter=iters)
                print(tag, "oracle broke down (%s), HIP path went on: not compared" % str(e)[:40])
            except pa.PreAlpsError:
                print(tag, "both break down")
            continue
        try:
            got = prob.solve(rhs, t, ortho_alg=a[0], bs_red=pa.ADAPT_BS if red else pa.NO_BS_RED, max_iter=iters)
        except pa.PreAlpsError as e:
            print(tag, "HIP path FAILED where the oracle did not:", str(e)[:80]); bad += 1
            continue
        k = min(10, len(got.res), len(ref["res"]))
        rel = np.abs(got.res[:k] - ref["res"][:k]) / np.abs(ref["res"][:k])
        ok = rel.max() <= 1e-7 and abs(got.iters - ref["iters"]) <= 2
        kb = min(k, len(got.bs), len(ref["bs"]))
        ok = ok and list(got.bs[:kb]) == list(ref["bs"][:kb])
        if not ok: bad += 1
        print(tag, "iters %3d / %3d  max rel diff of the first %d residuals %.1e  bs %s%s" %
              (got.iters, ref["iters"], k, rel.max(), list(got.bs[-1:]), "" if ok else "   <-- MISMATCH"), flush=True)
    prob.close()
print("%d solves, %d mismatches" % (done, bad))
sys.exit(1 if bad else 0)
