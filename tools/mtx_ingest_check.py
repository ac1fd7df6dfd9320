#!/usr/bin/env python3
"""Set-up ingest at full size (run on the GPU box): the headline matrix (Q1 elasticity 70^3, 81 M nonzeros)
written as a symmetric MatrixMarket file, read back by preAlps_OperatorBuild (mapped file, parsed by the host
threads) and solved; the same problem built from memory must take the same number of iterations.
usage: mtx_ingest_check.py [n=70] [dir=/tmp]"""
import os, sys, time, subprocess
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import prealps_amd as pa
from prealps_amd import gen

n = int(sys.argv[1]) if len(sys.argv) > 1 else 70
d = sys.argv[2] if len(sys.argv) > 2 else "/tmp"
rp, ci, v = gen.elasticity3d_csr(n)
part, P = gen.box_partition_nodes(n, (2, 4, 8))
N = len(rp) - 1
mtx, pf = os.path.join(d, "ela%d.mtx" % n), os.path.join(d, "ela%d.part" % n)
t0 = time.time()
rows = np.repeat(np.arange(N, dtype=np.int64), np.diff(rp))
keep = rows >= ci
import pandas as pd
with open(mtx, "w") as f:
    f.write("%%%%MatrixMarket matrix coordinate real symmetric\n%d %d %d\n" % (N, N, int(keep.sum())))
pd.DataFrame({"i": rows[keep] + 1, "j": ci[keep].astype(np.int64) + 1, "v": v[keep]}).to_csv(
    mtx, sep=" ", header=False, index=False, float_format="%.17g", mode="a")
np.savetxt(pf, part, fmt="%d")
print("wrote %s: %.2f GB, %d entries, %.0f s" % (mtx, os.path.getsize(mtx) / 1e9, int(keep.sum()), time.time() - t0), flush=True)
# in-memory build of the matrix the FILE describes: the lower triangle mirrored (the assembled values differ
# from their transposes in the last bit, and on this matrix that is enough to move the iteration count)
import scipy.sparse as sp
Lw = sp.tril(sp.csr_matrix((v, ci, rp), shape=(N, N)), 0, format="csr")
As = (Lw + sp.tril(Lw, -1).T).tocsr()
As.sort_indices()
rp, ci, v = As.indptr.astype(np.int32), As.indices.astype(np.int32), As.data.copy()
prob = pa.EcgProblem(rp, ci, v, P, part, scale=True, device=0)
r0 = prob.solve(prob.reference_rhs(), 4, max_iter=3000)
prob.close()
# through the file
os.environ["PREALPS_PARTITION_FILE"] = pf
os.environ["PREALPS_SETUP_TRACE"] = "1"
t0 = time.time()
prob = pa.EcgProblem.from_mtx(mtx, nparts=P, device=0)
tb = time.time() - t0
r1 = prob.solve(prob.reference_rhs(), 4, max_iter=3000)
prob.close()
print("preAlps_OperatorBuild(file): %.2f s in all (phases on stderr); iterations: from memory %d, from the file %d; "
      "final residuals %.6e / %.6e; host threads %s" % (tb, r0.iters, r1.iters, r0.final_res, r1.final_res, os.environ.get("OMP_NUM_THREADS")))
os.remove(mtx); os.remove(pf)
