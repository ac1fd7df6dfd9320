"""Solve a generated Q1 elasticity problem to tolerance and print one JSON line:
  python examples/solve_elasticity.py <nodes/side> <t> <odir|omin|fused> <0|1 block-size reduction> <bi,bj,bk>
"""
import sys, time, json
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np
import prealps_amd as pa
from prealps_amd import gen
nn, t, alg, red = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
box = tuple(int(x) for x in sys.argv[5].split(','))
t0 = time.time(); rp, ci, v = gen.elasticity3d_csr(nn); part, nparts = gen.box_partition_nodes(nn, box); tg = time.time() - t0
t0 = time.time(); prob = pa.EcgProblem(rp, ci, v, nparts, part); prob.create_block_jacobi(); ts = time.time() - t0
rhs = prob.reference_rhs()
algs = {"odir": pa.ORTHODIR, "omin": pa.ORTHOMIN, "fused": pa.ORTHODIR_FUSED}[alg]
r = prob.solve(rhs, t, ortho_alg=algs, bs_red=pa.ADAPT_BS if red == "1" else pa.NO_BS_RED, max_iter=3000)
print(json.dumps(dict(nn=nn, N=3*nn**3, nnz=len(v), t=t, alg=alg, red=red, box=box, nparts=nparts, gen_s=round(tg,1), setup_s=round(ts,1),
      iters=r.iters, final_res=r.final_res, normb=r.normb, final_bs=r.final_bs, solve_s=round(r.seconds,3), its_per_s=round(r.iters/r.seconds,1),
      bs_hist=[int(b) for b in r.bs[::max(1,len(r.bs)//12)]], w=prob.stat("bj_max_bandwidth"))))
