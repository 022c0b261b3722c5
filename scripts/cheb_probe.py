"""Experiment (iteration counts only): additive two-level preconditioner with a Chebyshev polynomial smoother
q_{k-1}(D^-1 A) D^-1 instead of plain block-Jacobi, through the distributed driver's preconditioner hook on one GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import numpy as np, torch, torch.distributed as dist
import meshfem_amd as M
from meshfem_amd import distributed as D

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dev = torch.device("cuda", 0)
lm = D.slab_local_mesh(n, 0, 1, 2, n, device=0)
c = M.Context(0); c.mesh_set(3, 2, lm.elem_nodes, lm.node_pos, lm.n_owned); c.material_isotropic(200.0, 0.35)
c.symbolic(False); c.set_option("reembed", 1); c.assemble()
ops = D.HipLocalOps(c, 3); halo = D.HaloExchange(lm, 0, 1, dev)
fixed_nodes = np.flatnonzero(lm.lattice[:, 0] == 0)
c.fix_variables((3 * fixed_nodes[:, None] + np.arange(3)[None, :]).ravel())
f = torch.as_tensor(D.slab_traction_load(lm, n, [0.0, -1.0, 0.0]).ravel(), device=dev)
t0 = time.time(); tl = D.DistributedTwoLevel(ops, halo, lm.node_pos, lm.n_owned, int(os.environ.get("AGG", 1000))); torch.cuda.synchronize(); print("aggregates", tl.n_agg, "coarse dim", tl.m, "setup %.2f s" % (time.time() - t0), flush=True)
nr = ops.n_rows

def Aop(v, out):
    ops.spmv(v, out); ops.mask_fixed(out)

# lambda_max(D^-1 A) by power iteration
v = torch.randn(nr, dtype=torch.float64, device=dev); ops.mask_fixed(v)
t1, t2 = ops.zeros(nr), ops.zeros(nr)
lam = 1.0
for _ in range(20):
    Aop(v, t1); ops.precond(t1, t2); lam = (torch.dot(v, t2) / torch.dot(v, v)).item(); v = t2 / t2.norm()
print("lambda_max(D^-1 A) ~", lam, flush=True)
lmax = 1.1 * lam

def make_cheb(k, ratio):
    """z = q(D^-1 A) D^-1 r: k-step Chebyshev iteration for A z = r from z = 0 on [lmax/ratio, lmax] (k-1 operator applies)."""
    lmin = lmax / ratio
    theta, delta = 0.5 * (lmax + lmin), 0.5 * (lmax - lmin)
    res, d, tmp, Ad = ops.zeros(nr), ops.zeros(nr), ops.zeros(nr), ops.zeros(nr)
    def smooth(r, z):
        sigma = theta / delta
        rho = 1.0 / sigma
        ops.precond(r, tmp)
        d.copy_(tmp).mul_(1.0 / theta)
        z.copy_(d)
        res.copy_(r)
        for _ in range(k - 1):
            Aop(d, Ad); res.sub_(Ad)
            rho_new = 1.0 / (2 * sigma - rho)
            ops.precond(res, tmp)
            d.mul_(rho_new * rho).add_(tmp, alpha=2 * rho_new / delta)
            z.add_(d)
            rho = rho_new
    return smooth

zc = ops.zeros(nr)
def make_pre(k, ratio):
    smooth = make_cheb(k, ratio) if ratio else None
    def pre(r, z):
        if k == 1 and ratio == 0:
            tl(r, z); return
        tl(r, zc)                      # D^-1 r + Q r
        ops.precond(r, z)              # D^-1 r
        zc.sub_(z)                     # Q r
        smooth(r, z)
        z.add_(zc)
    return pre

for k, ratio in ([(1, 0), (2, 10), (3, 10), (3, 30), (4, 30), (4, 60)] if not os.environ.get("ONLY_K1") else [(1, 0), (3, 10)]):
    torch.cuda.synchronize(); t0 = time.time()
    u, info = D.distributed_pcg(ops, halo, f, rtol=1e-8, maxit=3000, precond=make_pre(k, ratio))
    torch.cuda.synchronize()
    print("k=%d ratio=%g: %d iterations, %.3f s (python-driven), operator applies/iteration %d" % (k, ratio, info["iterations"], time.time() - t0, k), flush=True)
dist.destroy_process_group()
