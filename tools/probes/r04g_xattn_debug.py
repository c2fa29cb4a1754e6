"""Localise errors of the tiled cross-attention kernel: its q projection (AF_XATTN_DBG=1) and attention output (=2) against torch, per (16-token tile, head)."""
import os, sys
os.environ["AF_GEMM3_ABLATE_DYNAMIC"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from adaface_dev_amd import ops
from adaface_dev_amd.ldm.modules import attention as A
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, N, L, C = 1, 128, 77, 320
m = A.CrossAttention(C, 768, heads=8, dim_head=40).to(dev)
x = torch.randn(B * N, C, device=dev).half()
ctx = torch.randn(B, L, 768, device=dev).half()
k, vt = ops.gemm(ctx.reshape(B * L, 768), m._packed_kv(), rows_per_batch=L, split_col=C)
pq = m.to_q.packed()
q_ref = x.float() @ m.to_q.weight.detach().float().t()
kk = k.float().reshape(L, 8, 40).permute(1, 0, 2)
vv = vt[0, :, :L].float().reshape(8, 40, L).permute(0, 2, 1)
qq = q_ref.reshape(N, 8, 40).permute(1, 0, 2)
o_ref = (torch.softmax(qq @ kk.transpose(1, 2) * 40 ** -0.5, dim=-1) @ vv).permute(1, 0, 2).reshape(N, C)
for rep in range(3):
    for dbg, ref, scale in ((1, q_ref * (40 ** -0.5 * 1.4426950408889634), 1.0), (2, o_ref, 1.0)):
        os.environ["AF_XATTN_DBG"] = str(dbg)
        os.environ["AF_XATTN_TILED"] = "1"
        out = ops.xattn_fused(x, pq, k, vt, m.to_out[0].packed(), B=B, N=N, L=L, heads=8, scale=40 ** -0.5, ldk=C, residual=None).float()
        d = torch.nan_to_num(out - ref, nan=1e3).reshape(8, 16, 8, 40).abs().amax(dim=(1, 3))
        print("dbg", dbg, "nan", int(torch.isnan(out).sum()), "ref scale %.3f" % float(ref.abs().mean()))
        for tt in range(8):
            print("   tile", tt, " per head max err:", " ".join(f"{float(v):8.3f}" for v in d[tt]))
os.environ["AF_XATTN_DBG"] = "0"
