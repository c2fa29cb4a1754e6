"""Per-layer input-gradient check of ResNetFace-18 (HIP backward vs torch autograd through the CPU oracle)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from adaface_dev_amd import ops, rng
from adaface_dev_amd.evaluation.arcface_resnet import resnet_face18
from oracle import face_oracle as FO

dev = torch.device("cuda:0")
rel = lambda a, b: float((a - b).norm() / b.norm())
for use_se, lin in ((True, False), (False, False), (True, True)):
    m = resnet_face18(use_se=use_se).eval()
    sd = rng.synth_face_state_dict(m.state_dict(), seed=50)
    if lin:                                  # PReLU slope 1: no sign decisions left except the max-pool / SE sigmoid
        sd = {k: (torch.ones_like(v) if k.endswith("prelu.weight") else v) for k, v in sd.items()}
    m.load_state_dict(sd)
    m = m.to(dev)
    P, Pb = m._prepared(), m._prepared_bwd()
    x = rng.synth_input("face.gx", (3, 1, 128, 128), seed=58)
    # walk the oracle forward keeping block inputs
    with torch.no_grad():
        h = F.prelu(FO._bn(sd, "bn1.", F.conv2d(x, sd["conv1.weight"], None, 1, 1)), sd["prelu.weight"])
        h = F.max_pool2d(h, 2, 2)
    names = [(f"layer{li}.{bi}.", 2 if (li > 1 and bi == 0) else 1) for li in range(1, 5) for bi in range(2)]
    for (p, stride), blk, bp, bpb in zip(names, m.blocks(), P["blocks"], Pb["blocks"]):
        hin = h.half().float()
        hr = hin.clone().requires_grad_(True)
        out = FO.ir_block(sd, p, hr, stride, use_se)
        dy = torch.randn(out.shape, generator=torch.Generator().manual_seed(7)).half().float()
        out.backward(dy)
        xd = hin.permute(0, 2, 3, 1).contiguous().to(dev).half()
        y, sv = blk.hip_train(xd, bp)
        dx = blk.hip_bwd(sv, dy.permute(0, 2, 3, 1).contiguous().to(dev).half(), bp, bpb)
        print(f"lin={lin} se={use_se} {p} stride {stride}: fwd {rel(y.float().cpu().permute(0, 3, 1, 2), out.detach()):.2e}  "
              f"dx {rel(dx.float().cpu().permute(0, 3, 1, 2), hr.grad):.2e}")
        h = out.detach()
    # tail
    hin = h.half().float()
    hr = hin.clone().requires_grad_(True)
    t = FO._bn(sd, "bn5.", F.linear(FO._bn(sd, "bn4.", hr).reshape(3, -1), sd["fc5.weight"], sd["fc5.bias"]))
    dy = torch.randn(t.shape, generator=torch.Generator().manual_seed(8)).half().float()
    t.backward(dy)
    dh = ops.gemm(dy.to(dev).half(), Pb["fc5"]).reshape(3, 8, 8, 512)
    print(f"lin={lin} se={use_se} tail dx {rel(dh.float().cpu().permute(0, 3, 1, 2), hr.grad):.2e}")
    # stem
    xr = x.clone().requires_grad_(True)
    s = F.max_pool2d(F.prelu(FO._bn(sd, "bn1.", F.conv2d(xr, sd["conv1.weight"], None, 1, 1)), sd["prelu.weight"]), 2, 2)
    dy = torch.randn(s.shape, generator=torch.Generator().manual_seed(9)).half().float()
    s.backward(dy)
    c1 = ops.conv3x3(ops.nchw_f32_to_nhwc_f16(x.to(dev), cpad=8), P["conv1"])
    a1 = ops.affine_prelu(c1, slope=m.prelu.weight)
    dc1 = ops.affine_prelu_bwd(ops.maxpool2x2_bwd(a1, dy.permute(0, 2, 3, 1).contiguous().to(dev).half()), c1, slope=m.prelu.weight)
    dx = ops.nhwc_f16_to_nchw_f32(ops.conv3x3(dc1, Pb["conv1"]), 1)
    print(f"lin={lin} se={use_se} stem dx {rel(dx.cpu(), xr.grad):.2e}")
