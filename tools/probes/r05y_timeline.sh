#!/bin/bash
# r05y: s_memtime stamps at the part boundaries of the ping-pong loop (waves 0 / 4 of workgroup 0, stages 8..15): how long L and M parts really are
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
cd adaface-dev_amd/csrc && rm -f af_gemm3.o && make EXTRA=-DAF_CONV3H_ABLATIONS > /dev/null 2>&1; cd ../..
export AF_GEMM3_ABLATE_DYNAMIC=1
timeout 600 python - > gpurun_out/r05y_timeline.txt 2>&1 <<'PY'
import os, sys, torch, ctypes as C
sys.path.insert(0, '.')
from adaface_dev_amd import ops, _lib
dev = torch.device('cuda:0')
g = torch.Generator(device='cpu').manual_seed(0)
rnd = lambda *s: (torch.randn(*s, generator=g) * 0.5).half().to(dev)
for (B, H, W, ci, co) in [(8, 64, 64, 320, 320), (1, 64, 64, 320, 320), (8, 32, 32, 640, 640)]:
    x, w = rnd(B, H, W, ci), rnd(co, ci, 3, 3) * 0.05
    pw = ops.pack_conv3x3(w, None, dev)
    out = torch.empty((B, H, W, co), dtype=torch.float16, device=dev)
    ws = torch.zeros(4096, dtype=torch.int64, device=dev)
    d = _lib.GemmDesc()
    d.a1, d.wt, d.out = x.data_ptr(), pw.wt.data_ptr(), out.data_ptr()
    d.M, d.N, d.K, d.kpad, d.taps, d.c1 = B * H * W, co, pw.K, pw.kpad, 9, ci
    d.B, d.H, d.W, d.Ho, d.Wo, d.stride, d.rows_per_batch = B, H, W, H, W, 1, H * W
    d.tile, d.splits = 14, 1
    d.zeros = ops._zero_page(dev).data_ptr()
    d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 8
    st = torch.cuda.current_stream().cuda_stream
    os.environ['AF_GEMM3_ABLATE'] = '32'
    for _ in range(3):
        rc = _lib.lib().af_gemm(C.byref(d), st)
    torch.cuda.synchronize()
    t = ws.cpu().reshape(-1)[:8 * 2 * 8].reshape(8, 2, 8)
    print(f"conv B{B} {H}x{W} {ci}->{co} rc {rc}: s_memtime ticks (100 MHz constant clock? printed raw), per stage 8..15")
    t0 = int(t[0, 0, 0])
    for s in range(8):
        g0 = [int(v) - t0 for v in t[s, 0]]
        g1 = [int(v) - t0 for v in t[s, 1]]
        print(f"  st {8 + s}: G0 L-start {g0[0]} reads+dma-retired {g0[3]} barrier {g0[4]} mfma-issued {g0[5]} vmcnt {g0[6]} barrier {g0[7]} | G1 start {g1[0]} mfma-issued {g1[1]} vmcnt {g1[2]} barrier {g1[3]} L-retired {g1[6]} barrier {g1[7]}")
    # durations
    import statistics as S
    L0 = [int(t[s, 0, 3] - t[s, 0, 0]) for s in range(8)]; W0 = [int(t[s, 0, 4] - t[s, 0, 3]) for s in range(8)]
    M0 = [int(t[s, 0, 5] - t[s, 0, 4]) for s in range(8)]; V0 = [int(t[s, 0, 6] - t[s, 0, 5]) for s in range(8)]; X0 = [int(t[s, 0, 7] - t[s, 0, 6]) for s in range(8)]
    M1 = [int(t[s, 1, 1] - t[s, 1, 0]) for s in range(8)]; V1 = [int(t[s, 1, 2] - t[s, 1, 1]) for s in range(8)]; X1 = [int(t[s, 1, 3] - t[s, 1, 2]) for s in range(8)]
    L1 = [int(t[s, 1, 6] - t[s, 1, 3]) for s in range(8)]; W1 = [int(t[s, 1, 7] - t[s, 1, 6]) for s in range(8)]
    stage = [int(t[s + 1, 0, 0] - t[s, 0, 0]) for s in range(7)]
    print("  medians: G0 L", S.median(L0), "wait@barrier", S.median(W0), "M", S.median(M0), "vmcnt", S.median(V0), "wait@barrier", S.median(X0),
          "| G1 M", S.median(M1), "vmcnt", S.median(V1), "wait@barrier", S.median(X1), "L", S.median(L1), "wait@barrier", S.median(W1), "| stage", S.median(stage))
PY
cat gpurun_out/r05y_timeline.txt
