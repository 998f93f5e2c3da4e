"""micro-driver: the critic's GEMM shapes on bf16-STORED operands (ag_gemm_h) next to the fp32-operand bf16 kernel (ag_gemm in
'bf16' precision mode), microseconds and TFLOP/s per shape.  AG_GEMMH_VARIANT selects the staging structure (gemm_bf16s.hip)."""
import os
import sys
import torch
sys.path.insert(0, '.')
import audiogan_amd.kernels as K
K.set_precision('bf16')
shapes = [(16384, 2048, 512, 0, 1), (16384, 1024, 1024, 0, 1), (16384, 512, 1024, 0, 1), (8192, 2048, 512, 0, 1),
          (16384, 512, 4096, 0, 0), (16384, 1024, 1024, 0, 0), (16384, 1024, 512, 0, 0),
          (2048, 512, 16384, 1, 0), (1024, 1024, 16384, 1, 0), (512, 1024, 16384, 1, 0), (2048, 512, 8192, 1, 0)]
print('variant', os.environ.get('AG_GEMMH_VARIANT', '0'))
tot_h = tot_f = 0.0
for M, N, Kd, ta, tb in shapes:
    A = torch.randn((Kd, M) if ta else (M, Kd), device='cuda')
    B = torch.randn((N, Kd) if tb else (Kd, N), device='cuda')
    A16, B16 = A.to(torch.bfloat16), B.to(torch.bfloat16)
    C = torch.empty(M, N, device='cuda')
    C16 = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    res = []
    for fn in (lambda: K.gemm_h(A16, B16, C=C if ta else None, C16=None if ta else C16, ta=bool(ta), tb=bool(tb)),
               lambda: K.gemm(A, B, C, ta=bool(ta), tb=bool(tb))):
        for it in range(2):
            ev[0].record()
            for _ in range(10):
                fn()
            ev[1].record(); torch.cuda.synchronize()
        res.append(ev[0].elapsed_time(ev[1]) * 100)
    fl = 2.0 * M * N * Kd
    tot_h += res[0]; tot_f += res[1]
    print('M=%5d N=%4d K=%5d ta=%d tb=%d: stored bf16 %6.1f us %6.0f TF | fp32 operands %6.1f us %6.0f TF' % (
        M, N, Kd, ta, tb, res[0], fl / res[0] / 1e6, res[1], fl / res[1] / 1e6))
print('sum: stored %.0f us, fp32 operands %.0f us' % (tot_h, tot_f))
