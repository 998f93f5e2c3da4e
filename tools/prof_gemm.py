"""micro-driver: GEMM shapes of the C2 step through the C ABI"""
import sys, torch
sys.path.insert(0, '.')
import audiogan_amd.kernels as K
# (tests/... shapes logged from train.gd_step at B = 64: the critic iteration runs M = 16384 rows, the generator iteration 8192)
shapes = [(16384, 2048, 512, 0, 1), (16384, 1024, 1024, 0, 1), (16384, 512, 1024, 0, 1), (8192, 2048, 512, 0, 1),
          (8192, 1024, 1024, 0, 1), (8192, 512, 1024, 0, 1),
          (16384, 512, 2048, 0, 0), (16384, 1024, 1024, 0, 0), (16384, 1024, 512, 0, 0), (8192, 512, 2048, 0, 0),
          (8192, 1024, 1024, 0, 0), (8192, 1024, 512, 0, 0),
          (1024, 1024, 16384, 1, 0), (2048, 512, 16384, 1, 0), (2048, 512, 16256, 1, 0), (512, 1024, 16384, 1, 0),
          (4096, 1024, 1984, 1, 0), (4096, 256, 1984, 1, 0)]
for M, N, Kd, ta, tb in shapes:
    A = torch.randn((Kd, M) if ta else (M, Kd), device='cuda')
    B = torch.randn((N, Kd) if tb else (Kd, N), device='cuda')
    C = torch.empty(M, N, device='cuda')
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for it in range(2):
        ev[0].record()
        for _ in range(40):
            K.gemm(A, B, C, ta=bool(ta), tb=bool(tb), defer=False)
        ev[1].record(); torch.cuda.synchronize()
    us = ev[0].elapsed_time(ev[1]) * 25
    # vendor library on the same shape (torch.matmul fp32 -> hipBLASLt / rocBLAS), as a practical ceiling
    At, Bt = (A.t() if ta else A), (B.t() if tb else B)
    torch.backends.cuda.matmul.allow_tf32 = False
    for it in range(2):
        ev[0].record()
        for _ in range(40):
            torch.matmul(At, Bt, out=C)
        ev[1].record(); torch.cuda.synchronize()
    us2 = ev[0].elapsed_time(ev[1]) * 25
    print('M=%d N=%d K=%d ta=%d tb=%d: %.1f us  %.1f TF   (vendor sgemm %.1f us %.1f TF)' % (
        M, N, Kd, ta, tb, us, 2.0 * M * N * Kd / us / 1e6, us2, 2.0 * M * N * Kd / us2 / 1e6))
