"""could the generator's thin transposed convolutions run as GEMMs?  Their polyphase form is, per layer, ONE product
[rows = O * s] x [K = C * taps] times [K] x [N = B * L / s] (weights shared by all clips and columns).  This times the plain
product of the same shape through ag_gemm (no tap shift, no phase-interleaved store: a lower bound for a conv kernel built
that way) next to the conv engine's time for the layer (profiles/r04_conv_layers_b64.txt)."""
import sys, torch
sys.path.insert(0, '.')
import audiogan_amd.kernels as K
# (name, rows, K, N, conv-engine microseconds at batch 64)
shapes = [('G1.deconv fwd', 128, 256, 65536, 81.3), ('G2.deconv fwd', 128, 128, 131072, 81.4), ('G4.deconv fwd', 128, 64, 131072, 67.8),
          ('G2.conv bwd-x', 68, 192, 131072, 74.4), ('G3.conv bwd-x', 196, 192, 131072, 139.7), ('G4.conv bwd-x', 324, 96, 131072, 139.3)]
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for name, M, Kd, N, eng in shapes:
    A = torch.randn(M, Kd, device='cuda'); B = torch.randn(Kd, N, device='cuda'); C = torch.empty(M, N, device='cuda')
    for it in range(2):
        ev[0].record()
        for _ in range(40):
            K.gemm(A, B, C)
        ev[1].record(); torch.cuda.synchronize()
    us = ev[0].elapsed_time(ev[1]) * 25
    At = A.t().contiguous()      # weights k-major, as the conv engine's prepared layout
    for it in range(2):
        ev[0].record()
        for _ in range(40):
            K.gemm(At, B, C, ta=True)
        ev[1].record(); torch.cuda.synchronize()
    us_t = ev[0].elapsed_time(ev[1]) * 25
    for it in range(2):
        ev[0].record()
        for _ in range(40):
            torch.matmul(A, B, out=C)
        ev[1].record(); torch.cuda.synchronize()
    us_v = ev[0].elapsed_time(ev[1]) * 25
    fl = 2.0 * M * N * Kd
    print('%-14s M=%4d K=%4d N=%6d: ag_gemm %6.1f us (%5.1f TF)  ta %6.1f us   vendor %6.1f us   conv engine %6.1f us   bytes/8TB/s %5.1f us' % (
        name, M, Kd, N, us, fl / us / 1e6, us_t, us_v, eng, 4.0 * (Kd * N + M * N) / 8e6))
