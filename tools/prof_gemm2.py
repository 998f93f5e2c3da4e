"""does the timing method explain lab (255 us) vs prof_gemm (288 us) on [16384 x 1024] x [1024 x 1024]^T ?"""
import sys, torch, time
sys.path.insert(0, '.')
import audiogan_amd.kernels as K
M, N, Kd = 16384, 1024, 1024
A = torch.randn(M, Kd, device='cuda'); B = torch.randn(N, Kd, device='cuda'); C = torch.empty(M, N, device='cuda')
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for reps in (10, 20, 50, 200):
    for it in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev[0].record()
        for _ in range(reps):
            K.gemm(A, B, C, tb=True)
        ev[1].record()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        print('reps %3d  gpu %.1f us/launch   host issue %.1f us/launch' % (reps, ev[0].elapsed_time(ev[1]) * 1e3 / reps, (t1 - t0) * 1e6 / reps))
# in a graph
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    K.gemm(A, B, C, tb=True)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, capture_error_mode='thread_local'):
        for _ in range(20):
            K.gemm(A, B, C, tb=True)
for it in range(3):
    ev[0].record(); g.replay(); ev[1].record(); torch.cuda.synchronize()
    print('graph of 20: %.1f us/launch' % (ev[0].elapsed_time(ev[1]) * 1e3 / 20))
At = A.t().contiguous()
for it in range(3):
    ev[0].record()
    for _ in range(50):
        torch.matmul(A, B.t(), out=C)
    ev[1].record(); torch.cuda.synchronize()
    print('vendor 50: %.1f us/launch' % (ev[0].elapsed_time(ev[1]) * 1e3 / 50))
