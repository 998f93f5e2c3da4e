"""micro-driver: one conv weight gradient through the C ABI, N launches (for rocprofv3 --pmc / --kernel-trace):
python tools/prof_wgrad.py cin cout k s p lin [prec]   (batch 64)"""
import sys, torch
sys.path.insert(0, '.')
import audiogan_amd.kernels as K
cin, cout, k, s, p, lin = map(int, sys.argv[1:7])
K.set_precision(sys.argv[7] if len(sys.argv) > 7 else 'f32')
B = 64
lout = (lin + 2 * p - k) // s + 1
x = torch.randn(B, cin, lin, device='cuda'); dy = torch.randn(B, cout, lout, device='cuda')
dw = torch.zeros(cout, cin, k, device='cuda')
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for it in range(3):
    ev[0].record()
    for _ in range(10):
        K.conv_wgrad(dy, x, dw, k, s, p)
    ev[1].record(); torch.cuda.synchronize()
    us = ev[0].elapsed_time(ev[1]) * 100
    print('wgrad %s: %.1f us  %.1f TF' % (sys.argv[1:7], us, 2.0 * B * cout * cin * k * lout / us / 1e6))
