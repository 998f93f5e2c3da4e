"""micro-driver: one conv layer through the C ABI, N launches (for rocprofv3 --pmc / --kernel-trace)"""
import sys, torch
sys.path.insert(0, '.')
import audiogan_amd.kernels as K
kind, cin, cout, k, s, p, lin, mode_bwd = sys.argv[1], *map(int, sys.argv[2:9])
B = 64
w = torch.randn((cout, cin, k) if kind == 'conv' else (cin, cout, k), device='cuda') / (cin * k) ** 0.5
d0, d1, _ = w.shape
wpa, wpb = torch.zeros(K.wpa_numel(d0, d1, k), device='cuda'), torch.zeros(K.wpb_numel(d0, d1, k, s), device='cuda')
K.prep_conv_weight(w, wpa, wpb, s, pad=p)
lout = (lin + 2 * p - k) // s + 1 if kind == 'conv' else (lin - 1) * s - 2 * p + k
x = torch.randn(B, cin, lin, device='cuda'); y = torch.randn(B, cout, lout, device='cuda')
fx, aty = torch.empty_like(y), torch.empty_like(x)
mode = 0 if kind == 'conv' else 1
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for it in range(3):
    ev[0].record()
    for _ in range(10):
        if mode_bwd:
            K.conv_engine(y, wpb if mode == 0 else wpa, aty, k, s, p, 1 - mode, wp_pad=p)
        else:
            K.conv_engine(x, wpa if mode == 0 else wpb, fx, k, s, p, mode, wp_pad=p)
    ev[1].record(); torch.cuda.synchronize()
    us = ev[0].elapsed_time(ev[1]) * 100
    print('%s %s: %.1f us  %.1f TF' % (sys.argv[1:8], 'bwd' if mode_bwd else 'fwd', us, 2.0 * B * cout * cin * k * (lout if kind == 'conv' else lin) / us / 1e6))
