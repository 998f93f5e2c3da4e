"""Per-layer roofline of the C2 conv stacks: every conv layer of G and D at batch 64, forward /
backward-data / backward-weight through the C ABI, timed with HIP events (10 launches, best of 3 rounds).
Algorithmic FLOPs = 2*O*Lout*C*K per clip; algorithmic bytes = (input + output + weight) * 4 (forward), the same
traffic for backward-data, (x + dy + dw) * 4 for backward-weight.   python tools/prof_layers.py > profiles/..."""
import sys, torch
sys.path.insert(0, '.')
import audiogan_amd.kernels as K
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
PREC = sys.argv[2] if len(sys.argv) > 2 else 'f32'
K.set_precision(PREC)
LAYERS = [('G1.conv', 'conv', 1, 128, 17, 8, 8, 8192), ('G1.deconv', 'convt', 128, 16, 16, 8, 4, 1024),
          ('G2.conv', 'conv', 17, 64, 9, 4, 4, 8192), ('G2.deconv', 'convt', 64, 32, 8, 4, 2, 2048),
          ('G3.conv', 'conv', 49, 64, 9, 4, 4, 8192), ('G3.deconv', 'convt', 64, 32, 8, 4, 2, 2048),
          ('G4.conv', 'conv', 81, 32, 9, 4, 4, 8192), ('G4.deconv', 'convt', 32, 32, 8, 4, 2, 2048),
          ('G5.final', 'o1', 113, 1, 3, 1, 1, 8192),
          ('D1', 'conv', 1, 16, 7, 2, 3, 8192), ('D2', 'conv', 16, 32, 7, 2, 3, 4096), ('D3', 'conv', 32, 64, 7, 2, 3, 2048),
          ('D4', 'conv', 64, 128, 7, 2, 3, 1024), ('D5', 'conv', 128, 256, 7, 2, 3, 512), ('D6', 'conv', 256, 512, 7, 2, 3, 256)]
# priced against the dense matrix peak of the mode's MFMA: fp32 157.3 TFLOP/s, bf16 2500 TFLOP/s (MI355X_MICROARCH.md).
# In bf16 mode a few layers (single input channel, the 113 -> 1 final conv) still run fp32 FMA / MFMA kernels on rounded
# operands: their fraction of the bf16 peak is what it is.
PEAK_TF, PEAK_GBS = (2500.0 if PREC == 'bf16' else 157.3), 8000.0


def timed(fn):
    best = 1e9
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for _ in range(3):
        ev[0].record()
        for _ in range(10):
            fn()
        ev[1].record(); torch.cuda.synchronize()
        best = min(best, ev[0].elapsed_time(ev[1]) * 100)
    return best


print('batch %d, %s; TF = algorithmic TFLOP/s (peak %.1f), GB/s = algorithmic bytes / time (peak %.0f)' % (B, PREC, PEAK_TF, PEAK_GBS))
print('%-10s %-5s %9s %7s %7s %8s' % ('layer', 'pass', 'us', 'TF', 'GB/s', 'bound'))
tot = {}
for name, kind, cin, cout, k, s, p, lin in LAYERS:
    lout = lin if kind == 'o1' else ((lin + 2 * p - k) // s + 1 if kind == 'conv' else (lin - 1) * s - 2 * p + k)
    x = torch.randn(B, cin, lin, device='cuda'); y = torch.randn(B, cout, lout, device='cuda')
    flops = 2.0 * B * cout * cin * k * (lout if kind != 'convt' else lin)
    nbytes = 4.0 * (x.numel() + y.numel() + cin * cout * k)
    if kind == 'o1':
        w = torch.randn(1, cin, k, device='cuda'); b = torch.zeros(1, device='cuda'); dw = torch.zeros(1, cin, k, device='cuda')
        dx = torch.empty_like(x)
        runs = [('fwd', lambda: K.conv_o1_fwd(x, w, b, y, k, p)), ('bwd-x', lambda: K.conv_o1_bwd_data(y, w, dx, k, p)),
                ('bwd-w', lambda: K.conv_o1_wgrad(y, x, dw, k, p))]
    else:
        w = torch.randn((cout, cin, k) if kind == 'conv' else (cin, cout, k), device='cuda') / (cin * k) ** 0.5
        d0, d1, _ = w.shape
        wpa, wpb = torch.zeros(K.wpa_numel(d0, d1, k), device='cuda'), torch.zeros(K.wpb_numel(d0, d1, k, s), device='cuda')
        K.prep_conv_weight(w, wpa, wpb, s, pad=p)        # scatter layout prepared for this padding, as ops.py does
        dx, dw = torch.empty_like(x), torch.zeros_like(w)
        mode = 0 if kind == 'conv' else 1
        runs = [('fwd', lambda: K.conv_engine(x, wpa if mode == 0 else wpb, y, k, s, p, mode, wp_pad=p)),
                ('bwd-x', lambda: K.conv_engine(y, wpb if mode == 0 else wpa, dx, k, s, p, 1 - mode, wp_pad=p)),
                ('bwd-w', (lambda: K.conv_wgrad(y, x, dw, k, s, p)) if kind == 'conv' else (lambda: K.conv_wgrad(x, y, dw, k, s, p)))]
    for pname, fn in runs:
        if name == 'D1' and pname == 'bwd-x':
            continue          # the gradient w.r.t. the waveform is only needed in the generator iteration; keep it
        us = timed(fn)
        tf, gbs = flops / us / 1e6, nbytes / us / 1e3
        bound = 'mfma' if tf / PEAK_TF >= gbs / PEAK_GBS else 'hbm'
        print('%-10s %-5s %9.1f %7.1f %7.0f %8s' % (name, pname, us, tf, gbs, bound))
        t = tot.setdefault(name[0], [0.0, 0.0, 0.0]); t[0] += us; t[1] += flops; t[2] += nbytes
for stack, (us, fl, nb) in tot.items():
    print('%s stack, all passes: %.0f us, %.1f TF (%.2f of peak), %.0f GB/s (%.2f of peak)' % (
        stack, us, fl / us / 1e6, fl / us / 1e6 / PEAK_TF, nb / us / 1e3, nb / us / 1e3 / PEAK_GBS))
