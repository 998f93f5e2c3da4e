"""one generator layer through the conv engine (G2.deconv forward by default): python tools/prof_one_layer.py [name]"""
import sys, torch
sys.path.insert(0, '.')
import audiogan_amd.kernels as K
LAYERS = {'G1.deconv': ('convt', 128, 16, 16, 8, 4, 1024), 'G2.deconv': ('convt', 64, 32, 8, 4, 2, 2048),
          'G4.deconv': ('convt', 32, 32, 8, 4, 2, 2048),
          # (the same products with pad 4: every phase group of the output starts on a 16-byte boundary - is the 8-byte offset of
          # the k8 s4 p2 layers' stores what their epilogue costs?)
          'G2.deconv.p4': ('convt', 64, 32, 8, 4, 4, 2048), 'G4.deconv.p4': ('convt', 32, 32, 8, 4, 4, 2048), 'G3.conv': ('conv', 49, 64, 9, 4, 4, 8192), 'G4.conv': ('conv', 81, 32, 9, 4, 4, 8192)}
B = 64
for name in (sys.argv[1:] or ['G2.deconv', 'G4.conv']):
    kind, cin, cout, k, s, p, lin = LAYERS[name]
    lout = (lin + 2 * p - k) // s + 1 if kind == 'conv' else (lin - 1) * s - 2 * p + k
    x = torch.randn(B, cin, lin, device='cuda'); y = torch.randn(B, cout, lout, device='cuda')
    w = torch.randn((cout, cin, k) if kind == 'conv' else (cin, cout, k), device='cuda') / (cin * k) ** 0.5
    d0, d1, _ = w.shape
    wpa, wpb = torch.zeros(K.wpa_numel(d0, d1, k), device='cuda'), torch.zeros(K.wpb_numel(d0, d1, k, s), device='cuda')
    K.prep_conv_weight(w, wpa, wpb, s, pad=p)
    dx = torch.empty_like(x)
    if kind == 'convt':
        fn, what = (lambda: K.conv_engine(x, wpb, y, k, s, p, 1, wp_pad=p)), 'fwd'
    else:
        fn, what = (lambda: K.conv_engine(y, wpb, dx, k, s, p, 1, wp_pad=p)), 'bwd-x'
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    best = 1e9
    for _ in range(3):
        ev[0].record()
        for _ in range(20):
            fn()
        ev[1].record(); torch.cuda.synchronize()
        best = min(best, ev[0].elapsed_time(ev[1]) * 50)
    print('%-10s %-5s %7.1f us' % (name, what, best))
