"""micro-driver: one biLSTM layer forward/backward through the C ABI, persistent launch vs one launch per step
(python tools/prof_lstm.py T B H)"""
import sys, torch
sys.path.insert(0, '.')
import audiogan_amd.kernels as K
T, B, H = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
nd = 2
dev = 'cuda'
pre = [torch.randn(T, B, 4 * H, device=dev) for _ in range(nd)]
whh = [torch.randn(4 * H, H, device=dev) / H ** 0.5 for _ in range(nd)]
c = [torch.zeros(T + 1, B, H, device=dev) for _ in range(nd)]
hb = [torch.zeros(2, B, H, device=dev) for _ in range(nd)]
y = torch.empty(T, B, nd * H, device=dev)
dy = torch.randn(T, B, nd * H, device=dev)
dg = [torch.empty(T, B, 4 * H, device=dev) for _ in range(nd)]
dh = [torch.zeros(2, B, H, device=dev) for _ in range(nd)]
dc = [torch.zeros(2, B, H, device=dev) for _ in range(nd)]
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
ys = {}
for persist in (False, True, False, True):
    K.PERSIST[0] = persist
    for it in range(3):
        p = [t.clone() for t in pre]
        torch.cuda.synchronize()
        ev[0].record()
        K.lstm_seq_fwd(p, whh, c, hb, y, None)
        ev[1].record()
        K.lstm_seq_bwd(p, whh, c, dy, dg, dh, dc, None)
        ev[2].record(); torch.cuda.synchronize()
    ys[persist] = (y.clone(), dg[0].clone())
    print('T=%d B=%d H=%d persist=%d: fwd %.1f us/step   bwd %.1f us/step   status %d' % (
        T, B, H, persist, ev[0].elapsed_time(ev[1]) * 1e3 / T, ev[1].elapsed_time(ev[2]) * 1e3 / T,
        K.lstm_persist_status()), flush=True)
print('max |y_persist - y_steps| = %.3g   max |dg| diff = %.3g' % (
    float((ys[True][0] - ys[False][0]).abs().max()), float((ys[True][1] - ys[False][1]).abs().max())))
