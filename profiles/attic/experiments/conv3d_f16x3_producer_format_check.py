import ctypes, os, sys, torch
sys.path.insert(0, '.')
from diffuvolume_amd import submodule as S, _lib
dev = 'cuda:0'
lib = ctypes.CDLL(os.environ['DV_LIB_PATH'])
P, I = ctypes.c_void_p, ctypes.c_int
lib.dv_f16x3_split_bytes.restype = ctypes.c_size_t; lib.dv_f16x3_split_bytes.argtypes = [I] * 5
lib.dv_f16x3_split_pack_f32.restype = I; lib.dv_f16x3_split_pack_f32.argtypes = [P, P, P, I, I, I, I, I, P]
lib.dv_conv3d_f16x3p_f32.restype = I; lib.dv_conv3d_f16x3p_f32.argtypes = [P] * 8 + [I] * 7 + [P]
torch.manual_seed(0)
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
def split(x):
    b, c, d, h, w = x.shape
    n = lib.dv_f16x3_split_bytes(b, c, d, h, w)
    buf = torch.empty(n, dtype=torch.uint8, device=dev)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    rc = lib.dv_f16x3_split_pack_f32(x.data_ptr(), buf.data_ptr(), flag.data_ptr(), b, c, d, h, w, None)
    assert rc == 0, rc
    return buf, flag
def run(b, cin, cout, dims, time=False):
    x = torch.randn(b, cin, *dims, device=dev)
    w = torch.randn(cout, cin, 3, 3, 3, device=dev) * (2.0 / (27 * cin)) ** 0.5
    bn = (torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev) * 0.1, torch.randn(cout, device=dev) * 0.1, torch.rand(cout, device=dev) + 0.5)
    res = torch.randn(b, cout, *dims, device=dev)
    p16 = S.Conv3dPlan(w, bn, stride=1, act=S.ACT_RELU, precision="f16x3")
    p32 = S.Conv3dPlan(w, bn, stride=1, act=S.ACT_RELU, precision="f32")
    y16, y32 = p16(x, residual=res), p32(x, residual=res)
    xs, _ = split(x)
    out = torch.empty_like(y16)
    outs = torch.empty(lib.dv_f16x3_split_bytes(b, cout, *dims), dtype=torch.uint8, device=dev)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    def conv(o=out, osp=outs):
        rc = lib.dv_conv3d_f16x3p_f32(xs.data_ptr(), p16.wpacked.data_ptr(), p16.scale.data_ptr(), p16.shift.data_ptr(), res.data_ptr(),
                                      None if o is None else o.data_ptr(), None if osp is None else osp.data_ptr(), flag.data_ptr(),
                                      b, cin, *dims, cout, S.ACT_RELU, None)
        assert rc == 0, rc
    conv()
    torch.cuda.synchronize()
    ref = torch.relu(torch.nn.functional.batch_norm(torch.nn.functional.conv3d(x.double(), w.double(), None, 1, 1), bn[2].double(), bn[3].double(), bn[0].double(), bn[1].double(), False, 0.0, 1e-5) + res.double()) if x.numel() < 3e7 else None
    chk, _ = split(out)
    msg = f"B{b} {cin}->{cout} {dims}: p==f16x3 {torch.equal(out, y16)} max|p-f16x3| {float((out - y16).abs().max()):.2e} split(out)==out_split {torch.equal(chk, outs)}"
    if ref is not None:
        s = float(ref.abs().max())
        msg += f" err vs f64: p {float((out.double() - ref).abs().max()) / s:.2e} f32wino {float((y32.double() - ref).abs().max()) / s:.2e}"
    print(msg, flush=True)
    if time:
        print(f"   time ms: wino f32 {timeit(lambda: p32(x, residual=res)):.3f}  f16x3 {timeit(lambda: p16(x, residual=res)):.3f}  presplit both outs {timeit(conv):.3f}"
              f"  fp32 out only {timeit(lambda: conv(out, None)):.3f}  split out only {timeit(lambda: conv(None, outs)):.3f}  pack {timeit(lambda: split(x)):.3f}", flush=True)
run(1, 32, 32, (6, 9, 50))
run(2, 13, 20, (5, 7, 33))
run(1, 64, 64, (8, 16, 48))
run(8, 32, 32, (48, 128, 240), time=True)
run(8, 64, 64, (24, 64, 120), time=True)
