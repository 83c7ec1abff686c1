"""Layer enumeration of the YOLOv8 detect graph (Models/Yolo.cs:41-89, Modules/Head.cs:35-53) used to state the
ALGORITHMIC work of a step: conv MACs and compulsory activation bytes.  Pure host arithmetic."""

SIZES = {"n": (0.34, 0.25, 1024), "s": (0.34, 0.5, 1024), "m": (0.67, 0.75, 576), "l": (1.0, 1.0, 512), "x": (1.0, 1.25, 640)}


def v8_detect_convs(size="n", nc=80, reg_max=16, H=640, W=640):
    """Returns a list of dicts (name, cin, cout, k, s, hin, win, hout, wout, first) in execution order."""
    dm, wm, mc = SIZES[size]
    w = [min(int(x * wm), mc) for x in (64, 128, 256, 512, 1024)]
    d = [int(x * dm) for x in (3, 6, 9)]
    out = []

    def conv(name, cin, cout, k, s, h, ww):
        ho, wo = (h + 2 * (k // 2) - k) // s + 1, (ww + 2 * (k // 2) - k) // s + 1
        out.append(dict(name=name, cin=cin, cout=cout, k=k, s=s, hin=h, win=ww, hout=ho, wout=wo, first=not out))
        return ho, wo

    def c2f(name, c1, c2, n, h, ww):
        c = int(c2 * 0.5)
        conv(name + ".cv1", c1, 2 * c, 1, 1, h, ww)
        for i in range(n):
            conv(f"{name}.m.{i}.cv1", c, c, 3, 1, h, ww)
            conv(f"{name}.m.{i}.cv2", c, c, 3, 1, h, ww)
        conv(name + ".cv2", (2 + n) * c, c2, 1, 1, h, ww)

    h, ww = conv("model.0", 3, w[0], 3, 2, H, W)
    h, ww = conv("model.1", w[0], w[1], 3, 2, h, ww)
    c2f("model.2", w[1], w[1], d[0], h, ww)
    h8, w8 = conv("model.3", w[1], w[2], 3, 2, h, ww)
    c2f("model.4", w[2], w[2], d[1], h8, w8)
    h16, w16 = conv("model.5", w[2], w[3], 3, 2, h8, w8)
    c2f("model.6", w[3], w[3], d[1], h16, w16)
    h32, w32 = conv("model.7", w[3], w[4], 3, 2, h16, w16)
    c2f("model.8", w[4], w[4], d[0], h32, w32)
    conv("model.9.cv1", w[4], w[4] // 2, 1, 1, h32, w32)
    conv("model.9.cv2", 2 * w[4], w[4], 1, 1, h32, w32)
    c2f("model.12", w[4] + w[3], w[3], d[0], h16, w16)
    c2f("model.15", w[3] + w[2], w[2], d[0], h8, w8)
    conv("model.16", w[2], w[2], 3, 2, h8, w8)
    c2f("model.18", w[2] + w[3], w[3], d[0], h16, w16)
    conv("model.19", w[3], w[3], 3, 2, h16, w16)
    c2f("model.21", w[3] + w[4], w[4], d[0], h32, w32)
    ch = (w[2], w[3], w[4])
    hw = ((h8, w8), (h16, w16), (h32, w32))
    c2, c3 = max(16, ch[0] // 4, reg_max * 4), max(ch[0], min(nc, 100))
    for t, (cm, co) in enumerate(((c2, 4 * reg_max), (c3, nc))):
        for i in range(3):
            p = f"model.22.cv{2 + t}.{i}"
            conv(p + ".0", ch[i], cm, 3, 1, *hw[i])
            conv(p + ".1", cm, cm, 3, 1, *hw[i])
            conv(p + ".2", cm, co, 1, 1, *hw[i])
    return out


def step_work(size="n", nc=80, H=640, W=640, elem_bytes=2):
    """Per-image algorithmic work of one training step: conv MACs and compulsory activation bytes
    (per Conv unit: fwd in+out, bwd 2*in + 2*out), plus the per-launch-class split used for the roofline object."""
    L = v8_detect_convs(size, nc, 16, H, W)
    macs = sum(c["cout"] * c["cin"] * c["k"] ** 2 * c["hout"] * c["wout"] for c in L)
    s_in = sum(c["cin"] * c["hin"] * c["win"] for c in L)
    s_out = sum(c["cout"] * c["hout"] * c["wout"] for c in L)
    dgrad_macs = sum(c["cout"] * c["cin"] * c["k"] ** 2 * c["hout"] * c["wout"] for c in L if not c["first"])
    s_in_nofirst = sum(c["cin"] * c["hin"] * c["win"] for c in L if not c["first"])
    s_out_nofirst = sum(c["cout"] * c["hout"] * c["wout"] for c in L if not c["first"])
    return {
        "convs": len(L), "fwd_flop": 2 * macs, "train_flop": 2 * (2 * macs + dgrad_macs),
        "infer_bytes": (s_in + s_out) * elem_bytes, "train_bytes": 3 * (s_in + s_out) * elem_bytes,
        # conv_igemm launches of a training step: forward (in+out) and dgrad (dy in, dx out; stem has no dgrad)
        "igemm_launches": 2 * len(L) - 1,
        "igemm_bytes": ((s_in + s_out) + (s_in_nofirst + s_out_nofirst)) * elem_bytes,
        "igemm_flop": 2 * (macs + dgrad_macs),
        "wgrad_launches": len(L), "wgrad_bytes": (s_in + s_out) * elem_bytes, "wgrad_flop": 2 * macs,
    }
