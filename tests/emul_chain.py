"""Software model of csrc/mlp_chain.hip, used by the CPU tests.

It walks the packed weight blob with exactly the kernel's data flow - lane/register layout of
``v_mfma_f32_16x16x4_f32``, chunk order, k-outer accumulation, slot order of the positional
encoding, the in-register skip connection, the quad layout of the forward-mode tangents - but in
numpy on the CPU, so the packing plan and the kernel's index arithmetic can be validated against
the oracle without a GPU.  One call emulates one wave (16 columns).
"""
import numpy as np

HDR_INTS, HDR_OFFS, KT = 512, 16, 4096
LANE = np.arange(64)
G, J = LANE // 16, LANE % 16


def mfma_16x16x4(a, b, c):
    """a[64], b[64] (one VGPR each), c[64,4] -> d[64,4] with the gfx950 lane layouts:
    A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], C[row=4*(l>>4)+r][col=l&15]."""
    A = np.zeros((16, 4), np.float32)
    B = np.zeros((4, 16), np.float32)
    A[J, G] = a
    B[G, J] = b
    Cm = (A.astype(np.float64) @ B.astype(np.float64)).astype(np.float32)
    d = c.copy()
    for r in range(4):
        d[:, r] += Cm[4 * G + r, J]
    return d


class Blob:
    def __init__(self, blob: np.ndarray):
        self.f = np.ascontiguousarray(blob, dtype=np.float32)
        self.hdr = self.f[:HDR_INTS].view(np.int32)
        assert self.hdr[0] == 0x4E414631
        self.nc = int(self.hdr[2])
        self.offs = self.hdr[HDR_OFFS: HDR_OFFS + self.nc + 1]
        self.aux = self.f[self.hdr[4]: self.hdr[4] + self.hdr[5]]
        self.c = 0

    def acquire(self):
        w = self.f[self.offs[self.c]: self.offs[self.c + 1]]
        self.c = (self.c + 1) % self.nc
        return w


def mma_ktile(acc, xt, w, full16):
    for T in range(16 if full16 else 14):
        a = w[T * 256: (T + 1) * 256].reshape(64, 4)
        for r in range(4):
            acc[T] = mfma_16x16x4(a[:, r], xt[:, r], acc[T])


def softplus100(z):
    en = np.exp2(np.abs(z) * np.float32(-144.269504088896340736), dtype=np.float32)
    return (np.maximum(z, 0) + np.log2(1 + en, dtype=np.float32) * np.float32(0.69314718055994530942 / 100.0)).astype(np.float32)


def softplus100_grad(z):
    en = np.exp2(np.abs(z) * np.float32(-144.269504088896340736), dtype=np.float32)
    r = 1.0 / (1.0 + en)
    return np.where(z >= 0, r, en * r).astype(np.float32)


def run_layer(X, blob, bias, nt_base, nextra, full16, act, tangent):
    """X: list of [64,4] tiles (in place).  act in {'softplus','relu','none'}."""
    acc = [np.zeros((64, 4), np.float32) for _ in range(16)]
    for c in range((nt_base + 1) // 2):
        w = blob.acquire()
        mma_ktile(acc, X[2 * c], w, full16)
        if 2 * c + 1 < nt_base:
            mma_ktile(acc, X[2 * c + 1], w[KT:], full16)
    for c in range((nextra + 1) // 2):
        w = blob.acquire()
        mma_ktile(acc, X[nt_base + 2 * c], w, full16)
        if 2 * c + 1 < nextra:
            mma_ktile(acc, X[nt_base + 2 * c + 1], w[KT:], full16)
    is_val = (LANE & 3) == 0 if tangent else np.ones(64, bool)
    for T in range(16 if full16 else 14):
        b = bias[T * 16 + G[:, None] * 4 + np.arange(4)[None, :]]
        z = acc[T] + np.where(is_val[:, None], b, 0).astype(np.float32)
        if act == "softplus":
            if tangent:
                d = softplus100_grad(z)
                d = d[LANE & ~3]                       # quad broadcast of lane 0
                X[T] = np.where(is_val[:, None], softplus100(z), d * acc[T]).astype(np.float32)
            else:
                X[T] = softplus100(z)
        elif act == "relu":
            X[T] = np.maximum(z, 0).astype(np.float32)
        else:
            X[T] = z.astype(np.float32)


def encode_slots(p, q):
    """p[64,3] point of every lane, q[64] (-1 value / 0..2 derivative coordinate) -> 3 tiles [64,4]."""
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    last = G == 3
    cg = np.where(G == 0, x, np.where(G == 1, y, z))
    s, c = [], []
    for i in range(5):
        co = [x, y, z][i] if i < 3 else np.zeros_like(x)
        a = np.where(last, co * np.float32(32.0) if i < 3 else 0.0, cg * np.float32(1 << i)).astype(np.float32)
        s.append(np.sin(a, dtype=np.float32)); c.append(np.cos(a, dtype=np.float32))
    m = np.zeros((64, 12), np.float32)
    val = q < 0
    own = q == G
    mc = np.zeros((64, 12), np.float32); ml = np.zeros((64, 12), np.float32)
    mc[:, 0] = np.where(val, cg, np.where(own, 1.0, 0.0))
    for k in range(5):
        f = np.float32(1 << k)
        mc[:, 1 + 2 * k] = np.where(val, s[k], np.where(own, c[k] * f, 0.0))
        mc[:, 2 + 2 * k] = np.where(val, c[k], np.where(own, -(s[k] * f), 0.0))
    for i in range(3):
        ml[:, 2 * i] = np.where(val, s[i], np.where(q == i, c[i] * np.float32(32.0), 0.0))
        ml[:, 2 * i + 1] = np.where(val, c[i], np.where(q == i, -(s[i] * np.float32(32.0)), 0.0))
    m = np.where(last[:, None], ml, mc).astype(np.float32)
    return [m[:, 4 * t: 4 * t + 4].copy() for t in range(3)]


def surface_hidden(blob, p, q, tangent):
    X = [None] * 19
    E = encode_slots(p, q)
    X[0:3] = E
    run_layer(X, blob, blob.aux[0:256], 3, 0, True, "softplus", tangent)
    rs2 = np.float32(1.41421356237309504880)
    for L in range(1, 8):
        if L == 4:
            E = encode_slots(p, q)
            for t in range(14):
                X[t] = (X[t] / rs2).astype(np.float32)
            for t in range(3):
                X[14 + t] = (E[t] / rs2).astype(np.float32)
        run_layer(X, blob, blob.aux[L * 256:(L + 1) * 256], 16, 1 if L == 4 else 0, L != 3, "softplus", tangent)
    return X


def dot_row16(X, row):
    s = np.zeros(64, np.float32)
    for t in range(16):
        wv = row[t * 16 + G[:, None] * 4 + np.arange(4)[None, :]]
        s += (X[t] * wv).sum(1)
    tot = s.reshape(4, 16).sum(0)          # over the 4 lane groups
    return tot[J]


def emul_sdf_only(blob_np, pts16, R_bg):
    """pts16 [16,3] -> sdf[16]"""
    blob = Blob(blob_np)
    p = pts16[J].astype(np.float32)
    X = surface_hidden(blob, p, np.full(64, -1), False)
    sdf = dot_row16(X, blob.aux[2048:2304]) + blob.aux[2304]
    if R_bg > 0:
        sdf = np.minimum(sdf, R_bg - np.sqrt((p ** 2).sum(1)))
    return sdf[:16]


def emul_sdf_nabla(blob_np, pts4, R_bg):
    """pts4 [4,3] -> sdf[4], nabla[4,3], h7[4,256]"""
    blob = Blob(blob_np)
    p = pts4[J >> 2].astype(np.float32)
    cq = J & 3
    X = surface_hidden(blob, p, cq - 1, True)
    v = dot_row16(X, blob.aux[2048:2304])
    sdf = np.zeros(4, np.float32); nab = np.zeros((4, 3), np.float32); h7 = np.zeros((4, 256), np.float32)
    for lane in range(64):
        pi = J[lane] >> 2
        if cq[lane] == 0:
            if G[lane] == 0:
                sv = v[lane] + blob.aux[2304]
                if R_bg > 0:
                    d_bg = R_bg - np.sqrt((p[lane] ** 2).sum())
                    sv = d_bg if d_bg < sv else sv
                sdf[pi] = sv
            for t in range(16):
                h7[pi, t * 16 + G[lane] * 4: t * 16 + G[lane] * 4 + 4] = X[t][lane]
        elif G[lane] == 0:
            nab[pi, cq[lane] - 1] = v[lane]
    return sdf, nab, h7


# ---- reverse-mode kernel (k_sdf_grad): forward sweep with softplus' kept per layer, then the transposed chunks ----------------
def run_layer_fwd_d(X, blob, bias, nt_base, nextra, full16):
    """run_layer(softplus) that also returns softplus'(z) per output tile (what the kernel parks in its scratch)."""
    acc = [np.zeros((64, 4), np.float32) for _ in range(16)]
    for c in range((nt_base + 1) // 2):
        w = blob.acquire()
        mma_ktile(acc, X[2 * c], w, full16)
        if 2 * c + 1 < nt_base:
            mma_ktile(acc, X[2 * c + 1], w[KT:], full16)
    for c in range((nextra + 1) // 2):
        w = blob.acquire()
        mma_ktile(acc, X[nt_base + 2 * c], w, full16)
    D = [None] * 16
    for T in range(16 if full16 else 14):
        z = acc[T] + bias[T * 16 + G[:, None] * 4 + np.arange(4)[None, :]]
        X[T] = softplus100(z)
        D[T] = softplus100_grad(z)
    return D


def run_layer_T(X, blob, nt_k, full16, d_below, scale):
    acc = [np.zeros((64, 4), np.float32) for _ in range(16)]
    for c in range(nt_k // 2):
        w = blob.acquire()
        mma_ktile(acc, X[2 * c], w, full16)
        mma_ktile(acc, X[2 * c + 1], w[KT:], full16)
    for T in range(16 if full16 else 14):
        X[T] = (acc[T] * np.float32(scale) * d_below[T]).astype(np.float32)


def run_tail_T(X, blob, E, scale):
    acc = [np.zeros((64, 4), np.float32) for _ in range(3)]
    for c in range(2):
        w = blob.acquire()
        assert len(w) == 8 * 768
        for kt in range(8):
            for i in range(3):
                a = w[kt * 768 + i * 256: kt * 768 + (i + 1) * 256].reshape(64, 4)
                for r in range(4):
                    acc[i] = mfma_16x16x4(a[:, r], X[8 * c + kt][:, r], acc[i])
    for i in range(3):
        E[i] = (E[i] + acc[i] * np.float32(scale)).astype(np.float32)


def emul_sdf_grad(blob_np, pts16, R_bg):
    """pts16 [16,3] -> sdf[16], nabla[16,3], h7[16,256] the way k_sdf_grad computes them."""
    blob = Blob(blob_np)
    blob.nc = int(blob.hdr[6])                                   # forward + reverse chunks
    blob.offs = blob.hdr[HDR_OFFS: HDR_OFFS + blob.nc + 1]
    p = pts16[J].astype(np.float32)
    q = np.full(64, -1)
    X = [None] * 19
    X[0:3] = encode_slots(p, q)
    D = [None] * 8
    D[0] = run_layer_fwd_d(X, blob, blob.aux[0:256], 3, 0, True)
    rs2 = np.float32(1.41421356237309504880)
    for L in range(1, 8):
        if L == 4:
            E = encode_slots(p, q)
            for t in range(14):
                X[t] = (X[t] / rs2).astype(np.float32)
            for t in range(3):
                X[14 + t] = (E[t] / rs2).astype(np.float32)
        D[L] = run_layer_fwd_d(X, blob, blob.aux[L * 256:(L + 1) * 256], 16, 1 if L == 4 else 0, L != 3)
    sdf = dot_row16(X, blob.aux[2048:2304]) + blob.aux[2304]
    if R_bg > 0:
        d_bg = R_bg - np.sqrt((p ** 2).sum(1))
        sdf = np.where(d_bg < sdf, d_bg, sdf)
    h7 = np.zeros((16, 256), np.float32)
    for lane in range(64):
        for t in range(16):
            h7[J[lane], t * 16 + G[lane] * 4: t * 16 + G[lane] * 4 + 4] = X[t][lane]
    row = blob.aux[2048:2304]
    for T in range(16):
        X[T] = (row[T * 16 + G[:, None] * 4 + np.arange(4)[None, :]] * D[7][T]).astype(np.float32)
    E = [np.zeros((64, 4), np.float32) for _ in range(3)]
    inv = np.float32(0.70710678118654752440)
    for L in (7, 6, 5):
        run_layer_T(X, blob, 16, True, D[L - 1], 1.0)
    run_tail_T(X, blob, E, inv)
    run_layer_T(X, blob, 16, False, D[3], inv)
    run_layer_T(X, blob, 14, True, D[2], 1.0)
    for L in (2, 1):
        run_layer_T(X, blob, 16, True, D[L - 1], 1.0)
    run_tail_T(X, blob, E, 1.0)
    assert blob.c == 0, "the reverse sweep must end exactly at the end of the chunk table"
    nab = np.zeros((16, 3), np.float32)
    for qq in range(3):
        Jc = encode_slots(p, np.full(64, qq))
        a = sum((E[t] * Jc[t]).sum(1) for t in range(3))
        nab[:, qq] = a.reshape(4, 16).sum(0)
    return sdf[:16], nab, h7


def emul_radiance(blob_np, view_tiles, pts16, view16, nabla16, h7_16):
    """[16,3] x3, h7 [16,256] -> rgb[16,3]"""
    blob = Blob(blob_np)
    p, v, n = pts16[J].astype(np.float32), view16[J].astype(np.float32), nabla16[J].astype(np.float32)
    X = [None] * 19
    for t in range(16):
        X[t] = np.stack([h7_16[J[l], t * 16 + G[l] * 4: t * 16 + G[l] * 4 + 4] for l in range(64)]).astype(np.float32)
    ne = 9 if view_tiles == 1 else 33
    ex = np.zeros((64, 16 * view_tiles), np.float32)
    ex[:, 0:3] = p
    if view_tiles == 1:
        ex[:, 3:6] = v
    else:
        ex[:, 3:6] = v
        for k in range(4):
            f = np.float32(1 << k)
            ex[:, 6 + 6 * k: 9 + 6 * k] = np.sin(v * f, dtype=np.float32)
            ex[:, 9 + 6 * k: 12 + 6 * k] = np.cos(v * f, dtype=np.float32)
    ex[:, ne - 3: ne] = n
    for t in range(view_tiles):
        X[16 + t] = np.stack([ex[l, 16 * t + 4 * G[l]: 16 * t + 4 * G[l] + 4] for l in range(64)])
    for L in range(5):
        run_layer(X, blob, blob.aux[L * 256:(L + 1) * 256], 16, view_tiles if L == 1 else 0, True,
                  "none" if L == 0 else "relu", False)
    rgb = np.zeros((16, 3), np.float32)
    for c in range(3):
        z = dot_row16(X, blob.aux[1280 + 256 * c: 1280 + 256 * (c + 1)]) + blob.aux[2048 + c]
        rgb[:, c] = (1.0 / (1.0 + np.exp(-z)))[:16]
    return rgb


# ======================================================================================================
# Software model of csrc/mlp_chain_bf16.hip (split-bf16 "bf16x3", v_mfma_f32_16x16x32_bf16 layouts).
# One call emulates one wave: 16 columns, lane = 16 g + j.
# ======================================================================================================
import torch as _torch

TS_FLOATS = 512


TERM = "bf16"        # "fp16": the 2-MFMA variant (csrc/mlp_chain_f16x2.hip, precision 4) - fp16 fragments, single-term activations


def _bf16(x):
    dt = _torch.float16 if TERM == "fp16" else _torch.bfloat16
    return _torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(dt).to(_torch.float32).numpy()


def _decode16(raw):
    """uint16 fragment bits -> float32 values (bf16, or fp16 for the 2-MFMA blobs)"""
    if TERM == "fp16":
        return raw.view(np.float16).astype(np.float32)
    return (raw.astype(np.uint32) << 16).view(np.float32)


FULL_INPUTS = True   # fp16 variant as built: the READY-MADE input units (positional encodings, the radiance net's extras and h7 rows) keep
                     # their lo term (3 MFMAs on those k-steps); False = the pure 2-MFMA form (mean sdf error 2.4e-4 instead of 1.4e-4)


def split2(x, full=False):
    hi = _bf16(x)
    if TERM == "fp16" and not (full and FULL_INPUTS):
        return hi, np.zeros_like(hi)                      # one activation term: the a_hi . b_lo product of the emulation adds zero
    lo = _bf16(x.astype(np.float32) - hi)
    return hi, lo


def mfma_16x16x32(a, b, c):
    """a, b: [64 lanes, 8] (bf16 values as float32); c: [64, 4].  Slot (g, e) of A (lane 16g + i = row i) pairs
    with slot (g, e) of B (lane 16g + j = column j); C register r of lane (g, j) is row 4g + r, column j."""
    A = np.zeros((16, 32), np.float64); B = np.zeros((32, 16), np.float64)
    for e in range(8):
        A[J, 8 * G + e] = a[:, e]
        B[8 * G + e, J] = b[:, e]
    Cm = A @ B
    d = c.copy()
    for r in range(4):
        d[:, r] = (d[:, r].astype(np.float64) + Cm[4 * G + r, J]).astype(np.float32)
    return d


def _frag(w, kk, T):
    """chunk float array -> (A_hi, A_lo) [64, 8] bf16 values for k-step kk of the chunk, output tile T"""
    o = (kk * 16 + T) * TS_FLOATS
    raw = np.ascontiguousarray(w[o:o + TS_FLOATS]).view(np.uint16).reshape(2, 64, 8)
    f = _decode16(raw)
    return f[0], f[1]


def _tile_feat(T):
    return 16 * T + 4 * G[:, None] + np.arange(4)[None, :]          # [64, 4]


def run_layer_bf16(Xh, Xl, blob, bias, rows, nu_base, nextra, act, tangent, last, dots, h7=None):
    """k-outer: Xh/Xl lists of [64, 8] units, updated in place (units 0..7) unless last."""
    is_val = (LANE & 3) == 0 if tangent else np.ones(64, bool)
    acc = [np.where(is_val[:, None], bias[_tile_feat(T)], 0).astype(np.float32) for T in range(16)]

    def kstep(ks, w, kk):
        for T in range(16):
            ah, al = _frag(w, kk, T)
            acc[T] = mfma_16x16x32(ah, Xh[ks], acc[T])
            acc[T] = mfma_16x16x32(ah, Xl[ks], acc[T])
            acc[T] = mfma_16x16x32(al, Xh[ks], acc[T])
    for c0 in range(0, nu_base, 2):
        w = blob.acquire()
        for kk, ks in enumerate(range(c0, min(c0 + 2, nu_base))):
            kstep(ks, w, kk)
    if nextra:
        w = blob.acquire()
        for x in range(nextra):
            kstep(nu_base + x, w, x)
    Y = []
    for T in range(16):
        a = acc[T]
        if act == "softplus":
            if tangent:
                d = softplus100_grad(a)[LANE & ~3]
                y = np.where(is_val[:, None], softplus100(a), d * a).astype(np.float32)
            else:
                y = softplus100(a)
        elif act == "relu":
            y = np.maximum(a, 0).astype(np.float32)
        else:
            y = a.astype(np.float32)
        Y.append(y)
    if last:
        for T in range(16):
            feat = _tile_feat(T)
            for n in range(len(dots)):
                dots[n] += (Y[T] * rows[n * 256 + feat]).sum(1).astype(np.float32)
            if h7 is not None:
                for lane in range(64):
                    if is_val[lane]:
                        h7[J[lane] >> 2 if tangent else J[lane], feat[lane]] = Y[T][lane]
    else:
        for U in range(8):
            Xh[U], Xl[U] = split2(np.concatenate([Y[2 * U], Y[2 * U + 1]], axis=1))


def encode_units_bf16(p, dq):
    """2 units of [64, 8]: lane group g < 3 owns coordinate g; m = 8q + e: 0 raw, 1 + 2k sin, 2 + 2k cos (k < 6)."""
    cg = np.where(G == 0, p[:, 0], np.where(G == 1, p[:, 1], p[:, 2])).astype(np.float32)
    val = dq < 0
    own = dq == G
    m = np.zeros((64, 16), np.float32)
    m[:, 0] = np.where(val, cg, own.astype(np.float32))
    for k in range(6):
        f = np.float32(1 << k)
        s = np.sin(cg * f, dtype=np.float32); c = np.cos(cg * f, dtype=np.float32)
        m[:, 1 + 2 * k] = np.where(val, s, np.where(own, c * f, 0.0))
        m[:, 2 + 2 * k] = np.where(val, c, np.where(own, -(s * f), 0.0))
    m[G == 3] = 0
    return [split2(m[:, 8 * q: 8 * q + 8], full=True) for q in range(2)]


def _group_sum(d):
    return d.reshape(4, 16).sum(0)[J]


def surface_chain_bf16(blob, p, dq, tangent, h7=None):
    Xh, Xl = [None] * 10, [None] * 10
    for q, (hi, lo) in enumerate(encode_units_bf16(p, dq)):
        Xh[q], Xl[q] = hi, lo
    dots = [np.zeros(64, np.float32)]
    rows = blob.aux[2048:2304]
    run_layer_bf16(Xh, Xl, blob, blob.aux[0:256], rows, 2, 0, "softplus", tangent, False, dots)
    for L in range(1, 8):
        if L == 4:          # 1/sqrt(2) of the skip concat lives in layer 4's packed weights
            for q, (hi, lo) in enumerate(encode_units_bf16(p, dq)):
                Xh[7 + q], Xl[7 + q] = hi, lo
        run_layer_bf16(Xh, Xl, blob, blob.aux[L * 256:(L + 1) * 256], rows, 8, 1 if L == 4 else 0, "softplus",
                       tangent, L == 7, dots, h7 if L == 7 else None)
    return _group_sum(dots[0])


def emul_sdf_only_bf16(blob_np, pts16, R_bg):
    blob = Blob(blob_np)
    p = pts16[J].astype(np.float32)
    sdf = surface_chain_bf16(blob, p, np.full(64, -1), False) + blob.aux[2304]
    if R_bg > 0:
        sdf = np.minimum(sdf, R_bg - np.sqrt((p ** 2).sum(1)))
    return sdf[:16]


def emul_sdf_nabla_bf16(blob_np, pts4, R_bg):
    blob = Blob(blob_np)
    p = pts4[J >> 2].astype(np.float32)
    cq = J & 3
    h7 = np.zeros((4, 256), np.float32)
    v = surface_chain_bf16(blob, p, cq - 1, True, h7)
    sdf = np.zeros(4, np.float32); nab = np.zeros((4, 3), np.float32)
    for lane in range(16):
        pi = lane >> 2
        if cq[lane] == 0:
            sv = v[lane] + blob.aux[2304]
            if R_bg > 0:
                d_bg = R_bg - np.sqrt((p[lane] ** 2).sum())
                sv = d_bg if d_bg < sv else sv
            sdf[pi] = sv
        else:
            nab[pi, cq[lane] - 1] = v[lane]
    return sdf, nab, h7


def emul_radiance_bf16(blob_np, view_tiles, pts16, view16, nabla16, h7_16):
    blob = Blob(blob_np)
    p, v, n = pts16[J].astype(np.float32), view16[J].astype(np.float32), nabla16[J].astype(np.float32)
    Xh, Xl = [None] * 10, [None] * 10
    e8 = np.arange(8)
    for u in range(8):
        feat = 32 * u + np.where(e8[None, :] < 4, 4 * G[:, None] + e8[None, :], 16 + 4 * G[:, None] + e8[None, :] - 4)
        Xh[u], Xl[u] = split2(h7_16[J[:, None], feat].astype(np.float32), full=True)
    ne = 9 if view_tiles == 1 else 33
    ve = 1 if view_tiles == 1 else 2
    ex = np.zeros((64, 32 * ve), np.float32)
    ex[:, 0:3] = p; ex[:, 3:6] = v
    if view_tiles == 3:
        for k in range(4):
            f = np.float32(1 << k)
            ex[:, 6 + 6 * k: 9 + 6 * k] = np.sin(v * f, dtype=np.float32)
            ex[:, 9 + 6 * k: 12 + 6 * k] = np.cos(v * f, dtype=np.float32)
    ex[:, ne - 3: ne] = n
    for q in range(ve):
        idx = 32 * q + 8 * G[:, None] + e8[None, :]
        Xh[8 + q], Xl[8 + q] = split2(ex[np.arange(64)[:, None], idx], full=True)
    dots = [np.zeros(64, np.float32) for _ in range(3)]
    rows = blob.aux[1280:1280 + 768]
    for L in range(5):
        run_layer_bf16(Xh, Xl, blob, blob.aux[L * 256:(L + 1) * 256], rows, 8, ve if L == 1 else 0,
                       "none" if L == 0 else "relu", False, L == 4, dots)
    rgb = np.zeros((16, 3), np.float32)
    for c in range(3):
        z = _group_sum(dots[c]) + blob.aux[2048 + c]
        rgb[:, c] = (1.0 / (1.0 + np.exp(-z)))[:16]
    return rgb


# ======================================================================================================
# Software model of k_sdf_grad_bf16 (reverse-mode d sdf / d x): forward sweep keeping softplus'(z_l) as
# unorm16, backward sweep through the transposed-weight chunks that follow the forward program in the blob.
# ======================================================================================================
D_UNORM = np.float32(65535.0)


def _acc_layer(units, blob, init, ntiles=16, tiles_per_kstep=16):
    """units: list of (hi, lo) [64, 8] in k-step order; init: list of [64, 4].  Chunks of 2 k-steps."""
    acc = [a.copy() for a in init]
    ks = 0
    while ks < len(units):
        w = blob.acquire()
        for kk in range(min(2, len(units) - ks)):
            for T in range(ntiles):
                o = (kk * tiles_per_kstep + T) * TS_FLOATS
                raw = np.ascontiguousarray(w[o:o + TS_FLOATS]).view(np.uint16).reshape(2, 64, 8)
                f = _decode16(raw)
                acc[T] = mfma_16x16x32(f[0], units[ks + kk][0], acc[T])
                acc[T] = mfma_16x16x32(f[0], units[ks + kk][1], acc[T])
                acc[T] = mfma_16x16x32(f[1], units[ks + kk][0], acc[T])
        ks += 2
    return acc


def _units_of(Y):
    return [split2(np.concatenate([Y[2 * U], Y[2 * U + 1]], axis=1)) for U in range(8)]


def emul_sdf_grad_bf16(blob_np, pts16, R_bg):
    """pts16 [16,3] -> sdf[16], nabla[16,3], h7[16,256] with the data flow of k_sdf_grad_bf16."""
    blob = Blob(blob_np)
    blob.nc = int(blob.hdr[6])
    blob.offs = blob.hdr[HDR_OFFS: HDR_OFFS + blob.nc + 1]
    p = pts16[J].astype(np.float32)
    enc = encode_units_bf16(p, np.full(64, -1))
    zero = [np.zeros((64, 4), np.float32) for _ in range(16)]
    bias = lambda l: [blob.aux[l * 256:(l + 1) * 256][_tile_feat(T)].astype(np.float32) for T in range(16)]
    z = [None] * 8
    dq = [None] * 8
    z[0] = _acc_layer(enc, blob, bias(0))
    for l in range(1, 8):
        dq[l - 1] = [np.rint(softplus100_grad(a) * D_UNORM).astype(np.float32) for a in z[l - 1]]
        units = _units_of([softplus100(a) for a in z[l - 1]])
        if l == 4:
            units = units[:7] + enc
        z[l] = _acc_layer(units, blob, bias(l))
    a7 = [softplus100(a) for a in z[7]]
    row = blob.aux[2048:2304]
    dot = np.zeros(64, np.float32)
    h7 = np.zeros((16, 256), np.float32)
    for T in range(16):
        dot += (a7[T] * row[_tile_feat(T)]).sum(1).astype(np.float32)
        h7[J[:, None], _tile_feat(T)] = a7[T]
    sdf = _group_sum(dot) + blob.aux[2304]
    if R_bg > 0:
        d_bg = R_bg - np.sqrt((p ** 2).sum(1))
        sdf = np.where(d_bg < sdf, d_bg, sdf)
    # backward: layer 7 (inputs softplus'(z7), the sdf row is inside the weights), then 6..1
    P = _acc_layer(_units_of([softplus100_grad(a) for a in z[7]]), blob, zero)
    ge4 = None
    if TERM == "fp16":                                   # the fp16 blob does not absorb 1 / 65535: the kernel scales softplus' itself
        dq = [None if q is None else [(t * np.float32(1.0 / 65535.0)).astype(np.float32) for t in q] for q in dq]
    for l in range(6, 0, -1):
        P = _acc_layer(_units_of([(P[T] * dq[l][T]).astype(np.float32) for T in range(16)]), blob, zero)
        if l == 4:
            ge4 = [P[13].copy(), P[14].copy(), P[15].copy()]
    # layer 0: 3 output tiles (rows 217..255 = encoding features), all 8 k-steps in one chunk
    units = _units_of([(P[T] * dq[0][T]).astype(np.float32) for T in range(16)])
    w = blob.acquire()
    E = [a.copy() for a in ge4]
    for ks in range(8):
        for t in range(3):
            o = (ks * 3 + t) * TS_FLOATS
            raw = np.ascontiguousarray(w[o:o + TS_FLOATS]).view(np.uint16).reshape(2, 64, 8)
            f = _decode16(raw)
            E[t] = mfma_16x16x32(f[0], units[ks][0], E[t])
            E[t] = mfma_16x16x32(f[0], units[ks][1], E[t])
            E[t] = mfma_16x16x32(f[1], units[ks][0], E[t])
    # contraction with d enc / d x: lane (g, j), tile t, reg r holds d sdf / d enc[f], f = 16 t + 4 g + r - 9
    part = np.zeros((64, 3), np.float32)
    for t in range(3):
        for r in range(4):
            f = 16 * t + 4 * G + r - 9
            for lane in range(64):
                ff = f[lane]
                if ff < 0 or ff >= 39:
                    continue
                if ff < 3:
                    c, jac = ff, np.float32(1.0)
                else:
                    k, rem = divmod(ff - 3, 6)
                    s_, c = divmod(rem, 3)
                    fr = np.float32(1 << k)
                    arg = np.float32(p[lane, c] * fr)
                    jac = np.float32(np.cos(arg) * fr) if s_ == 0 else np.float32(-np.sin(arg) * fr)
                part[lane, c] += E[t][lane, r] * jac
    nabla = np.stack([_group_sum(part[:, c]) for c in range(3)], axis=1)
    return sdf[:16], nabla[:16], h7


# ======================================================================================================
# Software model of k_radiance_bwd_bf16: the transposed-weight chunks that follow the forward radiance program.
# Inputs are the forward activations as the forward kernel dumps them (relu outputs r0..r3) - only their sign is
# used (masks) - plus d loss / d rgb and rgb.  Returns g_h7 [16,256], g_n [16,3] and the deltas of every layer.
# ======================================================================================================
def emul_radiance_bwd_bf16(blob_np, rgb16, g_rgb16, r_acts):
    """r_acts: dict l -> [16, 256] relu outputs (natural feature order), l = 0..3."""
    blob = Blob(blob_np)
    nc_fwd, nc_all = int(blob.hdr[2]), int(blob.hdr[6])
    blob.nc = nc_all
    blob.offs = blob.hdr[HDR_OFFS: HDR_OFFS + nc_all + 1]
    blob.c = nc_fwd
    rows = blob.aux[1280:1280 + 768].reshape(3, 256)
    d4 = (g_rgb16 * rgb16 * (1.0 - rgb16)).astype(np.float32)[J]                   # [64, 3]
    P = [(d4[:, 0:1] * rows[0][_tile_feat(T)] + d4[:, 1:2] * rows[1][_tile_feat(T)] + d4[:, 2:3] * rows[2][_tile_feat(T)]).astype(np.float32)
         for T in range(16)]
    zero = [np.zeros((64, 4), np.float32) for _ in range(16)]
    mask = lambda l: [(r_acts[l][J[:, None], _tile_feat(T)] > 0).astype(np.float32) for T in range(16)]
    deltas = {}
    for l in (3, 2, 1):
        m = mask(l)
        dl = [(P[T] * m[T]).astype(np.float32) for T in range(16)]
        deltas[l] = dl
        P = _acc_layer(_units_of(dl), blob, zero)
    m = mask(0)
    d0 = [(P[T] * m[T]).astype(np.float32) for T in range(16)]
    deltas[0] = d0
    units = _units_of(d0)
    w = blob.acquire()                                   # normal rows of R0^T: 8 k-steps x 1 tile
    E = np.zeros((64, 4), np.float32)
    for ks in range(8):
        raw = np.ascontiguousarray(w[ks * TS_FLOATS:(ks + 1) * TS_FLOATS]).view(np.uint16).reshape(2, 64, 8)
        f = (raw.astype(np.uint32) << 16).view(np.float32)
        E = mfma_16x16x32(f[0], units[ks][0], E)
        E = mfma_16x16x32(f[0], units[ks][1], E)
        E = mfma_16x16x32(f[1], units[ks][0], E)
    g_n = E[:16, :3].copy()                              # lanes g = 0: rows 0..2
    P = _acc_layer(units, blob, zero)                    # g_feat
    deltas["f"] = P
    P = _acc_layer(_units_of(P), blob, zero)             # g_h7 = W8[1:]^T g_feat
    g_h7 = np.zeros((16, 256), np.float32)
    nat = lambda tiles: np.stack([np.concatenate([tiles[T][16 * g + j] for T in range(16) for g in [gg]]) for j in range(16) for gg in [0]]) if False else None
    for T in range(16):
        g_h7[J[:, None], _tile_feat(T)] = P[T]
    def to_nat(tiles):
        out = np.zeros((16, 256), np.float32)
        for T in range(16):
            out[J[:, None], _tile_feat(T)] = tiles[T]
        return out
    return g_h7, g_n, {k: to_nat(v) for k, v in deltas.items()}
